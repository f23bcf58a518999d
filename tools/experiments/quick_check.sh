#!/bin/bash
# round-4: the permutation alone and in the prologue (quick check between edits)
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
bash tools/experiments/glperm_ubench.sh | grep -v "constants [4-9]\|constants 1[01]"
timeout -k 10 200 python3 tools/launch_timing.py --batch 1 --reps 3 2>/dev/null | grep config
timeout -k 10 200 python3 tools/launch_timing.py --batch 1 --reps 3 --hash gl 2>/dev/null | grep config
timeout -k 10 200 python3 tools/launch_timing.py --batch 4 --reps 3 --hash gl 2>/dev/null | grep config
timeout -k 10 300 python -m pytest tests/test_gpu_batch.py -x -q -m gpu -k "small_shapes or mds_tables or published or valid_fri or cap_height_five or shape_edge or packed_shard_layout" 2>&1 | tail -2
