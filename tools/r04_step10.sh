#!/bin/bash
# round-4: the rewritten values-phase Goldilocks permutation - isolated check + timing, the GPU suite, one-proof kernel times
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash tools/r04_step9.sh
timeout -k 10 200 python3 tools/launch_timing.py --batch 1 --reps 3 > gpurun_out/r04_one_proof_after.txt 2>&1
timeout -k 10 200 python3 tools/launch_timing.py --batch 1 --reps 3 --hash gl >> gpurun_out/r04_one_proof_after.txt 2>&1
cat gpurun_out/r04_one_proof_after.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r04_t10.txt
