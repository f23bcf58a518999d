#!/bin/bash
# the record-and-replay path after a change of the interpreter: its GPU tests, then the launch time by segment
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_replay.py -x -q -m gpu 2>&1 | tail -2
timeout -k 10 400 python3 tools/replay_timing.py --streams 2 2>&1 | tail -12 | tee gpurun_out/r04_replay_timing.txt
