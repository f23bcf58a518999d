#!/bin/bash
# round-4 experiment: what one wavefront pays per instruction, and the prologue without its permutations
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value tools/ubench/ubench_wave1.hip -o gpurun_out/ubench_wave1 2>/dev/null
timeout -k 10 120 gpurun_out/ubench_wave1 > gpurun_out/r04_ubench_wave1.txt 2>&1
timeout -k 10 200 python3 tools/launch_timing.py --batch 1 --reps 3 > gpurun_out/r04_prologue_stub.txt 2>&1
H2W_LIB=$PWD/halo2-plonky2-verifier_amd/libh2w_pstub.so timeout -k 10 200 python3 tools/launch_timing.py --batch 1 --reps 3 >> gpurun_out/r04_prologue_stub.txt 2>&1
cat gpurun_out/r04_ubench_wave1.txt gpurun_out/r04_prologue_stub.txt
