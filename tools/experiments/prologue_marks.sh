#!/bin/bash
# round-4 diagnostics: where the prologue's cycles go (a library built with -DH2W_EXP_GLP_CLOCK), the permutation alone
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
H2W_LIB=$PWD/halo2-plonky2-verifier_amd/libh2w_clk.so timeout -k 10 200 python3 tools/launch_timing.py --batch 1 --reps 1 > gpurun_out/r04_t11.txt 2>&1

cat gpurun_out/r04_t11.txt
