#!/bin/bash
# round-4 experiment: the values-phase Goldilocks permutation, round-3 form against round-4 form, on one wavefront
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
hipcc -O3 -std=c++17 --offload-arch=gfx950 $1 -I halo2-plonky2-verifier_amd/csrc -I include -Wno-unused-result -Wno-unused-value tools/ubench/ubench_glperm.hip -o gpurun_out/ubench_glperm 2>/dev/null
timeout -k 10 120 gpurun_out/ubench_glperm | tee gpurun_out/r04_ubench_glperm.txt
