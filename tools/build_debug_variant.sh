#!/bin/bash
# Builds halo2-plonky2-verifier_amd/libh2w_dbg.so: the same library with -DH2W_DEBUG_HOOKS (kernel-skipping switches for timing
# experiments; results are garbage).  Select it with H2W_LIB=<path>.  Never used by tests, smoke or the default bench.
set -e
cd "$(dirname "$0")/../halo2-plonky2-verifier_amd/csrc"
FLAGS="-DH2W_DEBUG_HOOKS $H2W_EXTRA -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -Wno-unused-value -x hip"
mkdir -p ../build_dbg
hipcc $FLAGS -c batch.hip -o ../build_dbg/batch.o
hipcc --offload-arch=gfx950 -shared -fPIC ../build/expand.o ../build/eager.o ../build_dbg/batch.o ../build/abi_backend.o ../build/prover.o -o ../libh2w_dbg.so
echo "built $(realpath ../libh2w_dbg.so)"
