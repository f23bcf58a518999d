#!/bin/bash
# Builds halo2-plonky2-verifier_amd/libh2w_<name>.so: the same library with -DH2W_DEBUG_HOOKS (kernel-skipping switches for timing
# experiments; results are garbage) plus the flags in H2W_EXTRA (e.g. -DH2W_DBG_NOFLUSH).  Usage: build_debug_variant.sh [name=dbg].
# Select it with H2W_LIB=<path>.  Never used by tests, smoke or the default bench.
set -e
name=${1:-dbg}
cd "$(dirname "$0")/../halo2-plonky2-verifier_amd/csrc"
FLAGS="-DH2W_DEBUG_HOOKS $H2W_EXTRA -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -Wno-unused-value -x hip"
mkdir -p ../build_$name
hipcc $FLAGS -c batch.hip -o ../build_$name/batch.o
hipcc $FLAGS -c glue.hip -o ../build_$name/glue.o
hipcc $FLAGS -c expand.hip -o ../build_$name/expand.o
hipcc --offload-arch=gfx950 -shared -fPIC ../build_$name/expand.o ../build/eager.o ../build_$name/batch.o ../build_$name/glue.o ../build/chipbatch.o ../build/abi_backend.o ../build/prover.o ../build/comm.o -ldl -o ../libh2w_$name.so
echo "built $(realpath ../libh2w_$name.so)"
