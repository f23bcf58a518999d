#!/bin/bash
# Builds halo2-plonky2-verifier_amd/libh2w_<name>.so: the same sources with the extra compiler flags in H2W_EXTRA (compile-time knobs of the
# kernels: -DH2W_FAST_T=8, -DH2W_QUAD_BLOCK=512, ...; or whatever a local experiment patch reads).  Usage: H2W_EXTRA="..." build_debug_variant.sh [name=dbg].
# Select it with H2W_LIB=<path> (tools/experiments/variant.sh does both).  Never used by tests, smoke or the default bench.
set -e
name=${1:-dbg}
cd "$(dirname "$0")/../halo2-plonky2-verifier_amd/csrc"
FLAGS="$H2W_EXTRA -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -Wno-unused-value -x hip"
mkdir -p ../build_$name
hipcc $FLAGS -c batch.hip -o ../build_$name/batch.o
hipcc $FLAGS -c glue.hip -o ../build_$name/glue.o
hipcc $FLAGS -c expand.hip -o ../build_$name/expand.o
hipcc --offload-arch=gfx950 -shared -fPIC ../build_$name/expand.o ../build/eager.o ../build_$name/batch.o ../build_$name/glue.o ../build/chipbatch.o ../build/abi_backend.o ../build/prover.o ../build/comm.o ../build/replay.o -ldl -o ../libh2w_$name.so
echo "built $(realpath ../libh2w_$name.so)"
