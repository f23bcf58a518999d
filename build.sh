#!/bin/bash
# Builds libh2w.so (HIP, gfx950) in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")/halo2-plonky2-verifier_amd/csrc"
FLAGS="$H2W_EXTRA -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -Wno-unused-value -x hip"
mkdir -p ../build
pids=()
for f in expand.hip eager.cpp batch.hip glue.hip chipbatch.hip abi_backend.cpp prover.hip comm.cpp replay.hip; do
  o=../build/${f%.*}.o
  if [ ! -f $o ] || [ -n "$(find . -newer $o -name '*.h' -o -newer $o -name $f | head -1)" ] || [ ../../include/h2w.h -nt $o ]; then
    hipcc $FLAGS -c $f -o $o & pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC ../build/expand.o ../build/eager.o ../build/batch.o ../build/glue.o ../build/chipbatch.o ../build/abi_backend.o ../build/prover.o ../build/comm.o ../build/replay.o -ldl -o ../libh2w.so
echo "built $(realpath ../libh2w.so)"
