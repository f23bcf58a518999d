#!/bin/bash
# what one wavefront pays per instruction (tools/ubench/ubench_wave1.hip)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value tools/ubench/ubench_wave1.hip -o gpurun_out/ubench_wave1 2>/dev/null
timeout -k 10 120 gpurun_out/ubench_wave1 | tee gpurun_out/r04_ubench_wave1.txt
