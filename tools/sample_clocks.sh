#!/bin/bash
# usage: sample_clocks.sh <label> <cmd...> : runs the command while sampling rocm-smi clocks/power every 0.25 s
label=$1; shift
"$@" > gpurun_out/clk_$label.log 2>&1 &
pid=$!
: > gpurun_out/clk_$label.txt
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Average Graphics Package Power|Current Socket" | tr -s ' ' | tr '\n' '|' >> gpurun_out/clk_$label.txt; echo >> gpurun_out/clk_$label.txt
  sleep 0.25
done
wait $pid
