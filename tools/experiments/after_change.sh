#!/bin/bash
# round-4: after the strand objects left scratch memory (chips.h SelArr): per-kernel times of one launch, the GPU suite, the default bench
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
O=gpurun_out/r04_after_selarr.txt
: > $O
for args in "--batch 1" "--batch 1 --hash gl" "--batch 64 --passes 1" "--batch 64 --passes 2" "--batch 4 --hash gl"; do
  timeout -k 10 300 python3 tools/launch_timing.py $args --reps 3 2>/dev/null | grep config >> $O
done
cat $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r04_t12.txt
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_after_selarr.json 2> gpurun_out/r04_bench_after_selarr.err
python3 - <<'P'
import json
d=json.loads(open('gpurun_out/r04_bench_after_selarr.json').read().strip().split('\n')[-1])
e=d.get('eager') or {}; print({k:d[k] for k in ('value','ms_per_step')}, d.get('latency'), d.get('kernel_ms_isolated'), d.get('secondary',{}).get('value'), {k:v for k,v in e.items() if 'seconds' in k or k=='footprint'}, (e.get('replay') or {}).get('value'))
P
