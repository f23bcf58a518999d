#!/bin/bash
# round-4, last call: smoke(), the driver's bench line with the PMC traffic of the same sources on it
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
python3 - <<'P'
import json
d=json.loads(open('gpurun_out/r04_bench_default.json').read().strip().split('\n')[-1]); r=d['roofline']
print(d['value']/1e9, d['latency']['cfg3_bn254_batch1_ms'], {k:r.get(k) for k in ('frac','traffic','valu_issue','valu_issue_frac')}, d['secondary']['value']/1e9)
P
