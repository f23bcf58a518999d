#!/bin/bash
# Goldilocks-caps mode: proofs per launch x launches in flight under the same HBM budget
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for bs in "4 4" "5 3" "6 3" "8 2"; do set -- $bs
  timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras --hash gl --batch $1 --streams $2 > gpurun_out/r04_gl_b$1_s$2.json 2>/dev/null
  python3 -c "
import json;d=json.loads(open('gpurun_out/r04_gl_b$1_s$2.json').read().strip().splitlines()[-1])
print('batch',$1,'streams',$2,'G',round(d['value']/1e9,1),'frac',round(d['roofline']['whole_job_frac'],3))"
done
