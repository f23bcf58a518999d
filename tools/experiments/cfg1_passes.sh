#!/bin/bash
# cfg 1 (2^10 rows, 4 queries: short Merkle paths, many proofs per launch) in the pipeline with the Merkle paths in one pass and in two
cd "$(dirname "$0")/../.."
for cp in 1 2; do for b in 0; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --config cfg1 --chain-passes $cp > gpurun_out/r04_cfg1_p$cp.json 2>/dev/null
  python3 -c "
import json;d=json.loads(open('gpurun_out/r04_cfg1_p$cp.json').read().strip().splitlines()[-1]);r=d['roofline']
print('passes',$cp,'G',round(d['value']/1e9,1),d['config'].get('proofs_per_launch'),{k:round(v['ms_isolated'],2) for k,v in r['kernels'].items()})"
done; done
