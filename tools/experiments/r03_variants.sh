run() { name=$1; shift; timeout -k 10 200 python bench.py --steps 12 --warmup 4 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['value']/1e9,1), {k: round(v,2) for k,v in d['kernel_ms_isolated'].items()}, {k: round(v,2) for k,v in d['kernel_ms_timed_region'].items()}, round(d['expand_schedule_timed_region']['expand_end_to_next_expand_end_ms']['avg'],2))"; }
run fused --chain-passes 1
run twophase --chain-passes 2
