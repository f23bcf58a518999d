cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -f gpurun_out/exp_tr.txt
run() { timeout -k 10 250 python bench.py --proofs random --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('V=$H2W_EXPAND_VARIANT $*', 'G %.1f ms_per_step %.3f'%(d['value']/1e9, d['ms_per_step']), 'iso_expand %.2f b2b %.0f'%(d['kernel_ms_isolated']['expand'], r['achieved_back_to_back'] or 0), 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()})" >> gpurun_out/exp_tr.txt || echo "FAILED V=$H2W_EXPAND_VARIANT $*" >> gpurun_out/exp_tr.txt; }
for v in 16 0 64; do export H2W_EXPAND_VARIANT=$v; run; run --hash gl; done
cat gpurun_out/exp_tr.txt
