cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -f gpurun_out/exp_prio.txt
run() { timeout -k 10 200 python bench.py --proofs random --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$LABEL', '$*', 'ms_per_step %.3f G %.1f'%(d['ms_per_step'], d['value']/1e9), 'iso', {k:round(v,2) for k,v in d['kernel_ms_isolated'].items()}, 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()})" >> gpurun_out/exp_prio.txt || exit 1; }
LABEL=prio run
LABEL=prio run --streams 8
LABEL=prio run --streams 4
LABEL=prio run --hash gl
LABEL=prio H2W_EXPAND_VARIANT=3 run
LABEL=prio H2W_EXPAND_VARIANT=3 run --hash gl
cat gpurun_out/exp_prio.txt
