cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export H2W_EXPAND_VARIANT=3
timeout -k 10 600 python -m pytest tests/test_gpu_batch.py tests/test_layout_metadata.py -m gpu -x -q --timeout 150 2>&1 | tee gpurun_out/gpu_exph.log | tail -15 || exit 1
rm -f gpurun_out/exp_exph.txt
run() { timeout -k 10 200 python bench.py --proofs random --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$LABEL', '$*', 'ms_per_step %.3f G %.1f'%(d['ms_per_step'], d['value']/1e9), 'iso', {k:round(v,2) for k,v in d['kernel_ms_isolated'].items()}, 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()})" >> gpurun_out/exp_exph.txt || exit 1; }
LABEL=hinted run
LABEL=hinted run --hash gl
unset H2W_EXPAND_VARIANT
LABEL=base run
LABEL=base run --hash gl
cat gpurun_out/exp_exph.txt
