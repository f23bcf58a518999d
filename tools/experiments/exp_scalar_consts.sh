cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_batch.py -m gpu -x -q --timeout 150 -k "small_shapes or valid_fri or config3 or config2 or published or columns or sharding or lookup_bits" 2>&1 | tee gpurun_out/gpu_sc.log | tail -3 || exit 1
rm -f gpurun_out/exp_sc.txt
run() { timeout -k 10 200 python bench.py --proofs random --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$LABEL', '$*', 'ms_per_step %.3f G %.1f'%(d['ms_per_step'], d['value']/1e9), 'iso', {k:round(v,2) for k,v in d['kernel_ms_isolated'].items()}, 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()})" >> gpurun_out/exp_sc.txt || exit 1; }
LABEL=scalar run
LABEL=scalar run --streams 8
LABEL=scalar H2W_EXPAND_VARIANT=3 run
LABEL=scalar H2W_EXPAND_VARIANT=3 run --streams 8
cat gpurun_out/exp_sc.txt
