cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export H2W_LIB=$GRAFT_REPO_ROOT/halo2-plonky2-verifier_amd/libh2w_q32.so
timeout -k 10 600 python -m pytest tests/test_gpu_batch.py -m gpu -x -q -k "small_shapes or valid_fri or config3 or published or columns or sharding" > gpurun_out/gpu_q32.log 2>&1; tail -3 gpurun_out/gpu_q32.log
rm -f gpurun_out/exp_q32.txt
run() { timeout -k 10 200 python bench.py --proofs random --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$LABEL', '$*', 'ms_per_step %.3f G %.1f'%(d['ms_per_step'], d['value']/1e9), 'iso', {k:round(v,2) for k,v in d['kernel_ms_isolated'].items()}, 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()})" >> gpurun_out/exp_q32.txt || exit 1; }
LABEL=q32 run
LABEL=q32 run --streams 8
LABEL=q32 run --streams 4
LABEL=q32 run --streams 12
unset H2W_LIB
LABEL=base run
cat gpurun_out/exp_q32.txt
