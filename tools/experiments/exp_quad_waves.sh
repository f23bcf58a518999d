cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -f gpurun_out/exp_qw.txt
run() { timeout -k 10 200 python bench.py --proofs random --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$LABEL', '$*', 'ms_per_step %.3f G %.1f'%(d['ms_per_step'], d['value']/1e9), 'iso', {k:round(v,2) for k,v in d['kernel_ms_isolated'].items()}, 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()})" >> gpurun_out/exp_qw.txt || echo "FAILED $LABEL $*" >> gpurun_out/exp_qw.txt; }
for v in w2q64 w2q32 w3q32; do
  export H2W_LIB=$GRAFT_REPO_ROOT/halo2-plonky2-verifier_amd/libh2w_$v.so
  timeout -k 10 300 python -m pytest tests/test_gpu_batch.py -m gpu -x -q --timeout 150 -k "small_shapes or config3" 2>&1 | tail -1 >> gpurun_out/exp_qw.txt
  LABEL=$v run
  LABEL=$v run --streams 8
done
cat gpurun_out/exp_qw.txt
