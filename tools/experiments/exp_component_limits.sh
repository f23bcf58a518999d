cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_prover.py -m gpu -x -q > gpurun_out/gpu_prover.log 2>&1; tail -2 gpurun_out/gpu_prover.log
rm -f gpurun_out/bench_prover.log; for c in "cfg3 gl 8" "cfg2 gl 32" "cfg2 bn254 32"; do set -- $c; timeout -k 10 200 python tools/bench_prover.py --config $1 --hash $2 --batch $3 --reps 3 2>&1 | grep tool >> gpurun_out/bench_prover.log || exit 1; done; cut -c1-420 gpurun_out/bench_prover.log
export H2W_LIB=$GRAFT_REPO_ROOT/halo2-plonky2-verifier_amd/libh2w_dbg.so
rm -f gpurun_out/exp_skip.txt
for m in 0 5 3 6 4 2 1; do
  H2W_DBG_SKIP_KERNELS=$m timeout -k 10 200 python bench.py --proofs random --no-cpu-baseline --calib 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('skipmask $m ms_per_step %.3f'%d['ms_per_step'], 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()})" >> gpurun_out/exp_skip.txt || exit 1
done
for st in 2 3 4 8; do
  timeout -k 10 200 python bench.py --proofs random --no-cpu-baseline --calib 0 --streams $st 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('streams $st ms_per_step %.3f'%d['ms_per_step'], 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()})" >> gpurun_out/exp_skip.txt || exit 1
done
cat gpurun_out/exp_skip.txt
