cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -f gpurun_out/exp_qb.txt
run() { timeout -k 10 200 python bench.py --proofs random --no-cpu-baseline --calib 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$LABEL', '$*', 'ms_per_step %.3f G %.1f'%(d['ms_per_step'], d['value']/1e9), {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()})" >> gpurun_out/exp_qb.txt || exit 1; }
LABEL=base run
LABEL=base H2W_QUAD_PAD_LDS=8192 run
export H2W_LIB=$GRAFT_REPO_ROOT/halo2-plonky2-verifier_amd/libh2w_qb128.so
LABEL=qb128 run
LABEL=qb128 run --streams 8
LABEL=qb128 run --streams 4
LABEL=qb128 run --hash gl
cat gpurun_out/exp_qb.txt
