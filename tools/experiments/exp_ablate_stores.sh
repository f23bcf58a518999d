cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -f gpurun_out/exp_abl.txt
run() { timeout -k 10 200 python bench.py --proofs random --no-cpu-baseline --calib 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$LABEL', '$*', 'ms_per_step %.3f'%(d['ms_per_step']), 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()})" >> gpurun_out/exp_abl.txt || exit 1; }
LABEL=base run
for v in nf ens nfens; do export H2W_LIB=$GRAFT_REPO_ROOT/halo2-plonky2-verifier_amd/libh2w_$v.so; LABEL=$v run; done
cat gpurun_out/exp_abl.txt
