cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -f gpurun_out/exp_bs2.txt
run() { timeout -k 10 250 python bench.py --proofs random --no-cpu-baseline --calib 0 "$@" 2>gpurun_out/exp_bs2.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$*', 'S=%d'%d['config']['batches_in_flight'], 'G %.1f ms_per_step %.3f'%(d['value']/1e9, d['ms_per_step']), 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()})" >> gpurun_out/exp_bs2.txt || { echo "FAILED $*" >> gpurun_out/exp_bs2.txt; tail -3 gpurun_out/exp_bs2.err >> gpurun_out/exp_bs2.txt; }; }
run --batch 40 --streams 5
run --batch 48 --streams 4
run --batch 36 --streams 5
run --batch 32 --streams 5
run --batch 32 --streams 6
run --batch 28 --streams 6
cat gpurun_out/exp_bs2.txt
