cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -f gpurun_out/exp_bs.txt
run() { timeout -k 10 250 python bench.py --proofs random --no-cpu-baseline --calib 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$*', 'G %.1f ms_per_step %.3f'%(d['value']/1e9, d['ms_per_step']), 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()})" >> gpurun_out/exp_bs.txt || echo "FAILED $*" >> gpurun_out/exp_bs.txt; }
run --batch 16 --streams 12
run --batch 8 --streams 24
run --batch 64 --streams 3
run --batch 16 --streams 6
run --hash gl --batch 2 --streams 6
run --hash gl --batch 1 --streams 12
run --hash gl --batch 4 --streams 3
cat gpurun_out/exp_bs.txt
