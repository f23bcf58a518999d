cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -f gpurun_out/exp_hq.txt
run() { timeout -k 10 250 python bench.py --proofs random --no-cpu-baseline --calib 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('Q=$GPU_MAX_HW_QUEUES $*', 'G %.1f ms_per_step %.3f'%(d['value']/1e9, d['ms_per_step']), 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()})" >> gpurun_out/exp_hq.txt || echo "FAILED Q=$GPU_MAX_HW_QUEUES $*" >> gpurun_out/exp_hq.txt; }
export GPU_MAX_HW_QUEUES=16
run --streams 8
run --streams 12
run --batch 16 --streams 12
export GPU_MAX_HW_QUEUES=24
run --batch 16 --streams 16
run --streams 6
export GPU_MAX_HW_QUEUES=4
run --streams 6
cat gpurun_out/exp_hq.txt
