cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -f gpurun_out/exp_ms.txt
run() { timeout -k 10 250 python bench.py --proofs random --no-cpu-baseline --calib 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$*', 'S=%d'%d['config']['batches_in_flight'], 'G %.1f ms_per_step %.3f'%(d['value']/1e9, d['ms_per_step']), 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()})" >> gpurun_out/exp_ms.txt || echo "FAILED $*" >> gpurun_out/exp_ms.txt; }
export GPU_MAX_HW_QUEUES=16
run --streams 6
run --streams 7 --advice-cap-gb 250
run --streams 8 --advice-cap-gb 250
run --batch 24 --streams 10 --advice-cap-gb 250
run --batch 28 --streams 9 --advice-cap-gb 250
run --hash gl --streams 4 --advice-cap-gb 250
cat gpurun_out/exp_ms.txt
