set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --calib 0 --hash gl $1 > gpurun_out/r02_bench_s.log 2>&1 || { tail -20 gpurun_out/r02_bench_s.log; exit 1; }
python3 -c "
import json;d=json.loads(open('gpurun_out/r02_bench_s.log').read().strip().splitlines()[-1]);print('$1', 'G', round(d['value']/1e9,1), d['config']['step'][:60], 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()})"; }
run "--batch 4 --streams 4"
run "--batch 2 --streams 8"
run "--batch 2 --streams 6"
run "--batch 3 --streams 5"
run "--batch 1 --streams 8"
run "--batch 2 --streams 8 --serial-expand 1"
run "--batch 1 --streams 8 --serial-expand 1"
