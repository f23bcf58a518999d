set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --calib 0 $1 > gpurun_out/r02_bench_s.log 2>&1 || { tail -20 gpurun_out/r02_bench_s.log; exit 1; }
python3 -c "
import json;d=json.loads(open('gpurun_out/r02_bench_s.log').read().strip().splitlines()[-1]);print('$1', 'G', round(d['value']/1e9,1), 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()})"; }
run ""
run "--streams 3"
run "--streams 5 --batch 48"
run "--streams 6 --batch 40"
run "--streams 6 --batch 32"
run "--streams 5 --batch 40"
run "--streams 4 --batch 56"
run ""
run "--hash gl --streams 5 --batch 3"
run "--hash gl"
