set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --calib 2 $1 > gpurun_out/r02_bench_s.log 2>&1 || { tail -20 gpurun_out/r02_bench_s.log; exit 1; }
python3 -c "
import json;d=json.loads(open('gpurun_out/r02_bench_s.log').read().strip().splitlines()[-1]);print('$1', 'G', round(d['value']/1e9,1), 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()}, 'iso prologue', round(d['kernel_ms_isolated']['prologue'],2))"; }
run "--flat-prologue 0"
run "--flat-prologue 1"
run "--flat-prologue 0 --hash gl"
run "--flat-prologue 1 --hash gl"
run "--flat-prologue 1 --streams 3"
run "--flat-prologue 0 --streams 3"
