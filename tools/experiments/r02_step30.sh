set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export H2W_LIB=$GRAFT_REPO_ROOT/halo2-plonky2-verifier_amd/libh2w_dbg.so
run() { env $1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --calib 3 $2 > gpurun_out/r02_bench_s.log 2>&1 || { tail -20 gpurun_out/r02_bench_s.log; exit 1; }
python3 -c "
import json;d=json.loads(open('gpurun_out/r02_bench_s.log').read().strip().splitlines()[-1]);print('$1 $2', 'G', round(d['value']/1e9,1), 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()}, 'isolated', {k:round(v,2) for k,v in (d.get('kernel_ms_isolated') or {}).items()}, 'b2b', round(d['roofline']['expand_back_to_back_GBps']))"; }
run H2W_DBG_NO_ROAM=0 ""
run H2W_DBG_NO_ROAM=1 ""
run H2W_DBG_NO_ROAM=0 ""
run H2W_DBG_NO_ROAM=1 ""
run H2W_DBG_NO_ROAM=0 "--hash gl"
run H2W_DBG_NO_ROAM=1 "--hash gl"
