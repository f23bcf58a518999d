set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export H2W_LIB=$GRAFT_REPO_ROOT/halo2-plonky2-verifier_amd/libh2w_dbg.so
run() { env $1 timeout -k 10 300 python bench.py --no-cpu-baseline --proofs random --steps 10 --warmup 3 --calib 0 $2 > gpurun_out/r02_bench_s.log 2>&1 || { tail -20 gpurun_out/r02_bench_s.log; exit 1; }
python3 -c "
import json;d=json.loads(open('gpurun_out/r02_bench_s.log').read().strip().splitlines()[-1]);print('$1 $2', 'G', round(d['value']/1e9,1), 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()})"; }
run H2W_DBG_PRIO=0 "--hash gl"
run H2W_DBG_PRIO=2 "--hash gl"
run H2W_DBG_PRIO=3 "--hash gl"
run H2W_DBG_PRIO=2 "--hash gl --serial-expand 1"
run H2W_DBG_PRIO=3 "--hash gl --serial-expand 1"
run H2W_DBG_PRIO=0 ""
run H2W_DBG_PRIO=1 ""
run H2W_DBG_PRIO=4 ""
run H2W_DBG_PRIO=5 ""
