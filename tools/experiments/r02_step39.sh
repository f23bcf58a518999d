set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { env $1 H2W_LIB=$GRAFT_REPO_ROOT/halo2-plonky2-verifier_amd/$2 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --calib 2 $3 > gpurun_out/r02_bench_s.log 2>&1 || { tail -20 gpurun_out/r02_bench_s.log; exit 1; }
python3 -c "
import json;d=json.loads(open('gpurun_out/r02_bench_s.log').read().strip().splitlines()[-1]);print('$1 $2 $3', 'G', round(d['value']/1e9,1), 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()}, 'iso expand', round(d['kernel_ms_isolated']['expand'],2))"; }
run X=0 libh2w_dbg.so ""
run X=0 libh2w_k8.so ""
run X=0 libh2w_k2.so ""
run H2W_DBG_ROAM_BLOCKS=512 libh2w_dbg.so ""
run H2W_DBG_ROAM_BLOCKS=384 libh2w_dbg.so ""
run H2W_DBG_NO_ROAM=1 libh2w_dbg.so ""
run X=0 libh2w_dbg.so "--hash gl"
run X=0 libh2w_k8.so "--hash gl"
run H2W_DBG_ROAM_BLOCKS=256 libh2w_dbg.so "--hash gl"
