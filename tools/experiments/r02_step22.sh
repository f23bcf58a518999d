set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { env $1 H2W_LIB=$GRAFT_REPO_ROOT/halo2-plonky2-verifier_amd/$2 timeout -k 10 300 python bench.py --no-cpu-baseline --proofs random $3 > gpurun_out/r02_bench_s.log 2>&1 || { tail -20 gpurun_out/r02_bench_s.log; exit 1; }
python3 -c "
import json;d=json.loads(open('gpurun_out/r02_bench_s.log').read().strip().splitlines()[-1]);print('$1 $2 $3', 'G', round(d['value']/1e9,1), 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()}, 'isolated', {k:round(v,2) for k,v in (d.get('kernel_ms_isolated') or {}).items()})"; }
ISO="--batch 32 --steps 2 --warmup 1 --calib 3 --pair-chains 1"
run H2W_DBG_PRIO=0 libh2w_dbg.so "$ISO"
run H2W_DBG_PRIO=16 libh2w_dbg.so "$ISO"
run H2W_DBG_PRIO=8 libh2w_dbg.so "$ISO"
run H2W_DBG_PRIO=0 libh2w_pnf.so "$ISO"
run H2W_DBG_PRIO=0 libh2w_peu2.so "$ISO"
run H2W_DBG_PRIO=0 libh2w_dbg.so "--steps 10 --warmup 3 --calib 0 --pair-chains 1"
run H2W_DBG_PRIO=16 libh2w_dbg.so "--steps 10 --warmup 3 --calib 0 --pair-chains 1"
