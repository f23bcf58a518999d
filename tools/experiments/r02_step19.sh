set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export H2W_LIB=$GRAFT_REPO_ROOT/halo2-plonky2-verifier_amd/libh2w_dbg.so
for gx in 4 8 16 32 64 128; do
H2W_DBG_EXPAND_GX=$gx timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --calib 3 --proofs random > gpurun_out/r02_bench_s.log 2>&1 || { tail -20 gpurun_out/r02_bench_s.log; exit 1; }
python3 -c "
import json;d=json.loads(open('gpurun_out/r02_bench_s.log').read().strip().splitlines()[-1]);print('gx $gx', 'G', round(d['value']/1e9,1), 'isolated expand', round(d['kernel_ms_isolated']['expand'],2), 'b2b', round(d['roofline']['expand_back_to_back_GBps']))"
done
for gx in 128 512 2048; do
H2W_DBG_EXPAND_GX=$gx timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --calib 3 --proofs random --hash gl > gpurun_out/r02_bench_s.log 2>&1 || { tail -20 gpurun_out/r02_bench_s.log; exit 1; }
python3 -c "
import json;d=json.loads(open('gpurun_out/r02_bench_s.log').read().strip().splitlines()[-1]);print('gl gx $gx', 'G', round(d['value']/1e9,1), 'isolated expand', round(d['kernel_ms_isolated']['expand'],2), 'b2b', round(d['roofline']['expand_back_to_back_GBps']))"
done
