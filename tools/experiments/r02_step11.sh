set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in libh2w.so libh2w_noflush.so libh2w_noput.so; do
H2W_LIB=$GRAFT_REPO_ROOT/halo2-plonky2-verifier_amd/$lib timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --calib 3 --batch 32 --proofs random > gpurun_out/r02_bench_s.log 2>&1 || { tail -20 gpurun_out/r02_bench_s.log; exit 1; }
python3 -c "
import json;d=json.loads(open('gpurun_out/r02_bench_s.log').read().strip().splitlines()[-1]);print('$lib', 'G', round(d['value']/1e9,1), 'isolated', {k:round(v,2) for k,v in d['kernel_ms_isolated'].items()})"
done
