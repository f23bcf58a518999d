set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_tests.log 2>&1 || { tail -30 gpurun_out/r02_gpu_tests.log; exit 1; }
tail -3 gpurun_out/r02_gpu_tests.log
timeout -k 10 300 python bench.py --steps 24 --warmup 12 --no-cpu-baseline > gpurun_out/r02_bench_a.log 2>&1; tail -1 gpurun_out/r02_bench_a.log | cut -c1-400; python3 -c "
import json;d=json.loads(open('gpurun_out/r02_bench_a.log').read().strip().splitlines()[-1]);print(d['kernel_ms_isolated'],d['kernel_ms_timed_region'],d['roofline']['achieved'],d['roofline']['achieved_back_to_back'])"
timeout -k 10 300 python bench.py --steps 12 --warmup 6 --hash gl --no-cpu-baseline > gpurun_out/r02_bench_gl.log 2>&1; tail -1 gpurun_out/r02_bench_gl.log | cut -c1-300; python3 -c "
import json;d=json.loads(open('gpurun_out/r02_bench_gl.log').read().strip().splitlines()[-1]);print(d['kernel_ms_isolated'],d['kernel_ms_timed_region'],d['roofline']['achieved'],d['roofline']['achieved_back_to_back'])"
