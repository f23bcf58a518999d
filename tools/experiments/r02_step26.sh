set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_batch.py -x -q -m gpu -k "scheduling" 2>&1 | tail -2
timeout -k 10 300 python bench.py --gpus 2 --share-gpu0 --backend gloo --steps 4 --warmup 1 --batch 8 --no-cpu-baseline --calib 1 > gpurun_out/r02_bench_2rank_weak.log 2>&1 || { tail -20 gpurun_out/r02_bench_2rank_weak.log; exit 1; }
tail -1 gpurun_out/r02_bench_2rank_weak.log | cut -c1-400
timeout -k 10 300 python bench.py --gpus 2 --share-gpu0 --backend gloo --config cfg5 --steps 4 --warmup 1 --batch 4 --no-cpu-baseline --calib 1 > gpurun_out/r02_bench_2rank_cfg5.log 2>&1 || { tail -20 gpurun_out/r02_bench_2rank_cfg5.log; exit 1; }
tail -1 gpurun_out/r02_bench_2rank_cfg5.log | cut -c1-400
timeout -k 10 300 python bench.py --config cfg5 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_cfg5_1gpu.log 2>&1 || { tail -20 gpurun_out/r02_bench_cfg5_1gpu.log; exit 1; }
tail -1 gpurun_out/r02_bench_cfg5_1gpu.log | cut -c1-400
