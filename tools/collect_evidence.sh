set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; tail -2 gpurun_out/gpu_tests.log
timeout -k 10 300 python bench.py > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log
timeout -k 10 300 python bench.py --hash gl --no-cpu-baseline > gpurun_out/bench_gl.log 2>&1; tail -1 gpurun_out/bench_gl.log
rm -rf gpurun_out/prof_default gpurun_out/prof_s1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_default -o r -- python bench.py --no-cpu-baseline > gpurun_out/prof_default.log 2>&1; tail -1 gpurun_out/prof_default.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_s1 -o r -- python bench.py --no-cpu-baseline --streams 1 --calib 0 > gpurun_out/prof_s1.log 2>&1; tail -1 gpurun_out/prof_s1.log
rm -rf gpurun_out/pmc_w gpurun_out/pmc_f
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_w -o r -- python bench.py --no-cpu-baseline --streams 1 --calib 0 --steps 4 --warmup 2 > gpurun_out/pmc_w.log 2>&1; tail -1 gpurun_out/pmc_w.log
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_f -o r -- python bench.py --no-cpu-baseline --streams 1 --calib 0 --steps 4 --warmup 2 > gpurun_out/pmc_f.log 2>&1; tail -1 gpurun_out/pmc_f.log
find gpurun_out/prof_default gpurun_out/prof_s1 -name "*kernel_trace*" -delete
ls -la gpurun_out/pmc_w/* gpurun_out/prof_default/* | head
# N>1 logic rehearsal on one GPU (gloo, both ranks on cuda:0; random proof words): the driver runs the real RCCL scaling bench
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 2 --backend gloo --share-gpu0 --batch 8 --no-cpu-baseline > gpurun_out/bench_2rank.log 2>&1; tail -1 gpurun_out/bench_2rank.log | cut -c1-300
