set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log | cut -c1-400
timeout -k 10 300 python bench.py --hash gl --no-cpu-baseline > gpurun_out/bench_gl.log 2>&1; tail -1 gpurun_out/bench_gl.log | cut -c1-300
timeout -k 10 300 python bench.py --proofs random --no-cpu-baseline > gpurun_out/bench_random.log 2>&1; tail -1 gpurun_out/bench_random.log | cut -c1-300
rm -rf gpurun_out/prof_prover_gl gpurun_out/prof_prover_bn gpurun_out/pmc_prover
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_prover_gl -o r -- python tools/bench_prover.py --config cfg3 --hash gl --batch 8 --reps 2 > gpurun_out/prof_prover_gl.log 2>&1; tail -2 gpurun_out/prof_prover_gl.log | cut -c1-300
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_prover_bn -o r -- python tools/bench_prover.py --config cfg2 --hash bn254 --batch 32 --reps 2 > gpurun_out/prof_prover_bn.log 2>&1; tail -2 gpurun_out/prof_prover_bn.log | cut -c1-300
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/pmc_prover -o r -- python tools/bench_prover.py --config cfg3 --hash gl --batch 8 --reps 1 > gpurun_out/pmc_prover.log 2>&1; tail -2 gpurun_out/pmc_prover.log | cut -c1-300
find gpurun_out/prof_prover_gl gpurun_out/prof_prover_bn -name "*kernel_trace*" -delete
ls -la gpurun_out/prof_prover_gl/* gpurun_out/pmc_prover/* | head
