# rocprofv3 evidence for the synthetic-proof generator (csrc/prover.hip): kernel stats for both hash modes + VALU counters of the tree hashing
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_prover_gl gpurun_out/prof_prover_bn gpurun_out/pmc_prover gpurun_out/bench_prover.log
for c in "cfg3 gl 1" "cfg3 gl 8" "cfg3 bn254 8" "cfg2 gl 32" "cfg2 bn254 32"; do set -- $c; timeout -k 10 200 python tools/bench_prover.py --config $1 --hash $2 --batch $3 --reps 3 2>&1 | grep tool >> gpurun_out/bench_prover.log; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_prover_gl -o r -- python tools/bench_prover.py --config cfg3 --hash gl --batch 8 --reps 2 > gpurun_out/prof_prover_gl.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_prover_bn -o r -- python tools/bench_prover.py --config cfg2 --hash bn254 --batch 32 --reps 2 > gpurun_out/prof_prover_bn.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/pmc_prover -o r -- python tools/bench_prover.py --config cfg3 --hash gl --batch 8 --reps 1 > gpurun_out/pmc_prover.log 2>&1
find gpurun_out/prof_prover_gl gpurun_out/prof_prover_bn -name "*kernel_trace*" -delete
cut -c1-330 gpurun_out/bench_prover.log
