# Round-2 evidence: GPU tests, the bench as the driver runs it, kernel traces, PMC passes.  Run on the GPU box:
#   gpurun -- 'bash tools/collect_evidence.sh'   -> summaries land in gpurun_out/r02_*, copy the ones to keep into profiles/
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S1="--no-cpu-baseline --streams 1 --no-fork --calib 0 --steps 2 --warmup 1 --launches-per-step 3 --batch 32"
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_default.log 2>&1; tail -1 gpurun_out/r02_bench_default.log | cut -c1-300
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --hash gl --no-cpu-baseline > gpurun_out/r02_bench_gl.log 2>&1; tail -1 gpurun_out/r02_bench_gl.log | cut -c1-200
for tag in default s1; do
  rm -rf gpurun_out/prof_$tag
  if [ $tag = default ]; then ARGS="--no-cpu-baseline --calib 0 --steps 4 --warmup 2"; else ARGS="$S1"; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o r -- python3 bench.py $ARGS > gpurun_out/prof_$tag.log 2>&1
  python3 tools/rocpd_summary.py stats gpurun_out/prof_$tag/r_results.db gpurun_out/r02_${tag}_kernel_stats.csv
  if [ $tag = default ]; then python3 tools/rocpd_summary.py overlap gpurun_out/prof_$tag/r_results.db gpurun_out/r02_overlap_under_rocprof.txt; fi
  cut -c1-120 gpurun_out/r02_${tag}_kernel_stats.csv | head -6
done
for ctr in WRITE_SIZE FETCH_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  tag=$(echo $ctr | cut -d' ' -f1)
  rm -rf gpurun_out/pmc_$tag
  timeout -k 10 300 rocprofv3 --pmc $ctr -d gpurun_out/pmc_$tag -o r -- python3 bench.py $S1 > gpurun_out/pmc_$tag.log 2>&1
done
python3 tools/rocpd_summary.py pmc gpurun_out/pmc_WRITE_SIZE/r_results.db gpurun_out/pmc_FETCH_SIZE/r_results.db gpurun_out/r02_pmc_traffic_cfg3_bn254_b32.json
python3 tools/rocpd_summary.py pmc gpurun_out/pmc_SQ_INSTS_VALU/r_results.db gpurun_out/r02_pmc_issue_cfg3_bn254_b32.json
find gpurun_out -name "r_results.db" -size +8M -delete
ls -la gpurun_out/r02_*
