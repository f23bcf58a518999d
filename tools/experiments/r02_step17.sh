set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_batch.py -x -q -m gpu -k "scheduling or two_plans or separate_emit" 2>&1 | tail -3
for tag in bn gl; do
rm -rf gpurun_out/prof_ov_$tag
if [ $tag = gl ]; then H="--hash gl"; else H=""; fi
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/prof_ov_$tag -o r -- python3 bench.py --no-cpu-baseline --calib 0 --steps 5 --warmup 2 $H > gpurun_out/prof_ov_$tag.log 2>&1
tail -1 gpurun_out/prof_ov_$tag.log | cut -c1-120
python3 tools/rocpd_summary.py overlap gpurun_out/prof_ov_$tag/r_results.db gpurun_out/r02_overlap_$tag.txt
cat gpurun_out/r02_overlap_$tag.txt
done
find gpurun_out -name "r_results.db" -size +8M -delete
