cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_gl
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_gl -o r -- python bench.py --hash gl --no-cpu-baseline --streams 1 --calib 0 --proofs random > gpurun_out/prof_gl.log 2>&1; tail -1 gpurun_out/prof_gl.log | cut -c1-200
find gpurun_out/prof_gl -name "*kernel_trace*" -delete
