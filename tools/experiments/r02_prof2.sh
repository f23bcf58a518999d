set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_def
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_def -o r -- python3 bench.py --no-cpu-baseline --calib 0 --steps 4 --warmup 2 --proofs random > gpurun_out/prof_def.log 2>&1; grep -v simple_timer gpurun_out/prof_def.log | tail -1 | cut -c1-200
ls -la gpurun_out/prof_def/
