set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/prof_s1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_s1 -o r -- python3 bench.py --no-cpu-baseline --streams 1 --no-fork --calib 0 --steps 6 --warmup 3 --proofs random > gpurun_out/prof_s1.log 2>&1; tail -1 gpurun_out/prof_s1.log | cut -c1-200
python3 tools/rocpd_summary.py stats $(ls gpurun_out/prof_s1/*/*.db | head -1) gpurun_out/r02_s1_nofork_kernel_stats.csv
cut -c1-160 gpurun_out/r02_s1_nofork_kernel_stats.csv | head -8
find gpurun_out/prof_s1 -name "*.db" -delete
