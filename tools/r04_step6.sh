# round 4, step 6: the replay interpreter with its LDS ring: parity, then where a launch spends its time.  gpurun --timeout 1100 -- 'bash tools/r04_step6.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_replay.py -x -q -m gpu > gpurun_out/r04_t6.txt 2>&1
rc=$?; tail -8 gpurun_out/r04_t6.txt; [ $rc = 0 ] || exit $rc
bash tools/r04_step5.sh 64 && timeout -k 10 300 python tools/replay_timing.py --batch 128 --reps 2 2>&1 | tail -1 && timeout -k 10 300 python tools/replay_timing.py --hash gl --batch 4 --reps 2 2>&1 | tail -1
