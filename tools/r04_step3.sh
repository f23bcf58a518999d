# round 4, step 3: record / replay parity + the values pass after its trims.  gpurun --timeout 1100 -- 'bash tools/r04_step3.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_replay.py tests/test_gpu_batch.py -x -q -m gpu -k "replay or merkle_paths or values_pass" > gpurun_out/r04_t3.txt 2>&1
rc=$?; tail -12 gpurun_out/r04_t3.txt; [ $rc = 0 ] || exit $rc
: > gpurun_out/r04_values_forms.jsonl
for b in 1 4 8 12 16; do for f in 1 2; do
  timeout -k 10 200 python tools/launch_timing.py --batch $b --passes 2 --form $f 2>/dev/null | grep config >> gpurun_out/r04_values_forms.jsonl || exit 1
done; done
cat gpurun_out/r04_values_forms.jsonl
