# round 4, step 2: the row-cooperative values pass (VERDICT r03 task 2).  gpurun --timeout 1100 -- 'bash tools/r04_step2.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_batch.py -x -q -m gpu -k "merkle_paths or values_pass" > gpurun_out/r04_t2.txt 2>&1
rc=$?; tail -8 gpurun_out/r04_t2.txt; [ $rc = 0 ] || exit $rc
: > gpurun_out/r04_values_forms.jsonl
for b in 1 4 8 16; do for f in 1 2; do
  timeout -k 10 200 python tools/launch_timing.py --batch $b --passes 2 --form $f 2>/dev/null | grep config >> gpurun_out/r04_values_forms.jsonl || exit 1
done; done
cat gpurun_out/r04_values_forms.jsonl
