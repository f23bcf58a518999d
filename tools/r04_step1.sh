# round 4, step 1: the sharded schedule at the bench's launch size (VERDICT r03 task 1).  gpurun --timeout 1100 -- 'bash tools/r04_step1.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_batch.py -x -q -m gpu -k "shard or packed" > gpurun_out/r04_t1.txt 2>&1
rc=$?; tail -5 gpurun_out/r04_t1.txt; [ $rc = 0 ] || exit $rc
N="--steps 10 --warmup 3 --no-cpu-baseline --no-extras"
timeout -k 10 280 python bench.py --config cfg5 $N > gpurun_out/r04_bench_cfg5_unsharded.json 2> gpurun_out/r04_bench_cfg5_unsharded.err || { tail -c 800 gpurun_out/r04_bench_cfg5_unsharded.err; exit 1; }
timeout -k 10 280 python bench.py --config cfg5 --emulate-rank 0 --world 8 $N > gpurun_out/r04_rank0_of_8_cfg5.json 2> gpurun_out/r04_rank0_of_8_cfg5.err || { tail -c 800 gpurun_out/r04_rank0_of_8_cfg5.err; exit 1; }
timeout -k 10 280 python bench.py --config cfg5 --emulate-rank 3 --world 8 $N > gpurun_out/r04_rank3_of_8_cfg5.json 2> gpurun_out/r04_rank3_of_8_cfg5.err || { tail -c 800 gpurun_out/r04_rank3_of_8_cfg5.err; exit 1; }
python3 - <<'PY'
import json
for f in ("r04_bench_cfg5_unsharded", "r04_rank0_of_8_cfg5", "r04_rank3_of_8_cfg5"):
    d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"] / 1e9, 1), "G cells/s |", d["config"]["step"], "|", {k: round(v, 2) for k, v in d["kernel_ms_isolated"].items()})
PY
