# round 4, step 4: the whole GPU suite, then the driver's bench line.  gpurun --timeout 1100 -- 'bash tools/r04_step4.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -x -q -m gpu > gpurun_out/r04_gpu_tests.txt 2>&1
rc=$?; tail -6 gpurun_out/r04_gpu_tests.txt; [ $rc = 0 ] || exit $rc
timeout -k 10 380 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err || { tail -c 1500 gpurun_out/r04_bench_default.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_bench_default.json").read().strip().splitlines()[-1])
print("value", round(d["value"] / 1e9, 1), "G | latency", d["latency"], "| secondary", round(d["secondary"]["value"] / 1e9, 1), d["secondary"]["frac_of_hbm_peak"], d["secondary"]["roofline"]["frac"])
print("eager", {k: v for k, v in d["eager"].items() if k != "workload"})
print("roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["valu_issue_frac"], d["roofline"]["traffic_source"])
PY
