# Round-3 evidence, run on the GPU box:  gpurun --timeout 1200 -- 'bash tools/collect_evidence.sh [part]'   (parts: tests bench prof pmc shard; default all)
# Summaries land in gpurun_out/r03_*; the ones to keep are copied into profiles/ by hand.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
part=${1:-all}
brief() { python3 -c "
import json,sys;d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]);r=d['roofline']
print(sys.argv[1], 'G', round(d['value']/1e9,1), 'period', round(d['expand_schedule_timed_region']['expand_end_to_next_expand_end_ms']['avg'],2) if d.get('expand_schedule_timed_region') else None, r['kernel'], round(r['frac'],3), {k:(round(v['ms_isolated'],2), round(v.get('frac',0),3)) for k,v in r['kernels'].items()}, 'whole', round(r['whole_job_frac'],3), d.get('latency'))" $1; }
if [ $part = all ] || [ $part = tests ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r03_gpu_tests.txt 2>&1; tail -3 gpurun_out/r03_gpu_tests.txt
fi
if [ $part = all ] || [ $part = bench ]; then
  timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err && brief gpurun_out/r03_bench_default.json
  N="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
  timeout -k 10 300 python bench.py $N --hash gl > gpurun_out/r03_bench_gl.json 2>/dev/null && brief gpurun_out/r03_bench_gl.json
  timeout -k 10 300 python bench.py $N --chain-passes 2 > gpurun_out/r03_bench_two_pass_paths.json 2>/dev/null && brief gpurun_out/r03_bench_two_pass_paths.json
  timeout -k 10 300 python bench.py $N --layout columns > gpurun_out/r03_bench_columns.json 2>/dev/null && brief gpurun_out/r03_bench_columns.json
  for c in cfg1 cfg2 cfg5; do timeout -k 10 300 python bench.py $N --config $c > gpurun_out/r03_bench_$c.json 2>/dev/null && brief gpurun_out/r03_bench_$c.json; done
  timeout -k 10 300 python bench.py --gpus 2 --share-gpu0 --backend gloo --steps 4 --warmup 1 --batch 8 --no-cpu-baseline --calib 1 > gpurun_out/r03_bench_2rank_shared_gpu_weak.json 2>/dev/null; tail -c 300 gpurun_out/r03_bench_2rank_shared_gpu_weak.json
  timeout -k 10 300 python bench.py --gpus 2 --share-gpu0 --backend gloo --config cfg5 --steps 4 --warmup 1 --batch 8 --no-cpu-baseline --calib 1 > gpurun_out/r03_bench_2rank_shared_gpu_cfg5_strong.json 2>/dev/null; tail -c 300 gpurun_out/r03_bench_2rank_shared_gpu_cfg5_strong.json
  for b in 1 64; do for ps in 1 2; do timeout -k 10 200 python tools/launch_timing.py --batch $b --passes $ps; done; done 2>/dev/null | grep config > gpurun_out/r03_launch_timing.jsonl
  timeout -k 10 200 python tools/launch_timing.py --batch 4 --hash gl 2>/dev/null | grep config >> gpurun_out/r03_launch_timing.jsonl; timeout -k 10 200 python tools/launch_timing.py --batch 1 --hash gl 2>/dev/null | grep config >> gpurun_out/r03_launch_timing.jsonl
  cat gpurun_out/r03_launch_timing.jsonl
fi
if [ $part = all ] || [ $part = prof ]; then
  for tag in default s1_nofork s1_nofork_two_pass; do
    rm -rf gpurun_out/prof_$tag
    case $tag in
      default) ARGS="--no-cpu-baseline --no-extras --calib 0 --steps 4 --warmup 2";;
      s1_nofork) ARGS="--no-cpu-baseline --no-extras --streams 1 --no-fork --calib 0 --steps 2 --warmup 1 --launches-per-step 3";;
      *) ARGS="--no-cpu-baseline --no-extras --streams 1 --no-fork --calib 0 --steps 2 --warmup 1 --launches-per-step 3 --chain-passes 2";;
    esac
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o r --output-format csv -- python3 bench.py $ARGS > gpurun_out/prof_$tag.log 2>&1
    cp gpurun_out/prof_$tag/r_kernel_stats.csv gpurun_out/r03_bench_${tag}_cfg3_bn254_b64_kernel_stats.csv && cut -c1-110 gpurun_out/r03_bench_${tag}_cfg3_bn254_b64_kernel_stats.csv | head -9
    rm -f gpurun_out/prof_$tag/r_kernel_trace.csv
  done
fi
if [ $part = all ] || [ $part = pmc ]; then
  for ps in 1 2; do
    for ctr in WRITE_SIZE FETCH_SIZE; do
      rm -rf gpurun_out/pmc_${ctr}_$ps
      timeout -k 10 300 rocprofv3 --pmc $ctr -d gpurun_out/pmc_${ctr}_$ps -o r --output-format csv -- python3 tools/launch_timing.py --batch 64 --reps 2 --passes $ps > gpurun_out/pmc_${ctr}_$ps.log 2>&1
    done
  done
  python3 - <<'PY' | tee gpurun_out/r03_pmc_traffic_cfg3_bn254_b64.json
import csv, collections, glob, json
out = {}
for ps in (1, 2):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for ctr in ("WRITE_SIZE", "FETCH_SIZE"):
        for f in glob.glob(f"gpurun_out/pmc_{ctr}_{ps}/*counter_collection.csv"):
            for row in csv.DictReader(open(f)):
                acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out[f"merkle_path_passes_{ps}"] = {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"dispatches": max(len(v) for v in cs.values())} for k, cs in acc.items() if "h2w" in k}
out["units"] = "rocprofv3 WRITE_SIZE / FETCH_SIZE are in KiB-like units of 1024 B on this box? see DESIGN.md: compare WRITE_SIZE of expand_fast with its 35.97 GB algorithmic"
print(json.dumps(out, indent=1))
PY
fi
if [ $part = all ] || [ $part = shard ]; then
  timeout -k 10 300 python tools/shard_timing.py --config cfg5 --batch 16 --world 8 > gpurun_out/r03_shard_timing_cfg5_b16_w8.json 2>/dev/null; python3 -c "
import json;d=json.load(open('gpurun_out/r03_shard_timing_cfg5_b16_w8.json'));print(d['unsharded']['ms']);print(d['ranks'][0]['ms']);print(d['slowest_rank_over_unsharded'])"
fi
