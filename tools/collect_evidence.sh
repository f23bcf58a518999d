# Round-4 evidence, run on the GPU box in parts (each well under the 20-minute call limit):
#   gpurun --timeout 1150 -- 'bash tools/collect_evidence.sh tests'      the whole -m gpu suite
#   gpurun --timeout 1150 -- 'bash tools/collect_evidence.sh bench'      the driver's line + the other configs / modes
#   gpurun --timeout 1150 -- 'bash tools/collect_evidence.sh prof'       rocprofv3 --kernel-trace --stats summaries
#   gpurun --timeout 1150 -- 'bash tools/collect_evidence.sh pmc'        rocprofv3 --pmc passes (traffic, VALU issue)
# Summaries land in gpurun_out/r04_*; the ones to keep are copied into profiles/ by hand.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
part=${1:-bench}
brief() { python3 -c "
import json,sys;d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]);r=d['roofline']
print(sys.argv[1], 'G', round(d['value']/1e9,1), 'period', round(d['expand_schedule_timed_region']['expand_end_to_next_expand_end_ms']['avg'],2) if d.get('expand_schedule_timed_region') else None, r['kernel'], round(r['frac'],3), {k:(round(v['ms_isolated'],2), round(v.get('frac',0),3)) for k,v in r['kernels'].items()}, 'whole', round(r['whole_job_frac'],3), d.get('latency'))" $1; }
if [ $part = tests ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r04_gpu_tests.txt 2>&1; rc=$?; tail -3 gpurun_out/r04_gpu_tests.txt; exit $rc
fi
if [ $part = bench ]; then
  timeout -k 10 420 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err && brief gpurun_out/r04_bench_default.json || { tail -c 800 gpurun_out/r04_bench_default.err; exit 1; }
  N="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
  timeout -k 10 300 python bench.py $N --hash gl > gpurun_out/r04_bench_gl.json 2>/dev/null && brief gpurun_out/r04_bench_gl.json || exit 1
  timeout -k 10 300 python bench.py $N --layout columns > gpurun_out/r04_bench_columns.json 2>/dev/null && brief gpurun_out/r04_bench_columns.json || exit 1
  for c in cfg1 cfg2 cfg5; do timeout -k 10 300 python bench.py $N --config $c > gpurun_out/r04_bench_$c.json 2>/dev/null && brief gpurun_out/r04_bench_$c.json || exit 1; done
  timeout -k 10 300 python bench.py $N --config cfg5 --emulate-rank 0 --world 8 --advice-cap-gb 262 > gpurun_out/r04_rank0_of_8_cfg5_s4.json 2>/dev/null && brief gpurun_out/r04_rank0_of_8_cfg5_s4.json || exit 1
  timeout -k 10 300 python bench.py --gpus 2 --share-gpu0 --backend gloo --config cfg5 --steps 4 --warmup 1 --batch 4 --no-cpu-baseline --calib 1 > gpurun_out/r04_bench_2rank_shared_gpu_cfg5_strong.json 2>/dev/null; tail -c 300 gpurun_out/r04_bench_2rank_shared_gpu_cfg5_strong.json
fi
if [ $part = prof ]; then
  for tag in default s1_nofork s1_gl; do
    rm -rf gpurun_out/prof_$tag
    case $tag in
      default) ARGS="--no-cpu-baseline --no-extras --calib 0 --steps 4 --warmup 2"; name=default_cfg3_bn254_b64;;
      s1_nofork) ARGS="--no-cpu-baseline --no-extras --streams 1 --no-fork --calib 0 --steps 2 --warmup 1 --launches-per-step 3"; name=s1_nofork_cfg3_bn254_b64;;
      *) ARGS="--no-cpu-baseline --no-extras --hash gl --streams 1 --calib 0 --steps 2 --warmup 1 --launches-per-step 3"; name=gl_s1_cfg3_gl_b4;;
    esac
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o r --output-format csv -- python3 bench.py $ARGS > gpurun_out/prof_$tag.log 2>&1 || { tail -5 gpurun_out/prof_$tag.log; exit 1; }
    cp gpurun_out/prof_$tag/r_kernel_stats.csv gpurun_out/r04_bench_${name}_kernel_stats.csv && cut -c1-110 gpurun_out/r04_bench_${name}_kernel_stats.csv | head -8
    rm -f gpurun_out/prof_$tag/r_kernel_trace.csv
  done
  rm -rf gpurun_out/prof_lat
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_lat -o r --output-format csv -- python3 tools/launch_timing.py --batch 1 --reps 4 --passes 2 > gpurun_out/prof_lat.log 2>&1 && cp gpurun_out/prof_lat/r_kernel_stats.csv gpurun_out/r04_one_proof_kernel_stats.csv && cut -c1-110 gpurun_out/r04_one_proof_kernel_stats.csv | head -12
  rm -f gpurun_out/prof_lat/r_kernel_trace.csv
fi
if [ $part = pmc ]; then
  bash tools/r04_pmc.sh || exit 1
  # the Goldilocks-caps launch (its dominant kernel: expand_fast): traffic of one launch of 4 proofs
  for ctr in WRITE_SIZE FETCH_SIZE; do
    rm -rf gpurun_out/pmc4gl_$ctr
    timeout -k 10 300 rocprofv3 --pmc $ctr -d gpurun_out/pmc4gl_$ctr -o r --output-format csv -- python3 tools/launch_timing.py --hash gl --batch 4 --reps 2 > gpurun_out/pmc4gl_$ctr.log 2>&1 || exit 1
  done
  python3 - <<'PY' | tee gpurun_out/r04_pmc_cfg3_gl_b4.json
import csv, collections, glob, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for ctr in ("WRITE_SIZE", "FETCH_SIZE"):
    for f in glob.glob(f"gpurun_out/pmc4gl_{ctr}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"dispatches": max(len(v) for v in cs.values())} for k, cs in acc.items() if "h2w" in k}
print(json.dumps({"launch": "tools/launch_timing.py --hash gl --batch 4 (4 proofs x 450,165,266 record cells x 32 B = 57.62 GB algorithmic for expand_fast)", "kernels": out}, indent=1))
PY
fi
