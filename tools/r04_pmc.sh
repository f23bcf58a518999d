# round 4: PMC evidence for the bench line's roofline.traffic / roofline.valu_issue (bench.py reads profiles/r04_pmc_cfg3_bn254_b64.json and reports it only
# while the kernel sources are the ones the passes were taken on).  gpurun --timeout 1100 -- 'bash tools/r04_pmc.sh'
# Separate --pmc passes, nothing but --pmc on the rocprofv3 line (MI355X_MICROARCH.md); python3 directly behind `--`.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for ps in 1 2; do
  for ctr in WRITE_SIZE FETCH_SIZE SQ_INSTS_VALU SQ_WAVES; do
    rm -rf gpurun_out/pmc4_${ctr}_$ps
    timeout -k 10 300 rocprofv3 --pmc $ctr -d gpurun_out/pmc4_${ctr}_$ps -o r --output-format csv -- python3 tools/launch_timing.py --batch 64 --reps 2 --passes $ps > gpurun_out/pmc4_${ctr}_$ps.log 2>&1 || { tail -5 gpurun_out/pmc4_${ctr}_$ps.log; exit 1; }
  done
done
python3 - <<'PY' | tee gpurun_out/r04_pmc_cfg3_bn254_b64.json
import csv, collections, glob, json, subprocess, sys, os
sys.path.insert(0, os.getcwd())
import importlib.util
spec = importlib.util.spec_from_file_location("bench", "bench.py"); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
out = {"kernel_source_sha16": bench.kernel_source_sha16(), "commit": os.environ.get("H2W_COMMIT", "unknown")}
for ps in (1, 2):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for ctr in ("WRITE_SIZE", "FETCH_SIZE", "SQ_INSTS_VALU", "SQ_WAVES"):
        for f in glob.glob(f"gpurun_out/pmc4_{ctr}_{ps}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out[f"merkle_path_passes_{ps}"] = {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"dispatches": max(len(v) for v in cs.values())} for k, cs in acc.items() if "h2w" in k}
# the unit of WRITE_SIZE, confirmed on the kernel whose bytes are known: expand_fast writes 64 proofs x 17,564,958 record cells x 32 B = 35.973 GB per launch
try:
    ef = next(v for k, v in out["merkle_path_passes_1"].items() if "expand_fast" in k)
    out["unit_check"] = {"expand_fast_WRITE_SIZE": ef["WRITE_SIZE"], "algorithmic_bytes": 64 * 17564958 * 32, "bytes_per_unit": 64 * 17564958 * 32 / ef["WRITE_SIZE"],
                         "conclusion": "WRITE_SIZE counts KiB (1024 B) on this stack when bytes_per_unit is ~1024"}
except StopIteration:
    out["unit_check"] = None
print(json.dumps(out, indent=1))
PY
