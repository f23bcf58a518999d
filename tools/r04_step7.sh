# round 4, step 7: what the replay root kernel does with its time (PMC, one wavefront).  gpurun --timeout 900 -- 'bash tools/r04_step7.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf gpurun_out/pmc_replay_$tag
  timeout -k 10 300 rocprofv3 --pmc $set -d gpurun_out/pmc_replay_$tag -o r --output-format csv -- python3 tools/replay_timing.py --batch 16 --reps 1 > gpurun_out/pmc_replay_$tag.log 2>&1 || { tail -5 gpurun_out/pmc_replay_$tag.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_replay_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        by = collections.OrderedDict()
        for r in rows:
            if "k_replay" not in r["Kernel_Name"]: continue
            by.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
        for k, v in list(by.items())[-3:]:
            print(k, {a: f"{b:.4g}" for a, b in v.items()})
PY
