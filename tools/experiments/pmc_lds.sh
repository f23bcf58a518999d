cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES"; do
  n=$(echo $set | cut -d' ' -f3)
  timeout -k 10 240 rocprofv3 --pmc $set -d $R/gpurun_out/pmc_lds_$n -o r --output-format csv -- python3 $R/tools/launch_timing.py --batch 64 --passes 2 --reps 1 > $R/gpurun_out/pmc_lds_$n.log 2>&1 || { tail -5 $R/gpurun_out/pmc_lds_$n.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
for f in sorted(glob.glob(R+"/gpurun_out/pmc_lds_*/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"]
        for key in ("k_merkle_bn_emit","k_merkle_bn_values","expand_fast"):
            if key in k: acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for key,v in acc.items():
        print(key, {c: "%.3e"%(sum(x)/len(x)) for c,x in v.items()})
PY
