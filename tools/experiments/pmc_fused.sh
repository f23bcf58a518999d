# PMC pass over one isolated one-pass launch: instruction and wait counters of k_merkle_bn_fused (averages per dispatch)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; B=${1:-64}
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_CBRANCH_NOT_TAKEN SQ_WAIT_IFETCH SQ_IFETCH SQ_INSTS_SENDMSG SQ_INSTS_EXP_GDS"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout -k 10 240 rocprofv3 --pmc $set -d $R/gpurun_out/pmc_fused_$n -o r --output-format csv -- python3 $R/tools/launch_timing.py --batch $B --passes 1 --reps 1 > $R/gpurun_out/pmc_fused_$n.log 2>&1 || { tail -5 $R/gpurun_out/pmc_fused_$n.log; echo "(set $n failed)"; }
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
for f in sorted(glob.glob(R+"/gpurun_out/pmc_fused_*/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"]
        for key in ("k_merkle_bn_fused","k_strands","k_prologue_values"):
            if key in k: acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for key,v in acc.items():
        print(key, {c: "%.4e"%(sum(x)/len(x)) for c,x in v.items()})
PY
