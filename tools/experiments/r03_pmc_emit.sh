# PMC passes over isolated launches (tools/launch_timing.py, 64 proofs): issue / wait counters of the PoseidonBN254 emission kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L > gpurun_out/r03_counters.txt 2>&1
for v in pre nopre; do
  if [ $v = nopre ]; then export H2W_LIB=$PWD/halo2-plonky2-verifier_amd/libh2w_nopre.so; else unset H2W_LIB; fi
  i=0
  for ctr in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
    i=$((i+1)); rm -rf gpurun_out/pmc_${v}_$i
    timeout -k 10 300 rocprofv3 --pmc $ctr -d gpurun_out/pmc_${v}_$i -o r --output-format csv -- python3 tools/launch_timing.py --batch 64 --reps 2 > gpurun_out/pmc_${v}_$i.log 2>&1
    python3 - <<PY
import csv, collections, glob
f = glob.glob("gpurun_out/pmc_${v}_$i/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f)):
    acc[row["Kernel_Name"][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in acc:
    if "merkle_bn" in k or "expand" in k:
        print("$v", k, {c: round(sum(v) / len(v)) for c, v in acc[k].items()})
PY
  done
done
