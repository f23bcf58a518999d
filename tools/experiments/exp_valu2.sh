cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 3 0; do
rm -rf gpurun_out/pmc_valu_v$v
H2W_EXPAND_VARIANT=$v timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT -d gpurun_out/pmc_valu_v$v -o r -- python bench.py --proofs random --no-cpu-baseline --streams 1 --calib 0 --steps 4 --warmup 2 > gpurun_out/pmc_valu_v$v.log 2>&1; tail -1 gpurun_out/pmc_valu_v$v.log | cut -c1-120
done
