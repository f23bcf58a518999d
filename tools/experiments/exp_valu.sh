cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_valu
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/pmc_valu -o r -- python bench.py --proofs random --no-cpu-baseline --streams 1 --calib 0 --steps 4 --warmup 2 > gpurun_out/pmc_valu.log 2>&1; tail -1 gpurun_out/pmc_valu.log | cut -c1-200
