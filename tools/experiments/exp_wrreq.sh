cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -o -E "TCC_EA0?_WR[A-Z_0-9]*|TCC_WRITE[A-Z_0-9]*|TCC_EA0?_WRREQ[A-Z_0-9]*" | sort -u | tr '\n' ' ' > gpurun_out/counters_wr.txt; cat gpurun_out/counters_wr.txt; echo
rm -rf gpurun_out/pmc_wrreq
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum -d gpurun_out/pmc_wrreq -o r -- python bench.py --proofs random --no-cpu-baseline --streams 1 --calib 0 --steps 4 --warmup 2 > gpurun_out/pmc_wrreq.log 2>&1; tail -2 gpurun_out/pmc_wrreq.log | cut -c1-200
ls gpurun_out/pmc_wrreq/
