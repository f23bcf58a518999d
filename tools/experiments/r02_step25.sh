cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/dbg_chipbatch.py > gpurun_out/r02_chip.log 2>&1; echo rc $?
grep -v "^  File" gpurun_out/r02_chip.log | head -40 | cut -c1-200
