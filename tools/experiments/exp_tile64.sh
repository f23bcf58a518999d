cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -f gpurun_out/exp_t64.txt
for v in 64 364; do
  H2W_EXPAND_VARIANT=$v timeout -k 10 400 python -m pytest tests/test_gpu_batch.py -m gpu -x -q --timeout 150 -k "small_shapes or other_lookup_bits or valid_fri or sharding or config1 or config3 or shape_edge or expand_records" 2>&1 | tail -1 >> gpurun_out/exp_t64.txt
done
run() { timeout -k 10 250 python bench.py --proofs random --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('V=$H2W_EXPAND_VARIANT $*', 'G %.1f ms_per_step %.3f'%(d['value']/1e9, d['ms_per_step']), 'iso_expand %.2f b2b %.0f'%(d['kernel_ms_isolated']['expand'], r['achieved_back_to_back'] or 0))" >> gpurun_out/exp_t64.txt || echo "FAILED V=$H2W_EXPAND_VARIANT $*" >> gpurun_out/exp_t64.txt; }
for v in 364 64 3 0 364 64; do export H2W_EXPAND_VARIANT=$v; run; run --hash gl; done
cat gpurun_out/exp_t64.txt
