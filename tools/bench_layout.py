#!/usr/bin/env python3
"""Times the SURVEY §8(f)-row-1 kernels (flat advice -> FlexGate columns, lookup-advice columns) on cfg 3 / PoseidonBN254.
HBM-bound copies: algorithmic bytes = 32 B read + 32 B written per assigned cell (+ zero fill of the unassigned rows)."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
h2w = importlib.import_module("halo2-plonky2-verifier_amd"); api = importlib.import_module("halo2-plonky2-verifier_amd.api")
B, k = int(os.environ.get("B", 16)), int(os.environ.get("K", 22))
kh = h2w.PoseidonConsts()          # layout does not depend on the constants' values
sh = h2w.fibonacci_shape(20, 28)
plan = api.Plan(sh, kh)
t = time.time(); bp = plan.break_points(k); t_meta = time.time() - t
ncol, nl = len(bp) + 1, plan.num_lookup_columns(k)
advice = torch.empty(B * plan.num_cells * 32, dtype=torch.uint8, device="cuda").random_(0, 255)
cols = torch.empty(((B * ncol) << k) * 32, dtype=torch.uint8, device="cuda")
lk = torch.empty(((B * nl) << k) * 32, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def timed(f, n=5):
    f(); torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
ms_c = timed(lambda: plan.layout_columns(advice.data_ptr(), B, bp, k, cols.data_ptr(), st))
ms_l = timed(lambda: plan.layout_lookup_columns(advice.data_ptr(), B, k, lk.data_ptr(), stream=st))
ms_m = timed(lambda: plan.L.h2w_advice_to_montgomery(advice.data_ptr(), B * plan.num_cells, st))
nlk = len(plan.lookup_cells())
# fused: the generation kernels write the columns directly (h2w_fri_witness_batch_columns) vs flat generation + relayout
proofs = torch.zeros(B * plan.proof_words, dtype=torch.int64, device="cuda")
ws = torch.zeros(plan.workspace_bytes(B), dtype=torch.uint8, device="cuda")
ms_flat = timed(lambda: plan.run(proofs.data_ptr(), B, advice.data_ptr(), ws.data_ptr(), st))
ms_fused = timed(lambda: plan.run_columns(proofs.data_ptr(), B, bp, k, cols.data_ptr(), ws.data_ptr(), st))
print(json.dumps({"workload": f"cfg3 bn254, {B} proofs, k={k}", "metadata_replay_s": round(t_meta, 2), "columns": ncol, "lookup_columns": nl, "lookups_per_proof": nlk,
                  "layout_columns_ms": round(ms_c, 3), "layout_columns_GBps": round(B * (plan.num_cells * 32 + (ncol << k) * 32) / ms_c / 1e6, 1),
                  "generate_flat_ms": round(ms_flat, 3), "generate_flat_plus_relayout_ms": round(ms_flat + ms_c, 3), "generate_columns_fused_ms": round(ms_fused, 3),
                  "to_montgomery_ms": round(ms_m, 3), "to_montgomery_GBps": round(B * plan.num_cells * 64 / ms_m / 1e6, 1),
                  "layout_lookup_ms": round(ms_l, 3), "layout_lookup_GBps": round(B * (nlk * 32 + (nl << k) * 32) / ms_l / 1e6, 1)}))
