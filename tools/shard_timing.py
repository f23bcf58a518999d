#!/usr/bin/env python3
"""What one rank of a (proof, query)-sharded launch costs, measured on ONE GPU: the same batch of proofs through h2w_fri_witness_batch (world 1) and through
h2w_fri_witness_batch_shard[_compact] as rank r of `world` (every rank's launch, one after the other), kernel by kernel from the library's HIP events.
usage: shard_timing.py [--config cfg5] [--batch 16] [--world 8] [--passes 0]"""
import argparse, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CONFIGS = {"cfg1": (10, 4, 1), "cfg2": (16, 28, 2), "cfg3": (20, 28, 1), "cfg5": (20, 84, 1)}
KEYS = ("prologue_values", "perm_records", "glue_strands", "chain_values", "chain_emit", "expand", "launch")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg5"); ap.add_argument("--hash", default="bn254"); ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--world", type=int, default=8); ap.add_argument("--passes", type=int, default=0); ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import torch
    import numpy as np
    h2w = importlib.import_module("halo2-plonky2-verifier_amd"); api = importlib.import_module("halo2-plonky2-verifier_amd.api")
    d, q, rb = CONFIGS[a.config]
    sh = h2w.fibonacci_shape(d, q, rate_bits=rb, hash_mode=1 if a.hash == "bn254" else 0)
    plan = api.Plan(sh, h2w.published_consts(), 0)
    plan.configure(1, 0)                      # chain kernels on the caller's stream: every interval is one kernel's duration
    if a.passes:
        plan.configure(3, a.passes)
    B = a.batch
    prng = np.random.default_rng(7)
    proofs = torch.from_numpy(prng.integers(0, 1 << 60, B * plan.proof_words, dtype=np.int64)).cuda()
    adv = torch.empty(B * plan.num_cells * 32, dtype=torch.uint8, device="cuda"); ws = torch.zeros(plan.workspace_bytes(B), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def timed(fn):
        rows = []
        for i in range(a.reps + 1):
            torch.cuda.synchronize(); fn(); torch.cuda.synchronize()
            if i:
                rows.append(plan.timing_ex(0))
        return {k: round(sum(r[j] for r in rows) / len(rows), 3) for j, k in enumerate(KEYS)}, int(rows[0][7])
    full, pf = timed(lambda: plan.run(proofs.data_ptr(), B, adv.data_ptr(), ws.data_ptr(), st))
    out = {"config": a.config, "hash": a.hash, "proofs": B, "cells_per_proof": plan.num_cells, "world": a.world,
           "unsharded": {"ms": full, "merkle_path_passes": pf, "advice_GB": B * plan.num_cells * 32 / 1e9}, "ranks": []}
    for r in range(a.world):
        ms, pr = timed(lambda: plan.run_shard(proofs.data_ptr(), B, adv.data_ptr(), ws.data_ptr(), r, a.world, st))
        cms, _ = timed(lambda: plan.run_shard_compact(proofs.data_ptr(), B, adv.data_ptr(), ws.data_ptr(), r, a.world, st))
        out["ranks"].append({"rank": r, "ms": ms, "ms_packed_layout": cms, "merkle_path_passes": pr, "advice_GB_packed": plan.shard_cells(B, r, a.world) * 32 / 1e9})
    worst = {k: max(x["ms"][k] for x in out["ranks"]) for k in KEYS}
    out["slowest_rank_over_unsharded"] = {k: round(worst[k] / full[k], 3) if full[k] > 0.01 else None for k in KEYS}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
