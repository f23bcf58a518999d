#!/usr/bin/env python3
"""One h2w_fri_witness_batch launch at a time (isolated, chain kernels on the caller's stream): per-kernel times from the library's HIP events
(h2w_plan_timing_ex).  usage: launch_timing.py [--config cfg3] [--hash bn254|gl] [--batch 64] [--world 1] [--rank 0] [--reps 3] [--compact]"""
import argparse, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
CONFIGS = {"cfg1": (10, 4, 1), "cfg2": (16, 28, 2), "cfg3": (20, 28, 1), "cfg5": (20, 84, 1)}
KEYS = ("prologue_values", "perm_records", "glue_strands", "chain_values", "chain_emit", "expand", "launch")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg3"); ap.add_argument("--hash", default="bn254"); ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--world", type=int, default=1); ap.add_argument("--rank", type=int, default=0); ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--compact", action="store_true"); ap.add_argument("--fork", action="store_true"); ap.add_argument("--passes", type=int, default=0); ap.add_argument("--form", type=int, default=0)
    a = ap.parse_args()
    import torch
    import numpy as np
    h2w = importlib.import_module("halo2-plonky2-verifier_amd"); api = importlib.import_module("halo2-plonky2-verifier_amd.api")
    d, q, rb = CONFIGS[a.config]
    sh = h2w.fibonacci_shape(d, q, rate_bits=rb, hash_mode=1 if a.hash == "bn254" else 0)
    plan = api.Plan(sh, h2w.published_consts(), 0)
    if not a.fork:
        plan.configure(1, 0)
    if a.passes:
        plan.configure(3, a.passes)
    if a.form:
        plan.configure(4, a.form)      # H2W_OPT_VALUES_FORM
    B = a.batch
    prng = np.random.default_rng(7)
    proofs = torch.from_numpy(prng.integers(0, 1 << 60, B * plan.proof_words, dtype=np.int64)).cuda()
    cells = plan.shard_cells(B, a.rank, a.world) if a.compact else B * plan.num_cells
    adv = torch.empty(cells * 32, dtype=torch.uint8, device="cuda"); ws = torch.zeros(plan.workspace_bytes(B), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for i in range(a.reps + 1):
        torch.cuda.synchronize()
        if a.world > 1:
            (plan.run_shard_compact if a.compact else plan.run_shard)(proofs.data_ptr(), B, adv.data_ptr(), ws.data_ptr(), a.rank, a.world, st)
        else:
            plan.run(proofs.data_ptr(), B, adv.data_ptr(), ws.data_ptr(), st)
        torch.cuda.synchronize()
        if i:
            out.append(plan.timing_ex(0))
    avg = [sum(t[k] for t in out) / len(out) for k in range(7)]
    print(json.dumps({"config": a.config, "hash": a.hash, "batch": B, "world": a.world, "rank": a.rank, "compact": a.compact, "passes": a.passes, "values_form": a.form, "advice_GB": cells * 32 / 1e9,
                      "ms": {k: round(v, 3) for k, v in zip(KEYS, avg)}}))


if __name__ == "__main__":
    main()
