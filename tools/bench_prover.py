#!/usr/bin/env python3
"""Times h2w_prove_fri (GPU synthetic-proof generator) per phase on BASELINE.json's shapes.
usage: python tools/bench_prover.py [--config cfg2|cfg3|cfg5] [--hash bn254|gl] [--reps 5]
Inputs (random polynomials of degree < 2^degree_bits) are drawn on the device; no oracle involved."""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CONFIGS = {"cfg1": (10, 4, 1), "cfg2": (16, 28, 2), "cfg3": (20, 28, 1), "cfg5": (20, 84, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg3"); ap.add_argument("--hash", default="bn254"); ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1, help="proofs per h2w_prove_fri_batch call (lockstep)")
    a = ap.parse_args()
    import torch
    h2w = importlib.import_module("halo2-plonky2-verifier_amd"); api = importlib.import_module("halo2-plonky2-verifier_amd.api")
    d, q, rb = CONFIGS[a.config]
    mode = 1 if a.hash == "bn254" else 0
    sh = h2w.fibonacci_shape(d, q, rate_bits=rb, hash_mode=mode)
    kh = h2w.published_consts()
    pr = api.Prover(sh, kh)
    g = torch.Generator(device="cuda"); g.manual_seed(0xF1B0)
    B = a.batch
    coefs = torch.randint(0, 1 << 62, (B * pr.num_polys << d,), dtype=torch.int64, device="cuda", generator=g)     # < 2^62 < p: canonical
    proof = torch.zeros(B * pr.proof_words, dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    rows = []
    for r in range(a.reps + 1):
        t0 = time.perf_counter()
        pr.prove_batch(coefs.data_ptr(), [1, 2, 3] * B, proof.data_ptr(), B, s)
        torch.cuda.synchronize()
        t = pr.timing(); t["wall_incl_python"] = (time.perf_counter() - t0) * 1e3
        if r > 0:
            rows.append(t)
    avg = {k: sum(x[k] for x in rows) / len(rows) for k in rows[0]}
    lb = d + rb
    out = {"tool": "bench_prover", "config": a.config, "hash": a.hash, "degree_bits": d, "lde_bits": lb, "queries": q, "polys": pr.num_polys,
           "batch": B, "ms_per_batch": {k: round(v, 3) for k, v in avg.items()}, "proofs_per_s": round(B * 1e3 / avg["total_wall"], 2),
           "permutations": {"oracle_trees": 3 * (1 << lb) + ((1 << lb) if mode == 1 else 0)}}
    print(json.dumps(out))
    # feed the proof to the witness generator: a valid instance satisfies every gate and lookup
    plan = api.Plan(sh, kh)
    advice = torch.zeros(plan.num_cells * 32, dtype=torch.uint8, device="cuda"); ws = torch.zeros(plan.workspace_bytes(1), dtype=torch.uint8, device="cuda")
    plan.run(proof.data_ptr(), 1, advice.data_ptr(), ws.data_ptr(), s); torch.cuda.synchronize()
    print("witness of the first proof: status", plan.status(ws.data_ptr(), 1), "bad gates/lookups", plan.check_constraints(advice.data_ptr(), 1))


if __name__ == "__main__":
    main()
