#!/usr/bin/env python3
"""Trace h2w_chip_verify_stark once, lower it, replay it on --batch proofs: wall time per launch (and, under rocprofv3 --kernel-trace, the duration of
every k_replay dispatch: one per template, depth by depth).  usage: replay_timing.py [--config cfg3] [--hash bn254|gl] [--batch 64] [--reps 2]"""
import argparse, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CONFIGS = {"cfg1": (10, 4, 1), "cfg2": (16, 28, 2), "cfg3": (20, 28, 1), "cfg5": (20, 84, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg3"); ap.add_argument("--hash", default="bn254"); ap.add_argument("--batch", type=int, default=64); ap.add_argument("--reps", type=int, default=2); ap.add_argument("--streams", type=int, default=1)
    a = ap.parse_args()
    import numpy as np
    import torch
    h2w = importlib.import_module("halo2-plonky2-verifier_amd"); api = importlib.import_module("halo2-plonky2-verifier_amd.api")
    d, q, rb = CONFIGS[a.config]
    sh = h2w.fibonacci_shape(d, q, rate_bits=rb, hash_mode=1 if a.hash == "bn254" else 0)
    k = h2w.published_consts()
    ref = api.Plan(sh, k, 0)
    words = ref.proof_words
    prng = np.random.default_rng(7)
    host = prng.integers(0, 1 << 60, a.batch * words, dtype=np.int64)
    ctx = api.Context(21, True, 0); ctx.trace_begin()
    t0 = time.perf_counter(); api.verify_stark(ctx, sh, k, host[:words].astype(np.uint64)); t1 = time.perf_counter()
    plan = api.Plan.from_trace(ctx, words); t2 = time.perf_counter()
    ctx.close()
    proofs = torch.from_numpy(host).cuda()
    adv = torch.empty(a.batch * plan.num_cells * 32, dtype=torch.uint8, device="cuda"); ws = torch.zeros(plan.workspace_bytes(a.batch), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    if a.streams > 1:      # launches in flight: the root kernel of one (two wavefronts) runs beside the other's Merkle lanes
        strs = [torch.cuda.Stream() for _ in range(a.streams)]
        advs = [adv] + [torch.empty_like(adv) for _ in range(a.streams - 1)]; wss = [ws] + [torch.zeros_like(ws) for _ in range(a.streams - 1)]
        for j in range(a.streams):
            plan.run(proofs.data_ptr(), a.batch, advs[j].data_ptr(), wss[j].data_ptr(), strs[j].cuda_stream)
        torch.cuda.synchronize(); t = time.perf_counter(); nl = 3 * a.streams
        for j in range(nl):
            plan.run(proofs.data_ptr(), a.batch, advs[j % a.streams].data_ptr(), wss[j % a.streams].data_ptr(), strs[j % a.streams].cuda_stream)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(json.dumps({"config": a.config, "hash": a.hash, "batch": a.batch, "streams": a.streams, "launches": nl, "ms_per_launch": round(dt / nl * 1e3, 2), "G_cells_per_s": round(plan.num_cells * a.batch * nl / dt / 1e9, 2)}))
        del advs, wss
    ms = []
    for i in range(a.reps + 1):
        torch.cuda.synchronize(); t = time.perf_counter()
        plan.run(proofs.data_ptr(), a.batch, adv.data_ptr(), ws.data_ptr(), st)
        torch.cuda.synchronize(); ms.append((time.perf_counter() - t) * 1e3)
    assert plan.status(ws.data_ptr(), a.batch, st) == [0] * a.batch
    # tie to the compiled plan's stream of proof 0
    chk = torch.empty(ref.num_cells * 32, dtype=torch.uint8, device="cuda"); cws = torch.zeros(ref.workspace_bytes(1), dtype=torch.uint8, device="cuda")
    ref.run(proofs.data_ptr(), 1, chk.data_ptr(), cws.data_ptr(), st); torch.cuda.synchronize()
    print(json.dumps({"config": a.config, "hash": a.hash, "batch": a.batch, "trace_s": round(t1 - t0, 3), "lower_s": round(t2 - t1, 3), "records": plan.num_records, "ws_GB": round(plan.workspace_bytes(a.batch) / 1e9, 2),
                      "ms_per_launch": [round(x, 2) for x in ms[1:]], "G_cells_per_s": round(plan.num_cells * a.batch / (min(ms[1:]) * 1e-3) / 1e9, 2), "proof0_equals_compiled_plan": bool(torch.equal(chk, adv[:ref.num_cells * 32]))}))


if __name__ == "__main__":
    main()
