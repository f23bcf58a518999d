#!/usr/bin/env python3
"""Value-level parity hand-off: diff an advice stream dumped by the REFERENCE (rust/h2w-parity, run by someone with cargo) against this
repository's stream for the same proof, and name the chip call in which they first differ.

Inputs (one directory, written by rust/h2w-parity/parity_dump.rs inside the reference crate):
  case.json    {"degree_bits", "num_queries", "rate_bits", "cap_height", "pow_bits", "num_challenges", "arity_bits", "final_poly_bits",
                "n_cols", "n_perm_z", "n_quotient", "n_pis", "perm_batch_size", "hash_mode", "lookup_bits", "witness_load_range_check"}
  proof.words  the proof as little-endian u64 words in the flat layout of INTEGRATION.md (= WitnessChip::load_proof_with_pis order)
  advice.bin   ctx.advice of the reference after load_proof_with_pis + verify_proof (util/context_wrapper.rs:24-26), 32 bytes per cell,
               canonical little-endian (Fr::to_repr)

What is compared: advice.bin against the CPU oracle's stream (always) and against libh2w's stream from the GPU (`--gpu`, needs a device).
On a mismatch: the first differing cell, the #[count] call stack that appends it in the oracle (the chip function and hence the halo2-base
template) and its offset inside that call's block; INTEGRATION.md ("Closing the parity pin") maps each template whose cell
ORDER is recollection to the functions that encode it.  Exit code 0 = identical.

    python tools/compare_advice.py DIR [--gpu]
    python tools/compare_advice.py --self-test      # writes a case with the oracle's own stream (plus one flipped cell) and runs the diff on it
"""
import argparse, json, os, struct, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
FIELDS = ["degree_bits", "rate_bits", "cap_height", "num_queries", "pow_bits", "num_challenges", "arity_bits", "final_poly_bits", "n_cols", "n_perm_z",
          "n_quotient", "n_pis", "perm_batch_size", "hash_mode", "lookup_bits", "witness_load_range_check"]


def oracle_stream(case, words, watch=None):
    import pyoracle as O
    sh = O.Shape(**{k: int(case[k]) for k in FIELDS})
    assert O.lib().orc_proof_words(sh) == len(words), f"proof.words holds {len(words)} words, the shape needs {O.lib().orc_proof_words(sh)}"
    ctx = O.Ctx(int(case["lookup_bits"]), True, track_scopes=watch is not None)
    if watch is not None:
        ctx.watch_cell(watch)
    arr = (O.C.c_uint64 * len(words))(*words)
    rc = O.verify_stark(ctx, sh, O.published_consts(), arr)
    out = ctx.advice_bytes(), (ctx.watch_path() if watch is not None else None), rc
    ctx.close()
    return out


def gpu_stream(case, words):
    import importlib, torch
    h2w = importlib.import_module("halo2-plonky2-verifier_amd"); api = importlib.import_module("halo2-plonky2-verifier_amd.api")
    sh = h2w.Shape(**{k: int(case[k]) for k in FIELDS})
    plan = api.Plan(sh, h2w.published_consts(), 0)
    d = torch.tensor(words, dtype=torch.uint64).view(torch.int64).cuda() if hasattr(torch, "uint64") else torch.frombuffer(bytearray(struct.pack(f"<{len(words)}Q", *words)), dtype=torch.int64).cuda()
    adv = torch.zeros(plan.num_cells * 32, dtype=torch.uint8, device="cuda"); ws = torch.zeros(plan.workspace_bytes(1), dtype=torch.uint8, device="cuda")
    plan.run(d.data_ptr(), 1, adv.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return adv.cpu().numpy().tobytes()


def first_diff(a, b):
    n = min(len(a), len(b)) // 32
    import numpy as np
    x = np.frombuffer(a[:n * 32], dtype=np.uint64).reshape(n, 4); y = np.frombuffer(b[:n * 32], dtype=np.uint64).reshape(n, 4)
    bad = np.nonzero((x != y).any(axis=1))[0]
    return (int(bad[0]), len(bad)) if len(bad) else (None, 0)


def compare(directory, gpu=False):
    case = json.load(open(os.path.join(directory, "case.json")))
    raw = open(os.path.join(directory, "proof.words"), "rb").read()
    words = list(struct.unpack(f"<{len(raw) // 8}Q", raw))
    ref = open(os.path.join(directory, "advice.bin"), "rb").read()
    mine, _, rc = oracle_stream(case, words)
    print(f"reference: {len(ref) // 32} cells; oracle: {len(mine) // 32} cells (verify_stark status {rc})")
    ok = True
    streams = [("oracle (CPU restatement)", mine)] + ([("libh2w (GPU)", gpu_stream(case, words))] if gpu else [])
    for name, s in streams:
        cell, n = first_diff(ref, s)
        if cell is None and len(ref) == len(s):
            print(f"{name}: IDENTICAL to the reference's ctx.advice")
            continue
        ok = False
        if cell is None:
            print(f"{name}: equal on the common prefix, lengths differ ({len(ref) // 32} vs {len(s) // 32} cells)"); cell = min(len(ref), len(s)) // 32 - 1
        _, where, _ = oracle_stream(case, words, watch=cell)
        lo = max(cell - 2, 0)
        print(f"{name}: {n} cells differ, first at cell {cell}: appended by {where[0]} (cell {where[1]} of that call's block)")
        for i in range(lo, min(cell + 3, len(ref) // 32, len(s) // 32)):
            r = int.from_bytes(ref[32 * i:32 * i + 32], "little"); m = int.from_bytes(s[32 * i:32 * i + 32], "little")
            print(f"   cell {i}: reference {r:#x}   here {m:#x}{'   <--' if r != m else ''}")
    return 0 if ok else 1


def self_test():
    import pyoracle as O
    case = dict(zip(FIELDS, [6, 1, 2, 2, 16, 2, 4, 5, 4, 2, 2, 3, 1, 1, 21, 1]))
    sh = O.Shape(**case)
    words = list(O.synth_proof(sh, 77))
    adv, _, _ = oracle_stream(case, words)
    with tempfile.TemporaryDirectory() as d:
        json.dump(case, open(os.path.join(d, "case.json"), "w"))
        open(os.path.join(d, "proof.words"), "wb").write(struct.pack(f"<{len(words)}Q", *words))
        open(os.path.join(d, "advice.bin"), "wb").write(adv)
        assert compare(d) == 0
        cell = len(adv) // 64                      # flip one cell in the middle: the tool must find it and name its call stack
        bad = bytearray(adv); bad[32 * cell] ^= 1
        open(os.path.join(d, "advice.bin"), "wb").write(bytes(bad))
        assert compare(d) == 1
    print("self-test ok")
    return 0


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("dir", nargs="?"); ap.add_argument("--gpu", action="store_true"); ap.add_argument("--self-test", action="store_true")
    a = ap.parse_args()
    sys.exit(self_test() if a.self_test or not a.dir else compare(a.dir, a.gpu))
