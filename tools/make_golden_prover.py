#!/usr/bin/env python3
"""Golden digests of the ORACLE prover's output (oracle/prover.inc) at BASELINE.json's full sizes, where running it inside a test
would take minutes: sha256 of the flat proof words for the published Poseidon tables and the inputs of orc_prove_fri_inputs(seed).
-> tests/golden/prover_digests.json.  The GPU generator must reproduce them (tests/test_gpu_prover.py)."""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as O

CASES = [("cfg2_gl", 16, 28, 2, 0), ("cfg2_bn254", 16, 28, 2, 1), ("cfg3_gl", 20, 28, 1, 0)]
if __name__ == "__main__":
    k = O.published_consts()
    out = {}
    for name, d, q, rb, mode in CASES:
        sh = O.fibonacci_shape(d, q, rate_bits=rb, hash_mode=mode)
        seed = 0xF1B00000 + d
        t = time.time()
        coefs, pis = O.prove_fri_inputs(sh, seed)
        words = O.prove_fri_coef(sh, k, coefs, pis)
        out[name] = {"degree_bits": d, "queries": q, "rate_bits": rb, "hash_mode": mode, "seed": seed, "proof_words": len(words),
                     "sha256": hashlib.sha256(bytes(words)).hexdigest(), "pow_witness_word_index": None, "oracle_seconds": round(time.time() - t, 1)}
        print(name, out[name], flush=True)
    with open(os.path.join(ROOT, "tests", "golden", "prover_digests.json"), "w") as f:
        json.dump(out, f, indent=1)
