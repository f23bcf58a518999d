"""The value-level parity hand-off (VERDICT r02 task 6): tools/compare_advice.py diffs a reference-side dump of ctx.advice (rust/h2w-parity,
source only) against the oracle's stream and names the chip call of the first differing cell.  Here: the tool on a dump made from the
oracle itself - identical, then with one flipped cell."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_compare_tool_finds_a_flipped_cell_and_names_its_call_stack():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "compare_advice.py"), "--self-test"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "IDENTICAL" in r.stdout and "self-test ok" in r.stdout
    m = re.search(r"first at cell (\d+): appended by (all;verify_proof[^\s]*) \(cell (\d+) of that call", r.stdout)
    assert m, r.stdout


def test_rust_dump_serialises_the_proof_in_witness_load_order():
    """The Rust module is source only; what can be checked here: it walks the proof in the order of the flat layout - the order
    WitnessChip::load_proof_with_pis reads it (verifier/src/witness/mod.rs:236-294)."""
    src = open(os.path.join(ROOT, "rust", "h2w-parity", "parity_dump.rs")).read()
    body = src[src.index("fn flat_words"):src.index("fn write_case")]
    order = ["trace_cap", "quotient_polys_cap", "local_values", "next_values", "permutation_zs ", "permutation_zs_next", "quotient_polys {", "permutation_zs_cap",
             "pow_witness", "final_poly", "commit_phase_merkle_caps", "query_round_proofs", "initial_trees_proof", "steps", "public_inputs"]
    pos = [body.index(k) for k in order]
    assert pos == sorted(pos), "flat_words does not follow the witness load order"
    for name in ("case.json", "proof.words", "advice.bin"):
        assert name in src
