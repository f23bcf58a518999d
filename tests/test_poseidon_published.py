"""Known-answer pin of the two Poseidon permutations on the path against PUBLISHED vectors.

The reference's own tests hold no literals, but the parameter sets it links in are public and so are their
test vectors.  tools/gen_poseidon_constants.py re-derives the tables from the public generation procedures (Grain LFSR ->
circomlib's BN254 constants; ChaCha8Rng(0) -> plonky2's Goldilocks constants) and tests/golden/poseidon_published.json holds
the result.  Here: the generator reproduces the committed file and the published anchors; the oracle's value-domain and
cell-domain permutations, fed those tables, reproduce the published outputs:

  * plonky2 poseidon_goldilocks.rs test vectors: permute([0;12]), permute([0..11]), permute([-1;12])
  * circomlib / go-iden3-crypto: poseidon([1,2]) (t=3), poseidon([1,2,3,4]) (t=5) for the generator; the t=4 tables the path
    uses come from the same generator and the optimised form equals the plain permutation.
"""
import ctypes as C
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "poseidon_published.json")
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


@pytest.fixture(scope="module")
def gold():
    return json.load(open(GOLD))


@pytest.fixture(scope="module")
def gen():
    spec = importlib.util.spec_from_file_location("gen_poseidon_constants", os.path.join(ROOT, "tools", "gen_poseidon_constants.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_generator_reproduces_the_committed_tables_and_published_anchors(gen, gold):
    """bn254_tables() / goldilocks_tables() assert the published anchors internally (circomlib C[0], M[0][0], three hash
    outputs; plonky2 constant heads and three permutation vectors)."""
    assert gen.bn254_tables(verbose=False) == gold["bn254_t4"]
    assert gen.goldilocks_tables(verbose=False) == gold["goldilocks_w12"]
    assert gold["goldilocks_w12"]["fast_heads_match"] == {"first": True, "fast_rc": True}
    # decimal forms as printed in go-iden3-crypto's test-suite
    assert int(gold["bn254_t4"]["published"]["hash_1_2"], 16) == 7853200120776062878684798364095072458815029376092732009249414926327459813530
    c2, m2 = gen.grain_params(2, 8, 56)
    assert gen.poseidon_plain([0, 1], c2, m2, 8, 56)[0] == 18586133768512220936620570745912940619677854269274689475585506675881198879027


def test_oracle_goldilocks_permutation_known_answers(oracle, gold):
    k = oracle.published_consts()
    L = oracle.lib()
    vecs = gold["goldilocks_w12"]["permutation_vectors"]
    assert vecs[0]["in"] == ["0x0"] * 12 and vecs[0]["out"][0] == "0x3c18a9786cb0b359" and vecs[0]["out"][11] == "0x1792b1c4342109d7"
    for v in vecs:
        st = (C.c_uint64 * 12)(*[int(x, 16) for x in v["in"]])
        L.orc_nv_gl_permute(C.byref(k), st)                                    # value domain (fast partial rounds)
        assert [hex(x) for x in st] == v["out"]
        ctx = oracle.Ctx(21, witness_gen_only=False)                          # cell domain: 163,478 cells, constraints hold
        ins = (oracle.AV * 12)(*[L.orc_gl_load_constant(ctx.p, int(x, 16)) for x in v["in"]])
        outs = (oracle.AV * 12)()
        n0 = ctx.num_cells()
        L.orc_gl_poseidon_permute(ctx.p, C.byref(k), ins, outs)
        assert ctx.num_cells() - n0 == 163478
        assert [hex(o.v.to_int()) for o in outs] == v["out"]
        assert ctx.mock_prover()["bad"] == 0
        ctx.close()


def test_oracle_bn254_permutation_known_answers(oracle, gold):
    k = oracle.published_consts()
    L = oracle.lib()
    vecs = gold["bn254_t4"]["permutation_vectors"]
    assert vecs[0]["in"] == ["0x0", "0x1", "0x2", "0x3"]
    assert vecs[0]["out"][0] == "0xe7732d89e6939c0ff03d5e58dab6302f3230e269dc5b968f725df34ab36d732"    # circomlib poseidon([1,2,3])
    for v in vecs:
        st = (oracle.Fr * 4)(*[oracle.Fr.from_int(int(x, 16)) for x in v["in"]])
        L.orc_nv_bn_permute(C.byref(k), st)
        assert [hex(x.to_int()) for x in st] == v["out"]
        ctx = oracle.Ctx(21, witness_gen_only=False)
        ins = (oracle.AV * 4)(*[L.orc_load_constant(ctx.p, oracle.Fr.from_int(int(x, 16))) for x in v["in"]])
        outs = (oracle.AV * 4)()
        L.orc_load_zero(ctx.p)                                                 # halo2-base caches the zero cell: not part of the 4,032
        n0 = ctx.num_cells()
        L.orc_bn_poseidon_permute(ctx.p, C.byref(k), ins, outs)
        assert ctx.num_cells() - n0 == 4032
        assert [hex(o.v.to_int()) for o in outs] == v["out"]
        assert ctx.mock_prover()["bad"] == 0
        ctx.close()


def test_product_tables_are_the_published_ones(h2w, oracle):
    """h2w_poseidon_published (csrc/poseidon_tables.h, generated) == the golden file, byte for byte; pure data, no device."""
    assert bytes(h2w.published_consts()) == bytes(oracle.published_consts())


@pytest.mark.parametrize("mode", [1, 0])
def test_valid_fri_instance_under_the_published_tables(oracle, mode):
    """The whole restated verifier on a valid FRI instance hashed with the real parameter sets: every constraint holds."""
    k = oracle.published_consts()
    sh = oracle.fibonacci_shape(9, 2, rate_bits=1, cap_height=2, hash_mode=mode)
    pr = oracle.prove_fri(sh, k, 0xF1B0009)
    ctx = oracle.Ctx(21, witness_gen_only=False)
    assert oracle.verify_stark(ctx, sh, k, pr) == 0, ctx.error()
    mp = ctx.mock_prover()
    assert mp["bad"] == 0 and mp["semantic_failed"] == 0, mp
    ctx.close()
