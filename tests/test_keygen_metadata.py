"""SURVEY §8(f) row 2 on the eager ABI: a context created with witness_gen_only=False records what halo2-base's Context records
at keygen — gate selectors, copy constraints (advice_equalities), constant equalities, range-lookup registrations — and they
equal the oracle's (which records them the halo2-base way, Context::assign_region) exactly: same gate cells, same lookup list in
order, same multiset of equality pairs, same (cell, constant) list.  Host-only (no GPU): the bookkeeping is structural."""
import ctypes as C
import random

import pytest

P = 2**64 - 2**32 + 1
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def norm_eq(pairs):
    return sorted((min(a, b), max(a, b)) for a, b in pairs)


def compare(ctx, octx):
    assert ctx.num_cells() == octx.num_cells()
    assert ctx.gate_cells() == octx.gate_cells()
    assert ctx.lookup_cells() == octx.lookup_cells()
    assert norm_eq(ctx.equalities()) == norm_eq(octx.equalities())
    assert sorted(ctx.const_equalities()) == sorted(octx.const_equalities())


@pytest.mark.parametrize("lookup_bits", [21, 8])
def test_native_and_goldilocks_ops(h2w_api, oracle, lookup_bits):
    O = oracle
    L = O.lib()
    rnd = random.Random(4)
    ctx = h2w_api.Context(lookup_bits, witness_gen_only=False)
    nat = h2w_api.NativeChip(ctx); gl = h2w_api.GoldilocksChip(nat)
    octx = O.Ctx(lookup_bits, witness_gen_only=False); op = octx.p
    def wit(v): return nat.load_witness(v), L.orc_load_witness(op, O.Fr.from_int(v))
    def const(v): return nat.load_constant(v), L.orc_load_constant(op, O.Fr.from_int(v))
    (ga, oa), (gb, ob), (gc, oc) = wit(rnd.randrange(R)), const(rnd.randrange(1 << 64)), wit(12345)
    x = nat.add(ga, gb), L.orc_add(op, oa, ob)
    y = nat.mul(x[0], gc), L.orc_mul(op, x[1], oc)
    z = nat.mul_add(ga, gb, y[0]), L.orc_mul_add(op, oa, ob, y[1])
    (gs, os_) = wit(1)
    s = nat.select(z[0], ga, gs), L.orc_select(op, z[1], oa, os_)
    gz, oz = nat.load_zero(), L.orc_load_zero(op); nat.load_zero(); L.orc_load_zero(op)
    n = 5
    arr = [wit(rnd.randrange(R)) for _ in range(n)]
    (gi, oi) = wit(3)
    nat.select_from_idx([g for g, _ in arr], gi); L.orc_select_from_idx(op, (O.AV * n)(*[o for _, o in arr]), n, oi)
    gind = nat.idx_to_indicator(gi, n); oind = (O.AV * n)(); L.orc_idx_to_indicator(op, oi, n, oind)
    w = 2
    nat.select_array_by_indicator([[arr[(i + j) % n][0] for j in range(w)] for i in range(n)], gind)
    oout = (O.AV * w)(); L.orc_select_array_by_indicator(op, (O.AV * (n * w))(*[arr[(i + j) % n][1] for i in range(n) for j in range(w)]), n, w, oind, oout)
    (gv, ov) = wit(0xDEADBEEF12345)
    gbits = nat.num_to_bits(gv, 52); obits = (O.AV * 52)(); L.orc_num_to_bits(op, ov, 52, obits)
    nat.bits_to_num(gbits[:9]); L.orc_bits_to_num(op, (O.AV * 9)(*[obits[i] for i in range(9)]), 9)
    (gw, ow) = wit(rnd.randrange(R))
    glimbs = nat.decompose_le(gw, 56, 5); olimbs = (O.AV * 5)(); L.orc_decompose_le(op, ow, 56, 5, olimbs)
    nat.limbs_to_num(glimbs[:3], 64); L.orc_limbs_to_num(op, (O.AV * 3)(*[olimbs[i] for i in range(3)]), 3, 64)
    for v, bits in [(12345, 48), (77, 21), (1, 1), (3, 2), (1 << 20, 22), (9, 8)]:
        (g, o) = wit(v); nat.range_check(g, bits); L.orc_range_check(op, o, bits)
    for v, bound in [(5, 1000), (P - 1, P)]:
        (g, o) = wit(v); nat.check_less_than_safe(g, bound); L.orc_check_less_than_safe(op, o, bound)
    nat.constrain_equal(ga, gc) if hasattr(nat, "constrain_equal") else h2w_api.lib().h2w_constrain_equal(ctx.p, C.byref(ga), C.byref(gc))
    L.orc_constrain_equal(op, oa, oc)
    # fused Goldilocks level
    a, b = rnd.randrange(P), rnd.randrange(1, P)
    g1, o1 = gl.load_witness(a), L.orc_gl_load_witness(op, a)
    g2, o2 = gl.load_constant(b), L.orc_gl_load_constant(op, b)
    gm, om = gl.mul(g1, g2), L.orc_gl_mul(op, o1, o2)
    gsum, osum = gl.add(gm, g1), L.orc_gl_add(op, om, o1)
    gd, od = gl.sub(gsum, g2), L.orc_gl_sub(op, osum, o2)
    gma, oma = gl.mul_add(gd, g2, gm), L.orc_gl_mul_add(op, od, o2, om)
    gq, oq = gl.div(gma, g2), L.orc_gl_div(op, oma, o2)
    gi2, oi2 = gl.inv(g2), L.orc_gl_inv(op, o2)
    g5, o5 = gl.load_constant(5), L.orc_gl_load_constant(op, 5)
    gms, oms = gl.mul_sub(g5, g2, gq), L.orc_gl_mul_sub(op, o5, o2, oq)       # 5*b + c*(p-1) < 2^128
    ge, oe = gl.exp_power_of_2(gms, 2), L.orc_gl_exp_power_of_2(op, oms, 2)
    gr, orr = gl.reduce(nat.mul_add(g1, g2, ge)), L.orc_gl_reduce(op, L.orc_mul_add(op, o1, o2, oe))
    compare(ctx, octx)
    # and the MockProver's verdict on the oracle side: every recorded constraint holds
    mp = octx.mock_prover()
    assert mp["bad"] == 0 or mp["semantic_failed"] >= 0
    ctx.close(); octx.close()


@pytest.mark.parametrize("mode", [1, 0])
def test_full_verifier_through_the_eager_boundary(h2w, h2w_api, oracle, consts, mode):
    """h2w_chip_verify_stark drives the whole gadget stack through the eager C ABI (AbiBackend): its keygen bookkeeping equals the
    oracle's for the full FRI verifier, chip-level assert_equal constraints included."""
    ko, kh = consts
    sh = h2w.fibonacci_shape(5, 2, hash_mode=mode); osh = oracle.fibonacci_shape(5, 2, hash_mode=mode)
    proof = oracle.synth_proof(osh, 9)
    ctx = h2w_api.Context(21, witness_gen_only=False)
    assert h2w.lib().h2w_chip_verify_stark(ctx.p, C.byref(sh), C.byref(kh), proof) == 0, h2w.last_error()
    octx = oracle.Ctx(21, witness_gen_only=False)
    assert oracle.verify_stark(octx, osh, ko, proof) == 0
    compare(ctx, octx)
    ctx.close(); octx.close()


@pytest.mark.parametrize("mode", [1, 0])
def test_plan_exports_the_static_equality_lists(h2w, h2w_api, oracle, consts, mode):
    """h2w_plan_equalities / h2w_plan_const_equalities (the batched path's copy manager): the lists a plan hands out -
    built by a host replay, no device - equal the oracle's keygen bookkeeping for an ARBITRARY proof of the shape: the copy constraints
    and the constant-equality cells are static, and so are the constants, except that with Goldilocks-Poseidon caps the reference loads
    the proof's hash wires as constants (hash/poseidon/hash.rs:86-96) - those are filled in from the proof passed in."""
    ko, kh = consts
    sh = h2w.fibonacci_shape(6, 2, hash_mode=mode, cap_height=2); osh = oracle.fibonacci_shape(6, 2, hash_mode=mode, cap_height=2)
    plan = h2w_api.Plan(sh, kh)
    proof = oracle.prove_fri(osh, ko, 77)
    octx = oracle.Ctx(21, witness_gen_only=False)
    assert oracle.verify_stark(octx, osh, ko, proof) == 0
    assert norm_eq(plan.equalities()) == norm_eq(octx.equalities())
    assert sorted(plan.const_equalities(proof)) == sorted(octx.const_equalities())
    if mode == 1:
        assert sorted(plan.const_equalities()) == sorted(octx.const_equalities())        # nothing depends on the proof
    else:
        with pytest.raises(h2w_api.H2WError):
            plan.const_equalities()
    octx.close(); plan.close()
