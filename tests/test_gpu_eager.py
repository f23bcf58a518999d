"""GPU parity, API levels 1-2: the eager NativeChip / fused GoldilocksChip C-ABI (cells expanded by the HIP
kernel) against the CPU oracle, byte for byte, on the same seeded inputs.  Reads like the reference's unit tests
(field/goldilocks/base.rs:476-495 test_mul: 100 random pairs, etc.)."""
import random

import pytest

pytestmark = pytest.mark.gpu

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
P = 2**64 - 2**32 + 1


class Pair:
    """Runs the same call sequence on the product (GPU) and on the oracle (CPU)."""

    def __init__(self, h2w_api, oracle, lookup_bits=21):
        self.api, self.O = h2w_api, oracle
        self.ctx = h2w_api.Context(lookup_bits)
        self.nat = h2w_api.NativeChip(self.ctx)
        self.gl = h2w_api.GoldilocksChip(self.nat)
        self.octx = oracle.Ctx(lookup_bits)
        self.L = oracle.lib()

    def check(self):
        got = self.ctx.advice_bytes()
        want = self.octx.advice_bytes()
        assert self.ctx.num_cells() == self.octx.num_cells()
        if got != want:
            for i in range(self.octx.num_cells()):
                if got[i * 32:(i + 1) * 32] != want[i * 32:(i + 1) * 32]:
                    raise AssertionError(f"cell {i}: got {int.from_bytes(got[i*32:(i+1)*32],'little'):#x} want {int.from_bytes(want[i*32:(i+1)*32],'little'):#x}")
        self.ctx.close(); self.octx.close()

    # paired loads
    def const(self, v):
        return self.nat.load_constant(v), self.L.orc_load_constant(self.octx.p, self.O.Fr.from_int(v))

    def wit(self, v):
        return self.nat.load_witness(v), self.L.orc_load_witness(self.octx.p, self.O.Fr.from_int(v))


@pytest.mark.parametrize("lookup_bits", [21, 13, 8])
def test_goldilocks_ops(h2w_api, oracle, lookup_bits):
    rnd = random.Random(10 + lookup_bits)
    pr = Pair(h2w_api, oracle, lookup_bits)
    L, op = pr.L, pr.octx.p
    edge = [0, 1, P - 1, P - 2, 2**32, 2**32 - 1, 2**63]
    vals = [(rnd.choice(edge), rnd.choice(edge)) for _ in range(20)] + [(rnd.randrange(P), rnd.randrange(P)) for _ in range(100)]
    for a, b in vals:
        ga, oa = pr.gl.load_constant(a), L.orc_gl_load_constant(op, a)
        gb, ob = pr.gl.load_constant(b), L.orc_gl_load_constant(op, b)
        gw, ow = pr.gl.load_witness(b), L.orc_gl_load_witness(op, b)
        gm, om = pr.gl.mul(ga, gb), L.orc_gl_mul(op, oa, ob)
        assert gm.int_value() == om.v.to_int() == a * b % P and gm.offset == om.cell
        gs, os_ = pr.gl.add(gm, gw), L.orc_gl_add(op, om, ow)
        gd, od = pr.gl.sub(gs, ga), L.orc_gl_sub(op, os_, oa)
        gma, oma = pr.gl.mul_add(gd, gb, gm), L.orc_gl_mul_add(op, od, ob, om)
        assert gma.int_value() == oma.v.to_int() and gma.offset == oma.cell
        if b % P != 0:
            gq, oq = pr.gl.div(ga, gb), L.orc_gl_div(op, oa, ob)
            assert gq.int_value() == oq.v.to_int() and gq.offset == oq.cell
            gi, oi = pr.gl.inv(gb), L.orc_gl_inv(op, ob)
            assert gi.int_value() == oi.v.to_int()
        if a * b + b * (P - 1) < 2**128:       # beyond that the reference's reduce is out of its supported range (base.rs:345 TODO); the ABI returns an error
            gms, oms = pr.gl.mul_sub(ga, gb, gw), L.orc_gl_mul_sub(op, oa, ob, ow)
            assert gms.int_value() == oms.v.to_int() == (a * b - b) % P and gms.offset == oms.cell
            gsq, osq = pr.gl.exp_power_of_2(gms, 3), L.orc_gl_exp_power_of_2(op, oms, 3)
            assert gsq.int_value() == osq.v.to_int() == pow((a * b - b) % P, 8, P) and gsq.offset == osq.cell
        else:
            with pytest.raises(h2w_api.H2WError):
                pr.gl.mul_sub(ga, gb, gw)
        gn, on_ = pr.gl.neg(ga), L.orc_gl_mul(op, oa, L.orc_gl_load_constant(op, P - 1))        # neg = load_neg_one, mul (base.rs:234-238)
        assert gn.int_value() == on_.v.to_int() == (-a) % P and gn.offset == on_.cell
        # reduce of an unreduced native product / sum
        gp, opd = pr.nat.mul_add(ga, gb, gw), L.orc_mul_add(op, oa, ob, ow)
        gr, orr = pr.gl.reduce(gp), L.orc_gl_reduce(op, opd)
        assert gr.int_value() == orr.v.to_int() == (a * b + b) % P and gr.offset == orr.cell
    pr.check()


@pytest.mark.parametrize("lookup_bits", [21, 13])
def test_reduce_at_the_edges_of_its_range(h2w_api, oracle, lookup_bits):
    """GoldilocksChip::reduce (base.rs:346-368) on 128-bit values around the points where its hint changes shape: the largest
    mul_add value p(p-1)+(p-1), p^2 (quotient = p wraps to 0), 2^128-1 (quotient >= 2^64), multiples of p, tiny values.  The cells
    follow the reference even where its constraint would fail (quotient reduced mod p)."""
    pr = Pair(h2w_api, oracle, lookup_bits)
    L, op = pr.L, pr.octx.p
    for v in [0, 1, P - 1, P, P + 1, 2**64 - 1, 2**64, P * (P - 1) + (P - 1), P * P - 1, P * P, P * P + 1, (P - 2) * 2**64, 2**127, 2**128 - 2**64, 2**128 - 1,
              (2**64 + 2**32) * P - 1, (2**64 + 2**32) * P]:
        if v >= 2**128:
            continue
        (g, o) = pr.const(v)
        gr, orr = pr.gl.reduce(g), L.orc_gl_reduce(op, o)
        assert gr.int_value() == orr.v.to_int() == v % P and gr.offset == orr.cell, hex(v)
    pr.check()


def test_native_ops(h2w_api, oracle):
    rnd = random.Random(11)
    pr = Pair(h2w_api, oracle)
    L, op, O = pr.L, pr.octx.p, oracle
    small = [rnd.randrange(1 << 64) for _ in range(8)]
    wide = [rnd.randrange(R) for _ in range(8)] + [R - 1, 1 << 128, (1 << 200) + 12345]
    for a, b, c in [(rnd.choice(small + wide), rnd.choice(small + wide), rnd.choice(small + wide)) for _ in range(60)]:
        (ga, oa), (gb, ob), (gc, oc) = pr.wit(a), pr.const(b), pr.wit(c)
        x, y = pr.nat.add(ga, gb), L.orc_add(op, oa, ob); assert x.int_value() == y.v.to_int() == (a + b) % R
        x, y = pr.nat.mul(ga, gb), L.orc_mul(op, oa, ob); assert x.int_value() == y.v.to_int() == a * b % R
        x, y = pr.nat.mul_add(ga, gb, gc), L.orc_mul_add(op, oa, ob, oc); assert x.int_value() == y.v.to_int() == (a * b + c) % R and x.offset == y.cell
        for bit in (0, 1):
            (gs, os_) = pr.const(bit)
            x, y = pr.nat.select(ga, gb, gs), L.orc_select(op, oa, ob, os_); assert x.int_value() == y.v.to_int() == (a if bit else b) and x.offset == y.cell
    gz, oz = pr.nat.load_zero(), L.orc_load_zero(op)
    gz2, oz2 = pr.nat.load_zero(), L.orc_load_zero(op)   # cached: no new cell
    assert gz.offset == gz2.offset == oz.cell == oz2.cell
    for n, idx in [(16, 0), (16, 5), (16, 15), (2, 1), (1, 0), (16, 40)]:
        garr = []; oarr = []
        for i in range(n):
            g, o = pr.wit(rnd.choice(small + wide)); garr.append(g); oarr.append(o)
        (gi, oi) = pr.wit(idx)
        x, y = pr.nat.select_from_idx(garr, gi), L.orc_select_from_idx(op, (O.AV * n)(*oarr), n, oi)
        assert x.int_value() == y.v.to_int() and x.offset == y.cell
        gind = pr.nat.idx_to_indicator(gi, n); oind = (O.AV * n)(); L.orc_idx_to_indicator(op, oi, n, oind)
        assert [g.offset for g in gind] == [o.cell for o in oind]
        w = 3
        g2d = [[garr[(i + j) % n] for j in range(w)] for i in range(n)]
        o2d = (O.AV * (n * w))(*[oarr[(i + j) % n] for i in range(n) for j in range(w)])
        gout = pr.nat.select_array_by_indicator(g2d, gind); oout = (O.AV * w)(); L.orc_select_array_by_indicator(op, o2d, n, w, oind, oout)
        assert [g.int_value() for g in gout] == [o.v.to_int() for o in oout]
    for v, bits in [(small[0], 64), (5, 4), (0, 1), (P - 1, 64), ((1 << 252) + 99, 253)]:
        (g, o) = pr.wit(v)
        gb_ = pr.nat.num_to_bits(g, bits); ob_ = (O.AV * bits)(); L.orc_num_to_bits(op, o, bits, ob_)
        assert [x.offset for x in gb_] == [y.cell for y in ob_] and [x.int_value() for x in gb_] == [(v >> i) & 1 for i in range(bits)]
        x, y = pr.nat.bits_to_num(gb_), L.orc_bits_to_num(op, ob_, bits); assert x.int_value() == y.v.to_int() == v and x.offset == y.cell
    for v in wide[:4] + small[:2]:
        (g, o) = pr.wit(v)
        gl_ = pr.nat.decompose_le(g, 56, 5); ol_ = (O.AV * 5)(); L.orc_decompose_le(op, o, 56, 5, ol_)
        assert [x.offset for x in gl_] == [y.cell for y in ol_] and [x.int_value() for x in gl_] == [(v >> (56 * i)) & ((1 << 56) - 1) for i in range(5)]
    for n in (1, 2, 3):
        ls = [pr.wit(rnd.choice(small)) for _ in range(n)]
        x, y = pr.nat.limbs_to_num([g for g, _ in ls], 64), L.orc_limbs_to_num(op, (O.AV * n)(*[o for _, o in ls]), n, 64)
        assert x.int_value() == y.v.to_int() and x.offset == y.cell
    for v, bits in [(12345, 48), ((1 << 48) - 1, 48), (small[1], 64), (small[2] >> 8, 56), (1, 1), (3, 2), (small[3], 84), ((1 << 100) + 7, 105), (77, 21), (1 << 20, 22)]:
        (g, o) = pr.wit(v)
        pr.nat.range_check(g, bits); L.orc_range_check(op, o, bits)
    for v, bound in [(small[4] % P, P), (5, 1000), (P - 1, P), (0, P), ((1 << 40) + 3, (1 << 41) + 11)]:
        (g, o) = pr.wit(v)
        pr.nat.check_less_than_safe(g, bound); L.orc_check_less_than_safe(op, o, bound)
    pr.check()


def test_native_ops_off_domain(h2w_api, oracle):
    """Inputs the verifier never produces but the NativeChip API accepts (witness generation does not validate them; only the
    prover would fail): non-boolean selectors and "bits", values larger than the range they are checked against, non-indicator
    indicators.  The product either emits the reference's cells (oracle parity) or refuses loudly before emitting anything."""
    rnd = random.Random(12)
    pr = Pair(h2w_api, oracle)
    L, op, O = pr.L, pr.octx.p, oracle
    H2WError = h2w_api.H2WError

    def both(gfn, ofn):
        """run the op on both sides; if the product refuses, it must not have appended cells and the oracle side is skipped"""
        n0 = pr.ctx.num_cells()
        try:
            g = gfn()
        except H2WError:
            assert pr.ctx.num_cells() == n0
            return None
        return g, ofn()

    vals = [2, 3, R - 1, (1 << 64) + 5, rnd.randrange(R), rnd.randrange(1 << 128)]
    for sel in vals:
        (ga, oa), (gb, ob), (gs, os_) = pr.wit(rnd.randrange(R)), pr.wit(rnd.randrange(1 << 64)), pr.wit(sel)
        r = both(lambda: pr.nat.select(ga, gb, gs), lambda: L.orc_select(op, oa, ob, os_))
        if r: assert r[0].int_value() == r[1].v.to_int() and r[0].offset == r[1].cell
    for n in (3, 8):
        bits = [pr.wit(rnd.choice([0, 1, 2, 5, R - 1])) for _ in range(n)]
        r = both(lambda: pr.nat.bits_to_num([g for g, _ in bits]), lambda: L.orc_bits_to_num(op, (O.AV * n)(*[o for _, o in bits]), n))
        if r: assert r[0].int_value() == r[1].v.to_int() and r[0].offset == r[1].cell
    for v, nb in [(1 << 70, 64), ((1 << 20) + 1, 8), (R - 1, 16)]:
        (g, o) = pr.wit(v)
        ob_ = (O.AV * nb)()
        r = both(lambda: pr.nat.num_to_bits(g, nb), lambda: L.orc_num_to_bits(op, o, nb, ob_))
        if r: assert [x.int_value() for x in r[0]] == [ob_[i].v.to_int() for i in range(nb)]
    for v, nb in [(1 << 50, 48), ((1 << 64) - 1, 48), (rnd.randrange(1 << 128), 84), (rnd.randrange(R), 64), (R - 1, 21)]:
        (g, o) = pr.wit(v)
        both(lambda: pr.nat.range_check(g, nb), lambda: L.orc_range_check(op, o, nb))
    for v, bound in [(P, P), (P + 5, P), ((1 << 64) - 1, P), (1 << 90, P), (1000, 5), (rnd.randrange(R), P)]:
        (g, o) = pr.wit(v)
        both(lambda: pr.nat.check_less_than_safe(g, bound), lambda: L.orc_check_less_than_safe(op, o, bound))
    for n in (4,):
        arr = [pr.wit(rnd.randrange(R)) for _ in range(n)]
        ind = [pr.wit(rnd.choice([0, 1, 2, R - 1])) for _ in range(n)]
        oout = (O.AV * 1)()
        r = both(lambda: pr.nat.select_array_by_indicator([[g] for g, _ in arr], [g for g, _ in ind]),
                 lambda: L.orc_select_array_by_indicator(op, (O.AV * n)(*[o for _, o in arr]), n, 1, (O.AV * n)(*[o for _, o in ind]), oout))
        if r: assert r[0][0].int_value() == oout[0].v.to_int()
    for v in [rnd.randrange(R), R - 1, 1 << 253]:
        (g, o) = pr.wit(v)
        ol_ = (O.AV * 5)()
        r = both(lambda: pr.nat.decompose_le(g, 56, 5), lambda: L.orc_decompose_le(op, o, 56, 5, ol_))
        if r: assert [x.int_value() for x in r[0]] == [ol_[i].v.to_int() for i in range(5)]
    pr.check()
