"""Pins the C oracle's arithmetic against plain Python big-int arithmetic (independent restatement):
BN254 Fr Montgomery product / inverse, Goldilocks field, the u128 divmod hint, and the worked 65-cell layout of
GoldilocksChip::mul (reference field/goldilocks/base.rs:286-368)."""
import ctypes as C
import random

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
P = 2**64 - 2**32 + 1


def test_constants(oracle):
    m = oracle.Fr(); oracle.lib().orc_fr_modulus(C.byref(m))
    assert m.to_int() == R
    assert R == 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
    # plonky2 POWER_OF_TWO_GENERATOR quoted in SURVEY App. B must be 7^((p-1)/2^32)
    assert pow(7, (P - 1) >> 32, P) == 1753635133440165772
    assert oracle.lib().orc_glf_primitive_root_of_unity(32) == 1753635133440165772
    for k in (1, 4, 11, 21):
        g = oracle.lib().orc_glf_primitive_root_of_unity(k)
        assert pow(g, 1 << k, P) == 1 and pow(g, 1 << (k - 1), P) == P - 1


def test_fr_mul_inv(oracle):
    rnd = random.Random(1)
    L = oracle.lib()
    for _ in range(200):
        a, b = rnd.randrange(R), rnd.randrange(R)
        out = oracle.Fr()
        L.orc_fr_mul(C.byref(oracle.Fr.from_int(a)), C.byref(oracle.Fr.from_int(b)), C.byref(out))
        assert out.to_int() == a * b % R
    for a in (1, 2, R - 1, rnd.randrange(1, R)):
        out = oracle.Fr()
        L.orc_fr_inv(C.byref(oracle.Fr.from_int(a)), C.byref(out))
        assert out.to_int() * a % R == 1


def test_gl_field(oracle):
    rnd = random.Random(2)
    L = oracle.lib()
    for _ in range(200):
        a, b = rnd.randrange(P), rnd.randrange(P)
        assert L.orc_glf_mul(a, b) == a * b % P
    for a in (1, 7, P - 1, rnd.randrange(1, P)):
        assert L.orc_glf_inv(a) * a % P == 1


def limbs(x, L=21, n=4):
    return [(x >> (L * i)) & ((1 << L) - 1) for i in range(n)]


def rc84(x):
    l = limbs(x)
    return [l[0], l[1], 1 << 21, l[0] + (l[1] << 21), l[2], 1 << 42, x & ((1 << 63) - 1), l[3], 1 << 63, x]


def load_witness_cells(x):
    B = 1 << 84
    return [x] + rc84(x) + [x + B - P, P, 1, x + B, (-B) % R, 1, x] + rc84(x + B - P)


def test_gl_mul_cell_layout(oracle):
    """The 65 cells of GoldilocksChip::mul(a,b) at lookup_bits 21, from first principles."""
    rnd = random.Random(3)
    L = oracle.lib()
    for _ in range(50):
        a, b = rnd.randrange(P), rnd.randrange(P)
        ctx = oracle.Ctx(21)
        aw, bw = L.orc_gl_load_constant(ctx.p, a), L.orc_gl_load_constant(ctx.p, b)
        out = L.orc_gl_mul(ctx.p, aw, bw)
        assert out.v.to_int() == a * b % P
        raw = ctx.advice_bytes()
        cells = [int.from_bytes(raw[i * 32:(i + 1) * 32], "little") for i in range(ctx.num_cells())]
        v = a * b; q, r = divmod(v, P)
        expect = [a, b] + [0, a, b, v] + load_witness_cells(q) + load_witness_cells(r) + [P] + [r, q, P, v]
        assert cells == expect
        ctx.close()
