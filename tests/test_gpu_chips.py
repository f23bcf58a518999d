"""GPU parity of the reference's higher chips driven through the eager C-ABI (csrc/abi_backend.cpp): each test mirrors
one of the reference's own unit tests and compares the GPU-expanded advice stream with the oracle byte for byte.

  extension.rs:473-493  test_goldilocks_extension_chip      hash/poseidon/permutation.rs:325-347        test_permute
  hash/poseidon_bn254/permutation.rs:266-301 test_permute    hash/*/hash.rs  test_hash_no_pad / test_hash_two_to_one
  merkle/mod.rs:136-265 test_verify_proof_to_cap / test_verify_proof      stark/mod.rs:405-518 test_fibonacci_stark_{gl,bn254}
"""
import ctypes as C
import random

import pytest

pytestmark = pytest.mark.gpu

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
P = 2**64 - 2**32 + 1


class Pair:
    def __init__(self, h2w, h2w_api, oracle, lookup_bits=21):
        self.H, self.api, self.O = h2w, h2w_api, oracle
        self.ctx = h2w_api.Context(lookup_bits); self.L = h2w.lib(); self.p = self.ctx.p
        self.octx = oracle.Ctx(lookup_bits); self.OL = oracle.lib(); self.op = self.octx.p

    def gl_const(self, v):
        g = self.H.Assigned(); assert self.L.h2w_gl_load_constant(self.p, v, C.byref(g)) == 0
        return g, self.OL.orc_gl_load_constant(self.op, v)

    def gl_wit(self, v):
        g = self.H.Assigned(); assert self.L.h2w_gl_load_witness(self.p, v, C.byref(g)) == 0
        return g, self.OL.orc_gl_load_witness(self.op, v)

    def fr_const(self, v):
        g = self.H.Assigned(); assert self.L.h2w_load_constant(self.p, C.byref(self.H.Fr.from_int(v)), C.byref(g)) == 0
        return g, self.OL.orc_load_constant(self.op, self.O.Fr.from_int(v))

    def garr(self, items): return (self.H.Assigned * len(items))(*items)
    def oarr(self, items): return (self.O.AV * len(items))(*items)

    def check(self):
        assert self.ctx.num_cells() == self.octx.num_cells()
        got, want = self.ctx.advice_bytes(), self.octx.advice_bytes()
        if got != want:
            for i in range(self.octx.num_cells()):
                if got[i * 32:(i + 1) * 32] != want[i * 32:(i + 1) * 32]:
                    raise AssertionError(f"cell {i} of {self.octx.num_cells()} differs")
        self.ctx.close(); self.octx.close()


def test_goldilocks_extension_chip(h2w, h2w_api, oracle):
    rnd = random.Random(21); pr = Pair(h2w, h2w_api, oracle)
    for _ in range(20):
        a = [pr.gl_const(rnd.randrange(P)) for _ in range(2)]; b = [pr.gl_const(rnd.randrange(1, P)) for _ in range(2)]
        ga, oa = pr.garr([x[0] for x in a]), pr.oarr([x[1] for x in a]); gb, ob = pr.garr([x[0] for x in b]), pr.oarr([x[1] for x in b])
        go, oo = (h2w.Assigned * 2)(), (oracle.AV * 2)()
        assert pr.L.h2w_chip_ext_op(pr.p, 2, ga, gb, None, go) == 0; pr.OL.orc_ext_mul(pr.op, oa, ob, oo)     # mul (the reference's test)
        assert [g.int_value() for g in go] == [o.v.to_int() for o in oo]
        assert pr.L.h2w_chip_ext_op(pr.p, 4, gb, None, None, go) == 0; pr.OL.orc_ext_inv(pr.op, ob, oo)        # inv
        assert pr.L.h2w_chip_ext_op(pr.p, 5, ga, gb, None, go) == 0; pr.OL.orc_ext_div(pr.op, oa, ob, oo)      # div
        assert [g.int_value() for g in go] == [o.v.to_int() for o in oo]
    pr.check()


def test_gl_poseidon_permute(h2w, h2w_api, oracle, consts):
    ko, kh = consts; rnd = random.Random(22); pr = Pair(h2w, h2w_api, oracle)
    for _ in range(3):
        st = [pr.gl_const(rnd.randrange(P)) for _ in range(12)]
        go, oo = (h2w.Assigned * 12)(), (oracle.AV * 12)()
        assert pr.L.h2w_chip_gl_poseidon_permute(pr.p, C.byref(kh), pr.garr([x[0] for x in st]), go) == 0
        pr.OL.orc_gl_poseidon_permute(pr.op, C.byref(ko), pr.oarr([x[1] for x in st]), oo)
        assert [g.int_value() for g in go] == [o.v.to_int() for o in oo]
    pr.check()


def test_bn254_poseidon_permute(h2w, h2w_api, oracle, consts):
    ko, kh = consts; rnd = random.Random(23); pr = Pair(h2w, h2w_api, oracle)
    for _ in range(10):
        st = [pr.fr_const(rnd.randrange(R)) for _ in range(4)]
        go, oo = (h2w.Assigned * 4)(), (oracle.AV * 4)()
        assert pr.L.h2w_chip_bn_poseidon_permute(pr.p, C.byref(kh), pr.garr([x[0] for x in st]), go) == 0
        pr.OL.orc_bn_poseidon_permute(pr.op, C.byref(ko), pr.oarr([x[1] for x in st]), oo)
        assert [g.int_value() for g in go] == [o.v.to_int() for o in oo]
    pr.check()


@pytest.mark.parametrize("mode", [0, 1])
def test_hash_no_pad_and_two_to_one(h2w, h2w_api, oracle, consts, mode):
    ko, kh = consts; rnd = random.Random(24 + mode); pr = Pair(h2w, h2w_api, oracle)
    hashes = []
    for n in (1, 3, 4, 5, 9, 20):
        pre = [pr.gl_wit(rnd.randrange(P)) for _ in range(n)]
        go, oo = (h2w.Assigned * 4)(), (oracle.AV * 4)()
        assert pr.L.h2w_chip_hash_no_pad(pr.p, C.byref(kh), mode, pr.garr([x[0] for x in pre]), n, go) == 0
        pr.OL.orc_hash_no_pad(pr.op, C.byref(ko), mode, pr.oarr([x[1] for x in pre]), n, oo)
        assert [g.int_value() for g in go] == [o.v.to_int() for o in oo]
        hashes.append((list(go), list(oo)))
    for (g1, o1), (g2, o2) in zip(hashes[:-1], hashes[1:]):
        go, oo = (h2w.Assigned * 4)(), (oracle.AV * 4)()
        assert pr.L.h2w_chip_two_to_one(pr.p, C.byref(kh), mode, pr.garr(g1), pr.garr(g2), go) == 0
        pr.OL.orc_two_to_one(pr.op, C.byref(ko), mode, pr.oarr(o1), pr.oarr(o2), oo)
        assert [g.int_value() for g in go] == [o.v.to_int() for o in oo]
    pr.check()


@pytest.mark.parametrize("mode,cap_height", [(0, 1), (1, 1), (1, 0), (0, 2)])
def test_verify_proof_to_cap(h2w, h2w_api, oracle, consts, mode, cap_height):
    """8 leaves x 20 elements like merkle/mod.rs:136-200; siblings/cap are random (witness generation does not branch on validity)."""
    ko, kh = consts; rnd = random.Random(26 + mode + cap_height); pr = Pair(h2w, h2w_api, oracle)
    hw = 4 if mode == 0 else 1; depth = 3; n_sib = depth - cap_height; n_cap = 1 << cap_height
    def hash_wire():
        return [pr.gl_const(rnd.randrange(P)) for _ in range(4)] if mode == 0 else [pr.fr_const(rnd.randrange(R))]
    for leaf_len, idx in ((20, 5), (3, 0), (4, 7)):
        leaf = [pr.gl_wit(rnd.randrange(P)) for _ in range(leaf_len)]
        bits = [pr.gl_const((idx >> i) & 1) for i in range(depth)]
        cap_index = pr.gl_const(idx >> n_sib)
        cap = [w for _ in range(n_cap) for w in hash_wire()]; sib = [w for _ in range(n_sib) for w in hash_wire()]
        assert pr.L.h2w_chip_merkle_verify(pr.p, C.byref(kh), mode, pr.garr([x[0] for x in leaf]), leaf_len, pr.garr([x[0] for x in bits]), depth,
                                           C.byref(cap_index[0]), pr.garr([x[0] for x in cap]), n_cap, pr.garr([x[0] for x in sib]) if sib else None, n_sib) == 0
        pr.OL.orc_merkle_verify(pr.op, C.byref(ko), mode, pr.oarr([x[1] for x in leaf]), leaf_len, pr.oarr([x[1] for x in bits]), depth,
                                cap_index[1], pr.oarr([x[1] for x in cap]), n_cap, pr.oarr([x[1] for x in sib]) if sib else None, n_sib)
    pr.check()


@pytest.mark.parametrize("mode", [1, 0])
def test_fibonacci_stark_through_eager_boundary(h2w, h2w_api, oracle, consts, mode):
    """The reference's end-to-end test flow (stark/mod.rs:405-518) with every chip call going through the NativeChip-level
    C-ABI; num_rows = 1 << 3 like the reference's current tests, plus a shape with one fold step."""
    ko, kh = consts
    for d, q, rb in ((3, 2, 1), (7, 2, 2)):
        sh = h2w.fibonacci_shape(d, q, rate_bits=rb, hash_mode=mode); osh = oracle.fibonacci_shape(d, q, rate_bits=rb, hash_mode=mode)
        proof = oracle.synth_proof(osh, 77 + d)
        ctx = h2w_api.Context(21); octx = oracle.Ctx(21)
        assert h2w.lib().h2w_chip_verify_stark(ctx.p, C.byref(sh), C.byref(kh), proof) == 0, h2w.last_error()
        assert oracle.verify_stark(octx, osh, ko, proof) == 0
        assert ctx.num_cells() == octx.num_cells()
        assert ctx.advice_bytes() == octx.advice_bytes()
        # the same context for the next proof (h2w_ctx_reset: its host memory is kept, everything else is as new)
        proof2 = oracle.synth_proof(osh, 177 + d); octx2 = oracle.Ctx(21)
        ctx.reset()
        assert ctx.num_cells() == 0
        assert h2w.lib().h2w_chip_verify_stark(ctx.p, C.byref(sh), C.byref(kh), proof2) == 0, h2w.last_error()
        assert oracle.verify_stark(octx2, osh, ko, proof2) == 0
        assert ctx.advice_bytes() == octx2.advice_bytes()
        ctx.close(); octx.close(); octx2.close()


def test_permutations_reproduce_published_known_answers(h2w, h2w_api, oracle, published):
    """Both permutation chips through the eager C-ABI with the published parameter sets (h2w_poseidon_published): the output
    wires carry plonky2's published permute([0;12]) / permute([0..11]) / permute([-1;12]) vectors and circomlib's
    poseidon([1,2,3]) (tests/golden/poseidon_published.json), and the GPU-expanded advice equals the oracle's."""
    import json, os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "poseidon_published.json")))
    ko, kh = published; pr = Pair(h2w, h2w_api, oracle)
    for v in gold["goldilocks_w12"]["permutation_vectors"][:4]:
        st = [pr.gl_const(int(x, 16)) for x in v["in"]]
        go, oo = (h2w.Assigned * 12)(), (oracle.AV * 12)()
        assert pr.L.h2w_chip_gl_poseidon_permute(pr.p, C.byref(kh), pr.garr([x[0] for x in st]), go) == 0
        pr.OL.orc_gl_poseidon_permute(pr.op, C.byref(ko), pr.oarr([x[1] for x in st]), oo)
        assert [hex(g.int_value()) for g in go] == v["out"]
    for v in gold["bn254_t4"]["permutation_vectors"]:
        st = [pr.fr_const(int(x, 16)) for x in v["in"]]
        go, oo = (h2w.Assigned * 4)(), (oracle.AV * 4)()
        assert pr.L.h2w_chip_bn_poseidon_permute(pr.p, C.byref(kh), pr.garr([x[0] for x in st]), go) == 0
        pr.OL.orc_bn_poseidon_permute(pr.op, C.byref(ko), pr.oarr([x[1] for x in st]), oo)
        assert [hex(g.int_value()) for g in go] == v["out"]
    pr.check()


OPS = {"add": 0, "sub": 1, "mul": 2, "mul_add": 3, "div": 4, "inv": 5, "ext_mul": 6, "ext_inv": 7, "ext_div": 8}


@pytest.mark.parametrize("op", sorted(OPS))
@pytest.mark.parametrize("lookup_bits", [21, 13])
def test_batched_chip_ops_and_device_status_words(h2w, oracle, op, lookup_bits):
    """h2w_chipbatch_*: n independent instances of one GoldilocksChip / QuadExtChip op on the device (operands loaded as witnesses, then the
    op: the reference's test_mul shape, base.rs:476-495), every instance's cells equal to the oracle's.  Where the reference PANICS -
    GoldilocksChip::div by zero (base.rs:379), extension inverse of zero (extension.rs:327) - the device status word of that instance is
    1 / 2, the oracle's context reports the failed assertion, and both sides emit the cells of the substituted operand."""
    import numpy as np
    import torch
    L = h2w.lib(); OL = oracle.lib()
    h = L.h2w_chipbatch_new(OPS[op], lookup_bits, 0)
    assert h, h2w.last_error()
    nw, nc = int(L.h2w_chipbatch_num_operands(h)), int(L.h2w_chipbatch_num_cells(h))
    rnd = random.Random(sum(map(ord, op)))
    n = 200
    ops = [[rnd.randrange(P) for _ in range(nw)] for _ in range(n)]
    ops[0] = [0] * nw; ops[1] = [P - 1] * nw; ops[2] = [1] * nw                      # edges; all-zero operands hit the panics of div / inv
    if op in ("div", "ext_div"):
        ops[3] = [rnd.randrange(1, P) for _ in range(nw // 2)] + [0] * (nw // 2)      # non-zero numerator over a zero denominator
    d_ops = torch.tensor(np.array(ops, dtype=np.uint64).view(np.int64).reshape(-1), dtype=torch.int64, device="cuda")
    advice = torch.zeros(n * nc * 32, dtype=torch.uint8, device="cuda")
    status = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    assert L.h2w_chipbatch_run(h, d_ops.data_ptr(), n, advice.data_ptr(), status.data_ptr(), 0) == 0, h2w.last_error()
    torch.cuda.synchronize()
    got = advice.cpu().numpy().tobytes(); st = status.cpu().tolist()
    for i, w in enumerate(ops):
        ctx = oracle.Ctx(lookup_bits)
        av = [OL.orc_gl_load_witness(ctx.p, x) for x in w]
        if op in ("add", "sub", "mul", "div"):
            getattr(OL, "orc_gl_" + op)(ctx.p, av[0], av[1])
        elif op == "mul_add":
            OL.orc_gl_mul_add(ctx.p, av[0], av[1], av[2])
        elif op == "inv":
            OL.orc_gl_inv(ctx.p, av[0])
        else:
            a = (oracle.AV * 2)(av[0], av[1]); out = (oracle.AV * 2)()
            if op == "ext_inv":
                OL.orc_ext_inv(ctx.p, a, out)
            else:
                b = (oracle.AV * 2)(av[2], av[3])
                (OL.orc_ext_mul if op == "ext_mul" else OL.orc_ext_div)(ctx.p, a, b, out)
        assert ctx.num_cells() == nc
        assert got[i * nc * 32:(i + 1) * nc * 32] == ctx.advice_bytes(), (op, i)
        panics = bool(ctx.error())
        zero_gl = (op == "div" and w[1] == 0) or (op == "inv" and w[0] == 0)
        zero_ext = (op == "ext_inv" and w[0] == 0 and w[1] == 0) or (op == "ext_div" and w[2] == 0 and w[3] == 0)
        assert panics == (zero_gl or zero_ext), (op, i, ctx.error())
        assert st[i] == (1 if zero_gl else 2 if zero_ext else 0), (op, i, st[i])
        ctx.close()
    assert any(st) == (op in ("div", "inv", "ext_inv", "ext_div"))
    L.h2w_chipbatch_free(h)
