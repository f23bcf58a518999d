"""GPU parity of the synthetic-proof generator (csrc/prover.hip) against the oracle's native FRI prover
(oracle/prover.inc) on the same committed polynomials: every word of the flat proof, both hash modes, shapes with zero, one and
several fold steps; and end to end: GPU-generated proofs through the GPU witness generator satisfy every gate and lookup
(the restated MockProver on the device) and reproduce the oracle's advice stream."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu

SHAPES = [
    # degree_bits, queries, rate_bits, cap_height
    (6, 3, 1, 2),      # no fold step
    (9, 2, 1, 2),      # one fold step
    (7, 2, 2, 4),      # rate_bits 2, cap of 16
    (13, 5, 1, 4),     # two fold steps, NTT larger than one LDS tile
]


def gpu_prove(h2w, h2w_api, oracle, kh, sh, osh, seed):
    import numpy as np
    import torch
    coefs, pis = oracle.prove_fri_inputs(osh, seed)
    pr = h2w_api.Prover(sh, kh)
    d_coefs = torch.from_numpy(np.frombuffer(coefs, dtype=np.int64).copy()).cuda()
    assert d_coefs.numel() == pr.num_polys << sh.degree_bits
    d_proof = torch.zeros(pr.proof_words, dtype=torch.int64, device="cuda")
    pr.prove(d_coefs.data_ptr(), list(pis)[:sh.n_pis], d_proof.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t = pr.timing(); pr.close()
    return coefs, pis, d_proof, t


@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("shape", SHAPES)
def test_proof_words_equal_the_oracle_prover(h2w, h2w_api, oracle, published, mode, shape):
    import numpy as np
    ko, kh = published
    d, q, rb, cap = shape
    sh = h2w.fibonacci_shape(d, q, rate_bits=rb, cap_height=cap, hash_mode=mode)
    osh = oracle.fibonacci_shape(d, q, rate_bits=rb, cap_height=cap, hash_mode=mode)
    coefs, pis, d_proof, _ = gpu_prove(h2w, h2w_api, oracle, kh, sh, osh, 0xF1B00000 + d)
    want = np.frombuffer(oracle.prove_fri_coef(osh, ko, coefs, pis), dtype=np.uint64)
    got = d_proof.cpu().numpy().view(np.uint64)
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, f"{len(bad)} of {len(want)} proof words differ, first at {bad[:8]}"


def test_synthetic_constant_tables(h2w, h2w_api, oracle, consts):
    """Seeded full-width tables in every slot (the fast partial-round tables are then unrelated to the MDS matrix: the device
    permutation must follow the reference's round structure, not an equivalent one)."""
    import numpy as np
    ko, kh = consts
    for mode in (1, 0):
        sh = h2w.fibonacci_shape(9, 2, rate_bits=1, cap_height=2, hash_mode=mode)
        osh = oracle.fibonacci_shape(9, 2, rate_bits=1, cap_height=2, hash_mode=mode)
        coefs, pis, d_proof, _ = gpu_prove(h2w, h2w_api, oracle, kh, sh, osh, 77)
        want = np.frombuffer(oracle.prove_fri_coef(osh, ko, coefs, pis), dtype=np.uint64)
        assert (d_proof.cpu().numpy().view(np.uint64) == want).all()


@pytest.mark.parametrize("mode", [1, 0])
def test_gpu_proof_into_gpu_witness(h2w, h2w_api, oracle, published, mode):
    """prover -> witness generator without leaving the device: a valid instance, so every gate and lookup holds; the advice
    equals the oracle's on the same proof."""
    import torch
    ko, kh = published
    sh = h2w.fibonacci_shape(9, 3, rate_bits=1, cap_height=2, hash_mode=mode)
    osh = oracle.fibonacci_shape(9, 3, rate_bits=1, cap_height=2, hash_mode=mode)
    _, _, d_proof, _ = gpu_prove(h2w, h2w_api, oracle, kh, sh, osh, 4242)
    plan = h2w_api.Plan(sh, kh)
    assert plan.proof_words == d_proof.numel()
    advice = torch.zeros(plan.num_cells * 32, dtype=torch.uint8, device="cuda")
    ws = torch.zeros(plan.workspace_bytes(1), dtype=torch.uint8, device="cuda")
    plan.run(d_proof.data_ptr(), 1, advice.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert plan.status(ws.data_ptr(), 1) == [0]
    assert plan.check_constraints(advice.data_ptr(), 1) == (0, 0)
    ctx = oracle.Ctx(21, witness_gen_only=False)
    words = (C.c_uint64 * plan.proof_words).from_buffer_copy(d_proof.cpu().numpy().tobytes())
    assert oracle.verify_stark(ctx, osh, ko, words) == 0
    mp = ctx.mock_prover()
    assert mp["bad"] == 0 and mp["semantic_failed"] == 0, mp            # Merkle roots, fold consistency, final polynomial, PoW
    assert advice.cpu().numpy().tobytes() == ctx.advice_bytes()
    ctx.close(); plan.close()


@pytest.mark.parametrize("case", ["cfg2_gl", "cfg2_bn254", "cfg3_gl"])
def test_full_size_proofs_equal_the_oracle_provers_digest(h2w, h2w_api, oracle, published, case):
    """BASELINE.json's full sizes: the oracle prover needs 15-90 s per proof there, so its output is committed as a sha256
    (tests/golden/prover_digests.json, made by tools/make_golden_prover.py from the same seeded inputs)."""
    import hashlib, json, os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "prover_digests.json")))[case]
    ko, kh = published
    sh = h2w.fibonacci_shape(gold["degree_bits"], gold["queries"], rate_bits=gold["rate_bits"], hash_mode=gold["hash_mode"])
    osh = oracle.fibonacci_shape(gold["degree_bits"], gold["queries"], rate_bits=gold["rate_bits"], hash_mode=gold["hash_mode"])
    _, _, d_proof, _ = gpu_prove(h2w, h2w_api, oracle, kh, sh, osh, gold["seed"])
    assert d_proof.numel() == gold["proof_words"]
    assert hashlib.sha256(d_proof.cpu().numpy().tobytes()).hexdigest() == gold["sha256"]


def test_config3_bn254_proof_is_a_valid_fri_instance(h2w, h2w_api, oracle, published):
    """BASELINE.json configs[2] with PoseidonBN254 caps (the oracle prover would need ~12 minutes): the GPU-generated proof is
    accepted by the restated verifier - every gate, lookup and copy constraint of the 28.6 M-cell witness holds, i.e. all Merkle
    paths verify against the caps, every fold step is consistent, the final polynomial matches and the PoW response has its
    leading zeros - and the GPU witness of it equals the oracle's."""
    import torch
    ko, kh = published
    sh = h2w.fibonacci_shape(20, 28, rate_bits=1, hash_mode=1); osh = oracle.fibonacci_shape(20, 28, rate_bits=1, hash_mode=1)
    _, _, d_proof, _ = gpu_prove(h2w, h2w_api, oracle, kh, sh, osh, 0xF1B00003)
    plan = h2w_api.Plan(sh, kh)
    advice = torch.zeros(plan.num_cells * 32, dtype=torch.uint8, device="cuda"); ws = torch.zeros(plan.workspace_bytes(1), dtype=torch.uint8, device="cuda")
    plan.run(d_proof.data_ptr(), 1, advice.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert plan.status(ws.data_ptr(), 1) == [0] and plan.check_constraints(advice.data_ptr(), 1) == (0, 0)
    ctx = oracle.Ctx(21, witness_gen_only=False)
    words = (C.c_uint64 * plan.proof_words).from_buffer_copy(d_proof.cpu().numpy().tobytes())
    assert oracle.verify_stark(ctx, osh, ko, words) == 0
    mp = ctx.mock_prover()
    assert mp["bad"] == 0 and mp["semantic_failed"] == 0, mp
    assert advice.cpu().numpy().tobytes() == ctx.advice_bytes()
    ctx.close(); plan.close()


@pytest.mark.parametrize("mode", [1, 0])
def test_lockstep_batch_equals_proof_by_proof(h2w, h2w_api, oracle, published, mode):
    """h2w_prove_fri_batch: five instances with different polynomials through every kernel together; each proof equals the
    oracle prover's for its own inputs (two fold steps, so the per-proof betas / PoW states / query indices all differ)."""
    import numpy as np
    import torch
    ko, kh = published
    sh = h2w.fibonacci_shape(13, 4, rate_bits=1, cap_height=3, hash_mode=mode); osh = oracle.fibonacci_shape(13, 4, rate_bits=1, cap_height=3, hash_mode=mode)
    seeds = [900 + i for i in range(5)]
    ins = [oracle.prove_fri_inputs(osh, sd) for sd in seeds]
    pr = h2w_api.Prover(sh, kh)
    coefs = torch.from_numpy(np.concatenate([np.frombuffer(c, dtype=np.int64) for c, _ in ins])).cuda()
    pis = [int(x) for _, p_ in ins for x in list(p_)[:sh.n_pis]]
    proofs = torch.zeros(len(seeds) * pr.proof_words, dtype=torch.int64, device="cuda")
    pr.prove_batch(coefs.data_ptr(), pis, proofs.data_ptr(), len(seeds), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = proofs.cpu().numpy().view(np.uint64).reshape(len(seeds), pr.proof_words)
    for i, (c, p_) in enumerate(ins):
        want = np.frombuffer(oracle.prove_fri_coef(osh, ko, c, p_), dtype=np.uint64)
        assert (got[i] == want).all(), f"proof {i} of the batch differs"
    pr.close()


def test_full_width_mds_constants(h2w, h2w_api, oracle, consts):
    """MDS entries of 64 bits (no published table has them, a caller's may): the generic MDS layer, not the two-accumulator one."""
    import numpy as np
    ko0, _ = consts
    ko = oracle.Consts.from_buffer_copy(bytes(ko0))
    for i in range(12):
        ko.mds_circ[i] = (0x9E3779B97F4A7C15 * (i + 1)) % (2**64 - 2**32 + 1)
        ko.mds_diag[i] = (0xD1B54A32D192ED03 * (i + 3)) % (2**64 - 2**32 + 1)
    kh = h2w.PoseidonConsts.from_buffer_copy(bytes(ko))
    sh = h2w.fibonacci_shape(9, 2, rate_bits=1, cap_height=2, hash_mode=0); osh = oracle.fibonacci_shape(9, 2, rate_bits=1, cap_height=2, hash_mode=0)
    coefs, pis, d_proof, _ = gpu_prove(h2w, h2w_api, oracle, kh, sh, osh, 78)
    want = np.frombuffer(oracle.prove_fri_coef(osh, ko, coefs, pis), dtype=np.uint64)
    assert (d_proof.cpu().numpy().view(np.uint64) == want).all()


def _witness_of(h2w_api, plan, d_proofs, n):
    import torch
    advice = torch.zeros(n * plan.num_cells * 32, dtype=torch.uint8, device="cuda")
    ws = torch.zeros(plan.workspace_bytes(n), dtype=torch.uint8, device="cuda")
    plan.run(d_proofs.data_ptr(), n, advice.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert plan.status(ws.data_ptr(), n) == [0] * n
    return advice


def test_config4_pipeline_on_the_device(h2w, h2w_api, oracle, published):
    """BASELINE.json configs[3] at 1/16 scale on one GPU: 16 independent 2^16-row proofs (28 queries, rate_bits 2) generated in one
    lockstep batch, witnessed in one batch, every gate and lookup checked on the device; two of the 16 advice streams are compared
    with the oracle's, byte for byte (the proofs never leave the device except for that comparison)."""
    import numpy as np
    import torch
    ko, kh = published
    n = 16
    sh = h2w.fibonacci_shape(16, 28, rate_bits=2, hash_mode=1); osh = oracle.fibonacci_shape(16, 28, rate_bits=2, hash_mode=1)
    pr = h2w_api.Prover(sh, kh)
    g = torch.Generator(device="cuda"); g.manual_seed(404)
    coefs = torch.randint(0, 1 << 62, (n * pr.num_polys << 16,), dtype=torch.int64, device="cuda", generator=g)
    proofs = torch.zeros(n * pr.proof_words, dtype=torch.int64, device="cuda")
    pr.prove_batch(coefs.data_ptr(), [7, 8, 9] * n, proofs.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize(); pr.close()
    plan = h2w_api.Plan(sh, kh)
    advice = _witness_of(h2w_api, plan, proofs, n)
    assert plan.check_constraints(advice.data_ptr(), n) == (0, 0)
    host = proofs.cpu().numpy().view(np.uint64).reshape(n, plan.proof_words)
    assert len({host[i].tobytes() for i in range(n)}) == n                      # independent instances
    for i in (0, n - 1):
        ctx = oracle.Ctx(21)
        assert oracle.verify_stark(ctx, osh, ko, (C.c_uint64 * plan.proof_words).from_buffer_copy(host[i].tobytes())) == 0
        assert advice[i * plan.num_cells * 32:(i + 1) * plan.num_cells * 32].cpu().numpy().tobytes() == ctx.advice_bytes()
        ctx.close()
    plan.close()


def test_config4_full_size_streamed_with_digests(h2w, h2w_api, oracle, published):
    """BASELINE.json configs[3] at FULL size on one GPU: 256 independent 2^16-row proofs (28 queries, rate_bits 2, PoseidonBN254 caps),
    8 steps of 32: prover -> witness -> every gate and lookup checked on the device -> one h2w_advice_digest per proof, the advice
    buffer re-used by the next step (the 256 streams together are 191 GB).  Four sampled proofs' digests equal the checksum of the
    oracle's stream of the same proof, two of those streams are compared byte for byte, and all 256 digests are distinct."""
    import numpy as np
    import torch
    ko, kh = published
    steps, per = 8, 32
    sh = h2w.fibonacci_shape(16, 28, rate_bits=2, hash_mode=1); osh = oracle.fibonacci_shape(16, 28, rate_bits=2, hash_mode=1)
    pr = h2w_api.Prover(sh, kh); plan = h2w_api.Plan(sh, kh)
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda"); g.manual_seed(4040)
    proofs = torch.zeros(per * pr.proof_words, dtype=torch.int64, device="cuda")
    advice = torch.zeros(per * plan.num_cells * 32, dtype=torch.uint8, device="cuda")
    ws = torch.zeros(plan.workspace_bytes(per), dtype=torch.uint8, device="cuda")
    digests = torch.zeros(steps * per * 4, dtype=torch.int64, device="cuda")
    sampled = {(0, 0): True, (3, 17): False, (5, 9): False, (7, 31): True}        # (step, proof) -> compare every byte too
    for step in range(steps):
        coefs = torch.randint(0, 1 << 62, (per * pr.num_polys << 16,), dtype=torch.int64, device="cuda", generator=g)
        pr.prove_batch(coefs.data_ptr(), [step, 8, 9] * per, proofs.data_ptr(), per, st)
        plan.run(proofs.data_ptr(), per, advice.data_ptr(), ws.data_ptr(), st)
        torch.cuda.synchronize()
        assert plan.status(ws.data_ptr(), per) == [0] * per
        assert plan.check_constraints(advice.data_ptr(), per) == (0, 0)
        for i in range(per):
            plan.advice_digest(advice.data_ptr() + i * plan.num_cells * 32, plan.num_cells, digests.data_ptr() + (step * per + i) * 32, st)
        torch.cuda.synchronize()
        for (s_, i), every_byte in sampled.items():
            if s_ != step:
                continue
            words = proofs[i * plan.proof_words:(i + 1) * plan.proof_words].cpu().numpy().tobytes()
            ctx = oracle.Ctx(21); ctx.reserve(plan.num_cells)
            assert oracle.verify_stark(ctx, osh, ko, (C.c_uint64 * plan.proof_words).from_buffer_copy(words)) == 0
            want = ctx.advice_bytes(); ctx.close()
            got = [int(x) & 0xFFFFFFFFFFFFFFFF for x in digests[(step * per + i) * 4:(step * per + i + 1) * 4].cpu().tolist()]
            assert got == h2w_api.advice_digest_reference(want), (step, i)
            if every_byte:
                assert advice[i * plan.num_cells * 32:(i + 1) * plan.num_cells * 32].cpu().numpy().tobytes() == want
    d = digests.cpu().numpy().reshape(steps * per, 4)
    assert len({row.tobytes() for row in d}) == steps * per                       # 256 independent streams
    pr.close(); plan.close()


def test_streamed_goldilocks_caps_digest(h2w, h2w_api, oracle, published):
    """The streamed form of configs[3] with Goldilocks-Poseidon caps (87.8 G cells for the 256 proofs: SURVEY 7 "must stream with an
    on-device digest"): 343 M-cell streams generated into ONE re-used buffer, checked on the device and digested; the digest of one
    stream equals the checksum of the oracle's stream of the same proof (11 GB on the host), and the streams differ."""
    import torch
    ko, kh = published
    sh = h2w.fibonacci_shape(16, 28, rate_bits=2, hash_mode=0); osh = oracle.fibonacci_shape(16, 28, rate_bits=2, hash_mode=0)
    pr = h2w_api.Prover(sh, kh); plan = h2w_api.Plan(sh, kh)
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda"); g.manual_seed(4041)
    n = 3
    proof = torch.zeros(pr.proof_words, dtype=torch.int64, device="cuda")
    advice = torch.zeros(plan.num_cells * 32, dtype=torch.uint8, device="cuda")
    ws = torch.zeros(plan.workspace_bytes(1), dtype=torch.uint8, device="cuda")
    digests = torch.zeros(n * 4, dtype=torch.int64, device="cuda")
    keep = None
    for i in range(n):
        coefs = torch.randint(0, 1 << 62, (pr.num_polys << 16,), dtype=torch.int64, device="cuda", generator=g)
        pr.prove_batch(coefs.data_ptr(), [i, 2, 3], proof.data_ptr(), 1, st)
        plan.run(proof.data_ptr(), 1, advice.data_ptr(), ws.data_ptr(), st)
        torch.cuda.synchronize()
        assert plan.status(ws.data_ptr(), 1) == [0]
        assert plan.check_constraints(advice.data_ptr(), 1) == (0, 0)
        plan.advice_digest(advice.data_ptr(), plan.num_cells, digests.data_ptr() + i * 32, st)
        if i == 1:
            keep = proof.cpu().numpy().tobytes()
    torch.cuda.synchronize()
    got = [[int(x) & 0xFFFFFFFFFFFFFFFF for x in digests[4 * i:4 * i + 4].cpu().tolist()] for i in range(n)]
    assert len({tuple(x) for x in got}) == n
    ctx = oracle.Ctx(21); ctx.reserve(plan.num_cells)
    assert oracle.verify_stark(ctx, osh, ko, (C.c_uint64 * plan.proof_words).from_buffer_copy(keep)) == 0
    assert ctx.num_cells() == plan.num_cells
    assert got[1] == h2w_api.advice_digest_reference(ctx.advice_array())
    ctx.close(); pr.close(); plan.close()


@pytest.mark.parametrize("cfg", ["cfg3", "cfg5"])
def test_full_size_goldilocks_caps_digest(h2w, h2w_api, oracle, published, cfg):
    """The largest single streams of BASELINE.json with Goldilocks-Poseidon Merkle caps - configs[2]: 450 M cells = 14.4 GB, configs[4]'s shape: 1.33 G cells
    = 42.6 GB per proof - against the oracle without a host copy: the device's h2w_advice_digest equals the checksum the oracle's streaming context
    (a ring of the last cells + the same position-dependent sum) accumulates over its stream of the same proof; gates and lookups hold on the device."""
    import torch
    ko, kh = published
    q = 28 if cfg == "cfg3" else 84
    sh = h2w.fibonacci_shape(20, q, rate_bits=1, hash_mode=0); osh = oracle.fibonacci_shape(20, q, rate_bits=1, hash_mode=0)
    plan = h2w_api.Plan(sh, kh)
    st = torch.cuda.current_stream().cuda_stream
    proof = oracle.synth_proof(osh, 0xF1B00003 if cfg == "cfg3" else 0xF1B00005)
    d_proof = torch.frombuffer(bytearray(bytes(proof)), dtype=torch.int64).cuda()
    advice = torch.empty(plan.num_cells * 32, dtype=torch.uint8, device="cuda")
    ws = torch.zeros(plan.workspace_bytes(1), dtype=torch.uint8, device="cuda")
    digest = torch.zeros(4, dtype=torch.int64, device="cuda")
    plan.run(d_proof.data_ptr(), 1, advice.data_ptr(), ws.data_ptr(), st)
    plan.advice_digest(advice.data_ptr(), plan.num_cells, digest.data_ptr(), st)
    torch.cuda.synchronize()
    assert plan.status(ws.data_ptr(), 1) == [0]
    bad_gates, bad_lookups = plan.check_constraints(advice.data_ptr(), 1)
    assert bad_gates == 0 and bad_lookups <= 4                     # (a random proof fails only the limbs of its PoW response)
    got = [int(x) & 0xFFFFFFFFFFFFFFFF for x in digest.cpu().tolist()]
    del advice; torch.cuda.empty_cache()
    ctx = oracle.Ctx(21, streaming=True)
    assert oracle.verify_stark(ctx, osh, ko, proof) == 0
    assert ctx.num_cells() == plan.num_cells
    assert got == ctx.digest()
    ctx.close(); plan.close()


def test_config5_valid_proof(h2w, h2w_api, oracle, published):
    """BASELINE.json configs[4]: 2^20 rows, 84 queries, PoseidonBN254 caps of height 4: a GPU-generated valid instance, its
    59.7 M + load cells checked gate by gate on the device and compared with the oracle's stream."""
    import torch
    ko, kh = published
    sh = h2w.fibonacci_shape(20, 84, rate_bits=1, hash_mode=1); osh = oracle.fibonacci_shape(20, 84, rate_bits=1, hash_mode=1)
    _, _, d_proof, _ = gpu_prove(h2w, h2w_api, oracle, kh, sh, osh, 0xF1B00005)
    plan = h2w_api.Plan(sh, kh)
    advice = _witness_of(h2w_api, plan, d_proof, 1)
    assert plan.check_constraints(advice.data_ptr(), 1) == (0, 0)
    ctx = oracle.Ctx(21); ctx.reserve(plan.num_cells)
    assert oracle.verify_stark(ctx, osh, ko, (C.c_uint64 * plan.proof_words).from_buffer_copy(d_proof.cpu().numpy().tobytes())) == 0
    assert ctx.num_cells() == plan.num_cells == 59708779 + 328424 + 1                # verify_proof cells + witness-load cells + the flow's load_zero (stark/mod.rs:483-508)
    assert advice.cpu().numpy().tobytes() == ctx.advice_bytes()
    ctx.close(); plan.close()
