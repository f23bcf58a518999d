"""Record and replay (include/h2w.h 2d; VERDICT r03 task 4): ONE run of the verifier gadget driven through nothing but the level-1 / level-2 C ABI
(h2w_chip_verify_stark: csrc/abi_backend.cpp stands in for the reference's unchanged chips over the NativeChip shim, field/native.rs:28-193) is
recorded on proof A; h2w_plan_from_trace lowers the tape; h2w_fri_witness_batch on that plan then generates the witnesses of OTHER proofs of the shape
on the GPU - byte for byte the oracle's streams (and the traced proof's own)."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


def trace_and_replay(h2w, h2w_api, oracle, consts, shape_args, seed_a, seeds, cap_height=4, valid=False, lookup_bits=21):
    import numpy as np
    import torch
    ko, kh = consts
    sh = h2w.fibonacci_shape(*shape_args[:2], rate_bits=shape_args[2], hash_mode=shape_args[3], cap_height=cap_height, lookup_bits=lookup_bits)
    osh = oracle.fibonacci_shape(*shape_args[:2], rate_bits=shape_args[2], hash_mode=shape_args[3], cap_height=cap_height, lookup_bits=lookup_bits)
    mk = (lambda s: oracle.prove_fri(osh, ko, s)) if valid else (lambda s: oracle.synth_proof(osh, s))
    proof_a = mk(seed_a)
    ctx = h2w_api.Context(lookup_bits, True, 0)
    ctx.trace_begin()
    h2w_api.verify_stark(ctx, sh, kh, np.frombuffer(bytes(proof_a), dtype=np.uint64))
    plan = h2w_api.Plan.from_trace(ctx, len(proof_a))
    assert plan.num_cells == ctx.num_cells() and plan.proof_words == len(proof_a)
    traced_stream = ctx.advice_bytes()                       # the traced run's own stream (eager expansion on the GPU)
    ctx.close()
    proofs = [proof_a] + [mk(s) for s in seeds]
    n = len(proofs)
    host = torch.empty(n * plan.proof_words, dtype=torch.int64)
    for i, p in enumerate(proofs):
        host[i * plan.proof_words:(i + 1) * plan.proof_words] = torch.frombuffer(bytearray(bytes(p)), dtype=torch.int64)
    d_proofs = host.cuda()
    advice = torch.zeros(n * plan.num_cells * 32, dtype=torch.uint8, device="cuda")
    ws = torch.zeros(plan.workspace_bytes(n), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    plan.run(d_proofs.data_ptr(), n, advice.data_ptr(), ws.data_ptr(), st)
    torch.cuda.synchronize()
    assert plan.status(ws.data_ptr(), n, st) == [0] * n
    got = advice.cpu().numpy().tobytes()
    nb = plan.num_cells * 32
    assert got[:nb] == traced_stream, "the replay of the traced proof differs from the traced run itself"
    for i, p in enumerate(proofs):
        o = oracle.Ctx(lookup_bits, track_scopes=False)
        assert oracle.verify_stark(o, osh, ko, p) == 0
        want = o.advice_bytes(); o.close()
        g = got[i * nb:(i + 1) * nb]
        if g != want:
            a = np.frombuffer(g, dtype=np.uint64).reshape(-1, 4); b = np.frombuffer(want, dtype=np.uint64).reshape(-1, 4)
            bad = np.nonzero((a != b).any(axis=1))[0]
            raise AssertionError(f"proof {i}: {len(bad)} cells differ, first at {bad[:8]}: got {a[bad[0]]} want {b[bad[0]]}")
    plan.close()


@pytest.mark.parametrize("mode", [1, 0])
def test_replay_small_shapes(h2w, h2w_api, oracle, consts, mode):
    trace_and_replay(h2w, h2w_api, oracle, consts, (6, 2, 1, mode), 1, [2, 3])              # no fold step
    trace_and_replay(h2w, h2w_api, oracle, consts, (7, 3, 2, mode), 4, [5])                 # one fold step (rate_bits 2)
    trace_and_replay(h2w, h2w_api, oracle, consts, (9, 2, 1, mode), 33, [34], cap_height=2, valid=True)      # valid FRI instances


@pytest.mark.parametrize("mode", [1, 0])
def test_replay_config1(h2w, h2w_api, oracle, published, mode):
    """BASELINE.json configs[0] (2^10 rows, 4 queries), both hash modes, published tables: traced on one proof, replayed on three others."""
    trace_and_replay(h2w, h2w_api, oracle, published, (10, 4, 1, mode), 0xF1B00001, [0xF1B00002, 0xF1B00003, 0xF1B00004])


def test_replay_config3_bn254(h2w, h2w_api, oracle, published):
    """BASELINE.json configs[2] (2^20 rows, 28 queries, PoseidonBN254 caps): 28.6 M cells per proof, every byte of two replayed proofs."""
    trace_and_replay(h2w, h2w_api, oracle, published, (20, 28, 1, 1), 0xF1B00003, [0xF1B00013])


def test_replay_other_lookup_bits(h2w, h2w_api, oracle, consts):
    trace_and_replay(h2w, h2w_api, oracle, consts, (7, 2, 1, 1), 7, [8], lookup_bits=13)


def test_replay_reports_the_reference_panics_as_status_words(h2w, h2w_api, oracle, consts):
    """A replayed proof whose query lands on a zero denominator: the reference asserts (base.rs:379); the replay sets the proof's status word and goes on."""
    import numpy as np
    import torch
    ko, kh = consts
    sh = h2w.fibonacci_shape(6, 2, hash_mode=1); osh = oracle.fibonacci_shape(6, 2, hash_mode=1)
    proof_a = oracle.synth_proof(osh, 1)
    ctx = h2w_api.Context(21, True, 0); ctx.trace_begin()
    h2w_api.verify_stark(ctx, sh, kh, np.frombuffer(bytes(proof_a), dtype=np.uint64))
    plan = h2w_api.Plan.from_trace(ctx, len(proof_a)); ctx.close()
    # compare the status words with the batched (compiled) plan's on the same proofs
    ref = h2w_api.Plan(sh, kh)
    n = 3
    host = torch.from_numpy(np.random.default_rng(5).integers(0, 1 << 60, n * plan.proof_words, dtype=np.int64))
    d = host.cuda(); st = torch.cuda.current_stream().cuda_stream
    out = []
    for pl in (plan, ref):
        adv = torch.zeros(n * pl.num_cells * 32, dtype=torch.uint8, device="cuda"); ws = torch.zeros(pl.workspace_bytes(n), dtype=torch.uint8, device="cuda")
        pl.run(d.data_ptr(), n, adv.data_ptr(), ws.data_ptr(), st); torch.cuda.synchronize()
        out.append((pl.status(ws.data_ptr(), n, st), adv.cpu().numpy().tobytes()))
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1]
    plan.close(); ref.close()


@pytest.mark.parametrize("mode", [1, 0])
def test_replay_writes_the_flexgate_columns(h2w, h2w_api, oracle, consts, mode):
    """h2w_fri_witness_batch_columns on a traced plan: the recorded run replayed straight into the FlexGate column layout (direct cells through the column cursor,
    block records through the column form of the expansion kernel, boundary rows repeated by the fix-up kernel) - the columns the compiled plan of the same shape
    writes for the same proofs (tests/test_layout_metadata.py pins those to the oracle's relayout), at two column heights."""
    import numpy as np
    import torch
    ko, kh = consts
    args = (7, 3, 2, mode)
    sh = h2w.fibonacci_shape(*args[:2], rate_bits=args[2], hash_mode=args[3]); osh = oracle.fibonacci_shape(*args[:2], rate_bits=args[2], hash_mode=args[3])
    proofs = [oracle.synth_proof(osh, s) for s in (31, 32, 33)]
    ctx = h2w_api.Context(21, True, 0)
    ctx.trace_begin()
    h2w_api.verify_stark(ctx, sh, kh, np.frombuffer(bytes(proofs[0]), dtype=np.uint64))
    traced = h2w_api.Plan.from_trace(ctx, len(proofs[0])); ctx.close()
    compiled = h2w_api.Plan(sh, kh)
    assert traced.num_cells == compiled.num_cells
    n = len(proofs)
    host = torch.empty(n * compiled.proof_words, dtype=torch.int64)
    for i, p in enumerate(proofs):
        host[i * compiled.proof_words:(i + 1) * compiled.proof_words] = torch.frombuffer(bytearray(bytes(p)), dtype=torch.int64)
    d_proofs = host.cuda(); st = torch.cuda.current_stream().cuda_stream
    for k in (14, 17):
        bp = compiled.break_points(k); ncol = len(bp) + 1
        want = torch.full((((n * ncol) << k) * 32,), 0x5A, dtype=torch.uint8, device="cuda")
        ws = torch.zeros(compiled.workspace_bytes(n), dtype=torch.uint8, device="cuda")
        compiled.run_columns(d_proofs.data_ptr(), n, bp, k, want.data_ptr(), ws.data_ptr(), st)
        got = torch.full((((n * ncol) << k) * 32,), 0xA5, dtype=torch.uint8, device="cuda")          # poisoned: every row is written or zero-filled
        ws2 = torch.zeros(traced.workspace_bytes(n), dtype=torch.uint8, device="cuda")
        traced.run_columns(d_proofs.data_ptr(), n, bp, k, got.data_ptr(), ws2.data_ptr(), st)
        torch.cuda.synchronize()
        assert compiled.status(ws.data_ptr(), n, st) == [0] * n and traced.status(ws2.data_ptr(), n, st) == [0] * n
        if not torch.equal(want, got):
            a = want.cpu().numpy().view(np.uint64).reshape(-1, 4); b = got.cpu().numpy().view(np.uint64).reshape(-1, 4)
            bad = np.nonzero((a != b).any(axis=1))[0]; rows = 1 << k
            raise AssertionError(f"k={k}: {len(bad)} cells differ; first (proof, col, row) = {[(int(i) // (ncol * rows), (int(i) // rows) % ncol, int(i) % rows) for i in bad[:6]]} got {b[bad[0]]} want {a[bad[0]]}")
    traced.close(); compiled.close()
