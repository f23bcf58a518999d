"""keygen-side metadata of the cell stream and the FlexGate column layout.

Host part (no GPU): the product's selector bitmap, lookup registrations (in order) and break points — produced by the shape
compiler from template slot flags and the backend's gate/lookup markers — against the oracle, which records them the
halo2-base way (Context::assign_region gate offsets, RangeChip::range_check -> cells_to_lookup).  The semantics of the
third-party halo2-lib are restated from recollection ([R], oracle.c) — "parity unpinned" like the rest of that boundary.
GPU part: the column / lookup-column kernels against the oracle's restated assign_witnesses."""
import numpy as np
import pytest

SHAPES = [dict(d=6, q=2), dict(d=7, q=3, rb=2), dict(d=6, q=2, n_perm_z=0), dict(d=6, q=1, cap=0),
          dict(d=8, q=2, rb=2, cap=2, arity_bits=2, final_poly_bits=3), dict(d=6, q=2, pow_bits=10, n_cols=6, n_quotient=4, n_pis=1, num_challenges=3)]


def _shapes(h2w, oracle, mode, lookup_bits, kw):
    kw = dict(kw)
    args = dict(rate_bits=kw.pop("rb", 1), cap_height=kw.pop("cap", 4), hash_mode=mode, lookup_bits=lookup_bits)
    d, q = kw.pop("d"), kw.pop("q")
    sh, osh = h2w.fibonacci_shape(d, q, **args), oracle.fibonacci_shape(d, q, **args)
    for k, v in kw.items():
        setattr(sh, k, v); setattr(osh, k, v)
    return sh, osh


def _oracle_ctx(oracle, osh, ko, seed=3):
    ctx = oracle.Ctx(osh.lookup_bits, witness_gen_only=False)
    assert oracle.verify_stark(ctx, osh, ko, oracle.synth_proof(osh, seed)) == 0
    return ctx


@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("lookup_bits", [21, 13, 8])
def test_selectors_lookups_break_points_match_oracle(h2w, h2w_api, oracle, consts, mode, lookup_bits):
    ko, kh = consts
    for kw in (SHAPES if lookup_bits == 21 else SHAPES[:2]):
        sh, osh = _shapes(h2w, oracle, mode, lookup_bits, kw)
        plan = h2w_api.Plan(sh, kh)
        ctx = _oracle_ctx(oracle, osh, ko)
        assert plan.num_cells == ctx.num_cells()
        assert plan.selectors() == ctx.selectors(), kw
        assert plan.lookup_cells() == ctx.lookup_cells(), kw
        mp = ctx.mock_prover()
        assert mp["gates"] == int(plan.L.h2w_plan_num_gates(plan.p)) and mp["lookups"] == len(plan.lookup_cells())
        for k in (12, 15, 18):
            bp = plan.break_points(k)
            assert bp == ctx.break_points(k), (kw, k)
            # invariants of the packing: every column within max_rows, no gate straddles a break, the stream is covered once
            max_rows = (1 << k) - 9
            assert all(b + 1 <= max_rows for b in bp)
            used = sum(b + 1 for b in bp) - len(bp)
            assert 0 < plan.num_cells - used <= max_rows
            sel = np.unpackbits(np.frombuffer(plan.selectors(), dtype=np.uint8), bitorder="little")
            pos = 0
            for b in bp:                      # cells pos..pos+b live in this column; a gate starting at cell i needs rows i..i+3
                seg = sel[pos:pos + b + 1]
                gates = np.nonzero(seg)[0]
                assert gates.size == 0 or gates[-1] + 3 <= b or gates[-1] == b, (kw, k)   # the breaking cell's gate moves to the next column
                pos += b
        ctx.close(); plan.close()


def test_metadata_api_errors(h2w, h2w_api, consts):
    ko, kh = consts
    import ctypes as C
    L = h2w.lib()
    n = C.c_uint64()
    assert L.h2w_break_points(None, 10, 12, 9, None, 0, C.byref(n)) != 0
    assert L.h2w_break_points((C.c_uint8 * 2)(), 10, 2, 9, None, 0, C.byref(n)) != 0          # k too small
    assert L.h2w_plan_selectors(None, None) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 0])
def test_column_layout_on_device(h2w, h2w_api, oracle, consts, mode):
    import torch
    ko, kh = consts
    sh, osh = _shapes(h2w, oracle, mode, 21, dict(d=6, q=2))
    plan = h2w_api.Plan(sh, kh)
    seeds = [5, 6]
    proofs = [oracle.synth_proof(osh, s) for s in seeds]
    host = torch.empty(len(seeds) * plan.proof_words, dtype=torch.int64)
    for i, p in enumerate(proofs):
        host[i * plan.proof_words:(i + 1) * plan.proof_words] = torch.frombuffer(bytearray(bytes(p)), dtype=torch.int64)
    d_proofs = host.cuda()
    n = len(seeds)
    advice = torch.zeros(n * plan.num_cells * 32, dtype=torch.uint8, device="cuda")
    ws = torch.zeros(plan.workspace_bytes(n), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    plan.run(d_proofs.data_ptr(), n, advice.data_ptr(), ws.data_ptr(), stream)
    for k in (14, 17):
        bp = plan.break_points(k)
        ncol = len(bp) + 1
        cols = torch.full((((n * ncol) << k) * 32,), 0xAB, dtype=torch.uint8, device="cuda")          # poisoned: unassigned rows must come back zero
        plan.layout_columns(advice.data_ptr(), n, bp, k, cols.data_ptr(), stream)
        nl = plan.num_lookup_columns(k)
        lk = torch.full((((n * nl) << k) * 32,), 0xCD, dtype=torch.uint8, device="cuda")
        assert plan.layout_lookup_columns(advice.data_ptr(), n, k, lk.data_ptr(), stream=stream) == nl
        cols2 = torch.full((((n * ncol) << k) * 32,), 0x5A, dtype=torch.uint8, device="cuda")          # fused: the kernels write the columns directly
        ws2 = torch.zeros(plan.workspace_bytes(n), dtype=torch.uint8, device="cuda")
        plan.run_columns(d_proofs.data_ptr(), n, bp, k, cols2.data_ptr(), ws2.data_ptr(), stream)
        torch.cuda.synchronize()
        assert plan.status(ws2.data_ptr(), n, stream) == [0] * n
        if not torch.equal(cols, cols2):
            a = cols.cpu().numpy().view(np.uint64).reshape(-1, 4); b = cols2.cpu().numpy().view(np.uint64).reshape(-1, 4)
            bad = np.nonzero((a != b).any(axis=1))[0]
            rows = 1 << k
            raise AssertionError(f"k={k}: fused column-major emission differs from flat + relayout in {len(bad)} cells; first (proof, col, row) = "
                                 f"{[(int(i) // (ncol * rows), (int(i) // rows) % ncol, int(i) % rows) for i in bad[:8]]} bp={bp[:4]} got {b[bad[0]]} want {a[bad[0]]}")
        got, gotl = cols.cpu().numpy().tobytes(), lk.cpu().numpy().tobytes()
        for i, p in enumerate(proofs):
            ctx = oracle.Ctx(21, witness_gen_only=False)
            assert oracle.verify_stark(ctx, osh, ko, p) == 0
            assert ctx.break_points(k) == bp
            want = ctx.layout_columns(bp, k)
            sz = (ncol << k) * 32
            assert got[i * sz:(i + 1) * sz] == want, (k, i)
            onl, wantl = ctx.layout_lookup_columns(k)
            assert onl == nl
            szl = (nl << k) * 32
            assert gotl[i * szl:(i + 1) * szl] == wantl, (k, i)
            ctx.close()
    plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 0])
def test_device_constraint_check(h2w, h2w_api, oracle, consts, mode):
    """h2w_check_constraints: all gates and lookups of the GPU-generated stream hold (same counts as the oracle's MockProver
    checks), and it notices a corrupted cell."""
    import torch
    ko, kh = consts
    sh, osh = _shapes(h2w, oracle, mode, 21, dict(d=7, q=3, rb=2))
    plan = h2w_api.Plan(sh, kh)
    n = 2
    proofs = [oracle.synth_proof(osh, s) for s in (8, 9)]
    host = torch.empty(n * plan.proof_words, dtype=torch.int64)
    for i, p in enumerate(proofs):
        host[i * plan.proof_words:(i + 1) * plan.proof_words] = torch.frombuffer(bytearray(bytes(p)), dtype=torch.int64)
    d_proofs = host.cuda()
    advice = torch.zeros(n * plan.num_cells * 32, dtype=torch.uint8, device="cuda")
    ws = torch.zeros(plan.workspace_bytes(n), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    plan.run(d_proofs.data_ptr(), n, advice.data_ptr(), ws.data_ptr(), stream)
    # random proofs fail the PoW range check (a SEMANTIC lookup: the response has no leading zeros), nothing else
    ctx = _oracle_ctx(oracle, osh, ko, 8)
    mp = ctx.mock_prover(); ctx.close()
    bad_g, bad_l = plan.check_constraints(advice.data_ptr(), n, stream)
    assert bad_g == 0
    assert bad_l <= 2 * 4                      # at most the PoW limbs of each proof
    sel = np.unpackbits(np.frombuffer(plan.selectors(), dtype=np.uint8), bitorder="little")
    gate_cells = np.nonzero(sel)[0]
    a = advice.view(torch.int64)
    for g in (gate_cells[5], gate_cells[len(gate_cells) // 2], gate_cells[-1]):
        idx = (plan.num_cells + int(g) + 3) * 4          # proof 1, the gate's output cell, low limb
        old = a[idx].item(); a[idx] = old ^ 1
        assert plan.check_constraints(advice.data_ptr(), n, stream)[0] >= 1
        a[idx] = old
    assert plan.check_constraints(advice.data_ptr(), n, stream)[0] == 0
    lk = plan.lookup_cells()
    idx = int(lk[len(lk) // 3]) * 4 + 1                  # proof 0: a looked-up limb gets a high word
    a[idx] = 1
    assert plan.check_constraints(advice.data_ptr(), n, stream)[1] >= bad_l + 1
    plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(16, 28, 2, 0, 342848685 + 90444), (20, 28, 1, 0, 450331361 + 117724), (20, 84, 1, 1, 59708779 + 328424 + 1)],
                         ids=["cfg2-gl", "cfg3-gl", "cfg5-bn254"])
def test_full_size_streams_satisfy_all_gates_and_lookups(h2w, h2w_api, oracle, consts, cfg):
    """BASELINE.json configs at full size where a cell-by-cell comparison with the oracle is too large for a unit test (cfg 2 / 3
    with Goldilocks-Poseidon Merkle caps: 11 / 14.4 GB of advice per proof; cfg 5: 84 queries) -> the size-independent property:
    every gate and every lookup of the stream holds on the device (the PoW limbs of a random proof aside), and the stream has
    exactly the cell count of SURVEY §8d (verify_proof subtree + witness-load cells; BN254 mode + the one cached load_zero cell, App. A)."""
    import torch
    ko, kh = consts
    d, q, rb, mode, cells = cfg
    sh = h2w.fibonacci_shape(d, q, rate_bits=rb, hash_mode=mode)
    osh = oracle.fibonacci_shape(d, q, rate_bits=rb, hash_mode=mode)
    plan = h2w_api.Plan(sh, kh)
    assert plan.num_cells == cells
    p = oracle.synth_proof(osh, 0xF1B00002)
    d_proofs = torch.frombuffer(bytearray(bytes(p)), dtype=torch.int64).cuda()
    advice = torch.empty(plan.num_cells * 32, dtype=torch.uint8, device="cuda")
    ws = torch.zeros(plan.workspace_bytes(1), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    plan.run(d_proofs.data_ptr(), 1, advice.data_ptr(), ws.data_ptr(), stream)
    torch.cuda.synchronize()
    assert plan.status(ws.data_ptr(), 1, stream) == [0]
    bad_g, bad_l = plan.check_constraints(advice.data_ptr(), 1, stream)
    assert bad_g == 0 and bad_l <= 4, (bad_g, bad_l)
    assert int(plan.L.h2w_plan_num_gates(plan.p)) > plan.num_cells // 5
    plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 0])
def test_full_mockprover_on_the_device(h2w, h2w_api, oracle, consts, mode):
    """The restated MockProver entirely on the GPU-generated stream: gates and lookups (plan metadata) plus the copy constraints and
    constant equalities the plan exports (equal to what an eager keygen context of the same shape records).  On a VALID FRI instance every constraint holds,
    chip-level assert_equal included; on a corrupted proof only copy constraints break."""
    import ctypes as C
    import torch
    ko, kh = consts
    sh, osh = _shapes(h2w, oracle, mode, 21, dict(d=6, q=2, cap=2))
    good = oracle.prove_fri(osh, ko, 2024)
    # keygen: the plan's own lists (h2w_plan_equalities / _const_equalities: a host replay of the shape).  Gates, copy constraints and
    # lookups are static per shape; so are the constant equalities except for the reference's quirk that Goldilocks-Poseidon hash wires
    # are loaded as CONSTANTS (hash/poseidon/hash.rs:86-96), which ties those to the proof: they are filled in from `good`.
    plan = h2w_api.Plan(sh, kh)
    eqs, ceqs = plan.equalities(), plan.const_equalities(good)
    kctx = h2w_api.Context(21, witness_gen_only=False)          # ... and they are what an eager keygen context of the same proof records
    assert h2w.lib().h2w_chip_verify_stark(kctx.p, C.byref(sh), C.byref(kh), good) == 0, h2w.last_error()
    assert sorted(eqs) == sorted(kctx.equalities()) and sorted(ceqs) == sorted(kctx.const_equalities())
    assert kctx.num_cells() == plan.num_cells and kctx.gate_cells() == [i for i in np.nonzero(np.unpackbits(np.frombuffer(plan.selectors(), dtype=np.uint8), bitorder="little"))[0]]
    bad_proof = (C.c_uint64 * len(good))(*good); bad_proof[len(good) // 2] ^= 1
    st = torch.cuda.current_stream().cuda_stream
    for proof, valid in ((good, True), (bad_proof, False)):
        d_proofs = torch.frombuffer(bytearray(bytes(proof)), dtype=torch.int64).cuda()
        advice = torch.zeros(plan.num_cells * 32, dtype=torch.uint8, device="cuda")
        ws = torch.zeros(plan.workspace_bytes(1), dtype=torch.uint8, device="cuda")
        plan.run(d_proofs.data_ptr(), 1, advice.data_ptr(), ws.data_ptr(), st)
        torch.cuda.synchronize()
        bad_g, bad_l = plan.check_constraints(advice.data_ptr(), 1, st)
        bad_e, bad_c = plan.check_equalities(advice.data_ptr(), 1, eqs, ceqs, st)
        if valid:
            assert (bad_g, bad_l, bad_e, bad_c) == (0, 0, 0, 0)
        else:
            assert bad_g == 0 and bad_e + bad_c > 0 and (mode == 0 or bad_c == 0)
    kctx.close(); plan.close()
