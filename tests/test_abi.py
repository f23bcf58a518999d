"""The C-ABI library loads and exports every symbol include/h2w.h declares; layout queries work without a GPU;
compute calls fail loudly (no CPU fallback) when no HIP device is visible."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported(h2w):
    hdr = open(os.path.join(ROOT, "include", "h2w.h")).read()
    declared = set(re.findall(r"\b(h2w_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"h2w_fr_t", "h2w_assigned_t", "h2w_shape_t", "h2w_poseidon_consts_t"}
    lib = h2w.lib()
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(h2w.SYMBOLS), declared ^ set(h2w.SYMBOLS)
    assert lib.h2w_abi_version() == 1


def test_struct_sizes_match_oracle_side(h2w, oracle):
    assert C.sizeof(h2w.Shape) == C.sizeof(oracle.Shape) == 64
    assert C.sizeof(h2w.PoseidonConsts) == C.sizeof(oracle.Consts)
    assert C.sizeof(h2w.Assigned) == 48


@pytest.mark.parametrize("d,q,rb,mode,lb", [(5, 84, 1, 0, 21), (5, 84, 1, 1, 21), (10, 4, 1, 1, 21), (10, 4, 1, 0, 21),
                                              (16, 28, 2, 1, 21), (20, 28, 1, 1, 21), (10, 4, 1, 1, 13), (10, 3, 1, 0, 8)])
def test_plan_layout_matches_oracle_cell_count(h2w_api, h2w, oracle, consts, d, q, rb, mode, lb):
    """Shape compiler (host) vs oracle: same number of advice cells and proof words, incl. other lookup_bits."""
    ko, kh = consts
    plan = h2w_api.Plan(h2w.fibonacci_shape(d, q, rate_bits=rb, hash_mode=mode, lookup_bits=lb), kh)
    osh = oracle.fibonacci_shape(d, q, rate_bits=rb, hash_mode=mode, lookup_bits=lb)
    ctx = oracle.Ctx(lb)
    assert oracle.verify_stark(ctx, osh, ko, oracle.synth_proof(osh, 5)) == 0
    assert plan.num_cells == ctx.num_cells()
    assert plan.proof_words == oracle.lib().orc_proof_words(C.byref(osh))
    plan.close(); ctx.close()


def test_eager_cell_counts_without_gpu(h2w_api):
    """Host bookkeeping of the eager API: cell counts per call need no GPU."""
    ctx = h2w_api.Context(21)
    nat = h2w_api.NativeChip(ctx); gl = h2w_api.GoldilocksChip(nat)
    a = gl.load_constant(5); b = gl.load_constant(7)
    n0 = ctx.num_cells()
    c = gl.mul(a, b); assert c.int_value() == 35 and ctx.num_cells() - n0 == 65
    n0 = ctx.num_cells(); gl.sub(a, b); assert ctx.num_cells() - n0 == 66
    n0 = ctx.num_cells(); gl.load_witness(9); assert ctx.num_cells() - n0 == 28
    n0 = ctx.num_cells(); d = gl.div(a, b); assert ctx.num_cells() - n0 == 93 and d.int_value() * 7 % h2w_api.GL_P == 5
    n0 = ctx.num_cells(); nat.num_to_bits(a, 64); assert ctx.num_cells() - n0 == 446
    n0 = ctx.num_cells(); nat.select_from_idx([a] * 16, b); assert ctx.num_cells() - n0 == 237
    n0 = ctx.num_cells(); nat.range_check(a, 48); assert ctx.num_cells() - n0 == 11
    n0 = ctx.num_cells(); nat.decompose_le(nat.load_witness(1 << 200), 56, 5); assert ctx.num_cells() - n0 == 1 + 68
    with pytest.raises(h2w_api.H2WError):
        gl.div(a, gl.load_constant(0))     # reference asserts b != 0 (base.rs:379)
    ctx.close()


def test_scoped_cell_counters(h2w_api):
    """ContextWrapper push_context / pop_context: the #[count] tree of GoldilocksChip::mul (65 cells)."""
    ctx = h2w_api.Context(21); nat = h2w_api.NativeChip(ctx); gl = h2w_api.GoldilocksChip(nat)
    a, b = gl.load_constant(3), gl.load_constant(5)
    ctx.push_context("mul")
    ctx.push_context("mul_no_reduce"); p = nat.mul(a, b); ctx.pop_context()
    ctx.push_context("reduce"); gl.reduce(p); ctx.pop_context()
    ctx.pop_context()
    ctx.push_context("mul"); gl.mul(a, b); ctx.pop_context()
    cc = ctx.cell_counts()
    assert cc["all"] == ctx.num_cells() == 2 + 65 + 65
    assert cc["all;mul"] == 130 and cc["all;mul;mul_no_reduce"] == 4 and cc["all;mul;reduce"] == 61
    with pytest.raises(h2w_api.H2WError):
        ctx.pop_context()
    ctx.close()


def test_no_cpu_fallback(h2w_api, h2w):
    """Without a HIP device the product refuses to produce cells (it must never route through a CPU path)."""
    if h2w.lib().h2w_device_count() > 0:
        pytest.skip("GPU present")
    ctx = h2w_api.Context(21)
    h2w_api.GoldilocksChip(h2w_api.NativeChip(ctx)).load_witness(3)
    with pytest.raises(h2w_api.H2WError, match="no HIP device"):
        ctx.advice_bytes()
    ctx.close()


def test_prover_has_no_cpu_path_and_rejects_bad_shapes(h2w_api, h2w):
    """h2w_prover_new: without a HIP device it fails loudly (proof generation exists on the GPU only); unsupported shapes are
    refused with an error string instead of a crash."""
    kh = h2w.published_consts()
    if h2w.lib().h2w_device_count() == 0:
        with pytest.raises(h2w_api.H2WError, match="no HIP device"):
            h2w_api.Prover(h2w.fibonacci_shape(6, 2), kh)
        return
    bad = h2w.fibonacci_shape(6, 2); bad.arity_bits = 7
    with pytest.raises(h2w_api.H2WError, match="unsupported shape"):
        h2w_api.Prover(bad, kh)
    bad = h2w.fibonacci_shape(6, 2); bad.n_cols = 40
    with pytest.raises(h2w_api.H2WError, match="unsupported shape"):
        h2w_api.Prover(bad, kh)


def test_published_tables_are_a_plain_data_call(h2w):
    """h2w_poseidon_published needs no device; a null pointer is an error, not a crash."""
    assert h2w.lib().h2w_poseidon_published(None) != 0
    assert "null" in h2w.last_error()
    k = h2w.published_consts()
    assert k.mds_circ[0] == 17 and k.mds_diag[0] == 8 and k.all_round_constants[0] == 0xb585f766f2144405


def test_comm_entry_points_fail_loudly_without_a_device(h2w):
    """The ingest communicator (h2w_comm_*) validates its arguments and needs a HIP device: no silent single-rank stand-in."""
    import torch
    L = h2w.lib()
    assert L.h2w_comm_unique_id(None) != 0 and "null" in h2w.last_error()
    ident = (C.c_ubyte * 128)()
    assert not L.h2w_comm_init(ident, 2, 2, 0) and "rank" in h2w.last_error()          # rank outside the world
    assert not L.h2w_comm_init(None, 0, 1, 0)
    if torch.cuda.is_available():
        pytest.skip("GPU present: the no-device branch is not reachable here")
    assert not L.h2w_comm_init(ident, 0, 1, 0) and "no HIP device" in h2w.last_error()
    assert L.h2w_comm_rank(None) == -1 and L.h2w_comm_world(None) == 0
    L.h2w_comm_free(None)


def test_context_reset_is_a_fresh_context_with_its_memory(h2w, h2w_api):
    """h2w_ctx_reset: cell count, scopes, the cached load_zero cell, keygen lists and the trace are gone; a handle of the old run is not a cell of the new one."""
    ctx = h2w_api.Context(21, False, 0)
    chip = h2w_api.GoldilocksChip(h2w_api.NativeChip(ctx))
    ctx.push_context("a"); a = chip.load_constant(5); b = chip.mul(a, a); ctx.pop_context()
    n1, g1 = ctx.num_cells(), len(ctx.gate_cells())
    assert n1 > 0 and g1 > 0
    ctx.reset()
    assert ctx.num_cells() == 0 and len(ctx.gate_cells()) == 0 and ctx.cell_counts() == {"all": 0}
    ctx.push_context("a"); a2 = chip.load_constant(5); chip.mul(a2, a2); ctx.pop_context()
    assert ctx.num_cells() == n1 and len(ctx.gate_cells()) == g1
    ctx.reset(); ctx.trace_begin()
    chip.load_constant(7)
    with pytest.raises(h2w.H2WError):        # an operand of the previous run: another context id
        chip.mul(a2, a2)
        h2w_api.Plan.from_trace(ctx, 4)
    ctx.close()


def test_context_reserve_sizes_a_new_context_and_changes_nothing_else(h2w, h2w_api):
    """h2w_ctx_footprint / h2w_ctx_reserve: the footprint of a run sizes a NEW context ahead of the same run; records and cells are what they were, reserving
    again (or less) is a no-op, an absurd size is refused."""
    def run(ctx):
        native = h2w_api.NativeChip(ctx); chip = h2w_api.GoldilocksChip(native)
        a = chip.load_witness(0xFFFFFFFF00000000); b = chip.load_constant(12345)
        for _ in range(300):
            a = chip.mul_add(a, b, a)
        w = native.load_witness(2**200 + 5)        # values beyond 64 bits: literal cells
        for _ in range(20):
            w = native.mul(w, w)
        return ctx.num_cells(), ctx.footprint(), (bytes(a.value), bytes(w.value))
    c1 = h2w_api.Context(21, True, 0)
    n1, fp1, v1 = run(c1)
    assert fp1[0] > 0 and fp1[1] > 0
    c2 = h2w_api.Context(21, True, 0)
    assert c2.footprint() == (0, 0)
    c2.reserve(*fp1); c2.reserve(1, 1); c2.reserve(4 * fp1[0], 4 * fp1[1])
    assert c2.footprint() == (0, 0) and c2.num_cells() == 0
    n2, fp2, v2 = run(c2)
    assert (n2, fp2, v2) == (n1, fp1, v1)
    with pytest.raises(h2w.H2WError):
        c2.reserve(1 << 40, 1)
    c1.close(); c2.close()


def test_product_library_has_no_experiment_switches(h2w):
    """No experiment switches exist in the sources any more (round 1 shipped wrong-output kernel variants behind getenv): kernel variants are
    local patches / compile-time flags built as a separately named library by tools/experiments/variant.sh, and the library the tests, smoke()
    and bench.py load reads no such environment variable."""
    blob = open(h2w.LIB_PATH, "rb").read()
    assert b"H2W_DBG" not in blob and b"H2W_EXPAND_VARIANT" not in blob and b"H2W_BN_UNITS" not in blob
    assert os.path.basename(h2w.LIB_PATH) == "libh2w.so" or os.environ.get("H2W_LIB")
