"""Record and replay, the host side (no GPU): an eager context in trace mode records the verifier gadget's run through the level-1 / level-2 C ABI;
h2w_plan_from_trace lowers it (static widths, segments, templates) and reports what cannot be replayed."""
import ctypes as C

import numpy as np
import pytest


def _trace(h2w, h2w_api, oracle, shape_args, seed=42):
    sh = h2w.fibonacci_shape(*shape_args[:2], rate_bits=shape_args[2], hash_mode=shape_args[3]); osh = oracle.fibonacci_shape(*shape_args[:2], rate_bits=shape_args[2], hash_mode=shape_args[3])
    proof = oracle.synth_proof(osh, seed)
    ctx = h2w_api.Context(21, True, 0); ctx.trace_begin()
    h2w_api.verify_stark(ctx, sh, h2w.published_consts(), np.frombuffer(bytes(proof), dtype=np.uint64))
    return ctx, proof, sh


@pytest.mark.parametrize("mode", [1, 0])
def test_a_traced_run_lowers_to_a_plan_of_the_same_stream(h2w, h2w_api, oracle, mode):
    ctx, proof, sh = _trace(h2w, h2w_api, oracle, (7, 3, 2, mode))
    plan = h2w_api.Plan.from_trace(ctx, len(proof))
    ref = h2w_api.Plan(sh, h2w.published_consts())
    assert plan.num_cells == ctx.num_cells() == ref.num_cells and plan.proof_words == ref.proof_words
    assert plan.num_records > 0 and plan.workspace_bytes(3) > 0
    plan.close(); ref.close(); ctx.close()


def test_what_cannot_be_replayed_is_refused(h2w, h2w_api, oracle):
    L = h2w.lib()
    # an untagged witness: its value exists only on the host
    ctx = h2w_api.Context(21, True, 0); ctx.trace_begin()
    chip = h2w_api.GoldilocksChip(h2w_api.NativeChip(ctx))
    chip.load_witness(5)
    with pytest.raises(h2w.H2WError, match="h2w_trace_input"):
        h2w_api.Plan.from_trace(ctx, 16)
    ctx.close()
    # scopes claimed parallel whose instances feed one another
    ctx = h2w_api.Context(21, True, 0); ctx.trace_begin()
    native = h2w_api.NativeChip(ctx); chip = h2w_api.GoldilocksChip(native)
    L.h2w_trace_input(ctx.p, 0, 1); a = chip.load_witness(3)
    ctx.push_context("step"); b = chip.mul(a, a); ctx.pop_context()
    ctx.push_context("step"); chip.mul(b, a); ctx.pop_context()          # reads the first instance's result
    with pytest.raises(h2w.H2WError, match="not independent"):
        h2w_api.Plan.from_trace(ctx, 16, parallel_scopes=("step",))
    plan = h2w_api.Plan.from_trace(ctx, 16, parallel_scopes=())            # sequentially it is fine
    assert plan.num_cells == ctx.num_cells()
    plan.close(); ctx.close()
    # a context that is not tracing
    ctx = h2w_api.Context(21, True, 0)
    with pytest.raises(h2w.H2WError, match="trace mode"):
        h2w_api.Plan.from_trace(ctx, 16)
    ctx.close()
