"""Host-side mirrors of the reference interface over the C-ABI (include/h2w.h).

`Context` + `NativeChip` + `GoldilocksChip` mirror halo2-base Context / verifier NativeChip
(verifier/src/field/native.rs) / GoldilocksChip (verifier/src/field/goldilocks/base.rs): same method names,
argument meaning and error behaviour (a non-zero status raises H2WError where the reference panics).
`Plan` drives the batched hot path.  torch is used only for device memory / streams.
"""
import ctypes as C

from . import (Assigned, Fr, H2WError, PoseidonConsts, Shape, _ck, last_error, lib)

GL_P = 0xFFFFFFFF00000001
FR_MODULUS = 21888242871839275222246405745257275088548364400416034343698204186575808495617


class Context:
    """halo2-base Context as seen through ContextWrapper (util/context_wrapper.rs): an advice stream."""

    def __init__(self, lookup_bits=21, witness_gen_only=True, device_id=0):
        self.L = lib()
        self.p = self.L.h2w_ctx_new(lookup_bits, 1 if witness_gen_only else 0, device_id)
        if not self.p:
            raise H2WError("h2w_ctx_new: " + last_error())
        self.lookup_bits = lookup_bits

    def reset(self):
        """The context as new, its host memory kept (h2w_ctx_reset): for the next proof's run."""
        _ck(self.L.h2w_ctx_reset(self.p), "h2w_ctx_reset")

    def footprint(self):
        """(block records, literal cells) the run so far appended (h2w_ctx_footprint): the sizes for `reserve` on a new context."""
        out = (C.c_uint64 * 2)()
        _ck(self.L.h2w_ctx_footprint(self.p, out), "h2w_ctx_footprint")
        return int(out[0]), int(out[1])

    def reserve(self, n_records, n_literal_cells):
        """Host vectors of a NEW context sized and mapped ahead of its first run (h2w_ctx_reserve)."""
        _ck(self.L.h2w_ctx_reserve(self.p, n_records, n_literal_cells), "h2w_ctx_reserve")

    def trace_begin(self):
        """Record the op tape of this context's run (h2w_ctx_trace_begin; Plan.from_trace turns it into a replayable plan)."""
        _ck(self.L.h2w_ctx_trace_begin(self.p), "h2w_ctx_trace_begin")

    def close(self):
        if getattr(self, "p", None):
            self.L.h2w_ctx_free(self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def num_cells(self):  # ContextWrapper::num_cells
        return int(self.L.h2w_num_cells(self.p))

    def push_context(self, name):  # ContextWrapper::push_context (what #[count] emits on entry)
        _ck(self.L.h2w_push_context(self.p, name.encode()), "h2w_push_context")

    def pop_context(self):
        _ck(self.L.h2w_pop_context(self.p), "h2w_pop_context")

    def cell_counts(self):
        """{"all;verify_proof;...": inclusive cells} — the data behind the reference's profile/*.svg flamegraphs."""
        n = self.L.h2w_context_dump(self.p, None, 0)
        buf = C.create_string_buffer(n + 1)
        self.L.h2w_context_dump(self.p, buf, n + 1)
        return {line.rsplit(" ", 1)[0]: int(line.rsplit(" ", 1)[1]) for line in buf.value.decode().splitlines()}

    # keygen-side bookkeeping (contexts created with witness_gen_only=False): halo2-base Context's selector / copy manager / lookups
    def _u64s(self, nfn, fn, per=1):
        n = int(nfn(self.p))
        buf = (C.c_uint64 * max(per * n, 1))()
        _ck(fn(self.p, buf), fn.__name__ if hasattr(fn, "__name__") else "h2w_ctx_*")
        return n, buf

    def gate_cells(self):
        n, b = self._u64s(self.L.h2w_ctx_num_gates, self.L.h2w_ctx_gate_cells)
        return [int(b[i]) for i in range(n)]

    def lookup_cells(self):
        n, b = self._u64s(self.L.h2w_ctx_num_lookups, self.L.h2w_ctx_lookup_cells)
        return [int(b[i]) for i in range(n)]

    def equalities(self):
        n, b = self._u64s(self.L.h2w_ctx_num_equalities, self.L.h2w_ctx_equalities, 2)
        return [(int(b[2 * i]), int(b[2 * i + 1])) for i in range(n)]

    def const_equalities(self):
        n = int(self.L.h2w_ctx_num_const_equalities(self.p))
        cells = (C.c_uint64 * max(n, 1))(); vals = (Fr * max(n, 1))()
        _ck(self.L.h2w_ctx_const_equalities(self.p, cells, vals), "h2w_ctx_const_equalities")
        return [(int(cells[i]), int.from_bytes(bytes(vals[i]), "little")) for i in range(n)]

    def advice_device(self):
        """Expands the pending records on the GPU; device pointer to num_cells * 32 bytes owned by the context (h2w_ctx_advice_device)."""
        ptr = C.c_void_p()
        _ck(self.L.h2w_ctx_advice_device(self.p, C.byref(ptr)), "h2w_ctx_advice_device")
        return ptr.value

    def advice_bytes(self, first=0, count=None):
        """Expands the pending records on the GPU and returns canonical-LE cells as bytes."""
        n = self.num_cells() if count is None else count
        buf = (Fr * max(n, 1))()
        _ck(self.L.h2w_ctx_download(self.p, first, n, buf), "h2w_ctx_download")
        return bytes(buf)[: n * 32]


def verify_stark(ctx, shape, consts, proof_words):
    """The reference's test flow over the eager boundary (stark/mod.rs:483-508: load_zero, WitnessChip::load_proof_with_pis, StarkChip::verify_proof):
    the gadget stack driven through nothing but the level-1 / level-2 C ABI (h2w_chip_verify_stark).  proof_words: numpy uint64 / ctypes array."""
    import numpy as np
    if isinstance(proof_words, np.ndarray):
        proof_words = np.ascontiguousarray(proof_words, dtype=np.uint64).ctypes.data_as(C.POINTER(C.c_uint64))
    _ck(ctx.L.h2w_chip_verify_stark(ctx.p, C.byref(shape), C.byref(consts), proof_words), "h2w_chip_verify_stark")


def _arr(items):
    a = (Assigned * len(items))()
    for i, x in enumerate(items):
        a[i] = x
    return a


class NativeChip:
    """verifier/src/field/native.rs:11-194."""

    def __init__(self, ctx):
        self.ctx, self.L, self.p = ctx, ctx.L, ctx.p

    def _out(self):
        return Assigned()

    def load_constant(self, a):
        o = self._out(); _ck(self.L.h2w_load_constant(self.p, C.byref(Fr.from_int(a)), C.byref(o)), "load_constant"); return o

    def load_zero(self):
        o = self._out(); _ck(self.L.h2w_load_zero(self.p, C.byref(o)), "load_zero"); return o

    def load_constants(self, cs):
        arr = (Fr * len(cs))(*[Fr.from_int(c) for c in cs]); out = (Assigned * len(cs))()
        _ck(self.L.h2w_load_constants(self.p, arr, len(cs), out), "load_constants"); return list(out)

    def load_witness(self, a):
        o = self._out(); _ck(self.L.h2w_load_witness(self.p, C.byref(Fr.from_int(a)), C.byref(o)), "load_witness"); return o

    def add(self, a, b):
        o = self._out(); _ck(self.L.h2w_add(self.p, C.byref(a), C.byref(b), C.byref(o)), "add"); return o

    def mul(self, a, b):
        o = self._out(); _ck(self.L.h2w_mul(self.p, C.byref(a), C.byref(b), C.byref(o)), "mul"); return o

    def mul_add(self, a, b, c):
        o = self._out(); _ck(self.L.h2w_mul_add(self.p, C.byref(a), C.byref(b), C.byref(c), C.byref(o)), "mul_add"); return o

    def select(self, a, b, sel):
        o = self._out(); _ck(self.L.h2w_select(self.p, C.byref(a), C.byref(b), C.byref(sel), C.byref(o)), "select"); return o

    def select_from_idx(self, arr, idx):
        o = self._out(); _ck(self.L.h2w_select_from_idx(self.p, _arr(arr), len(arr), C.byref(idx), C.byref(o)), "select_from_idx"); return o

    def select_array_by_indicator(self, array2d, indicator):
        ln, w = len(array2d), len(array2d[0])
        flat = _arr([x for row in array2d for x in row]); out = (Assigned * w)()
        _ck(self.L.h2w_select_array_by_indicator(self.p, flat, ln, w, _arr(indicator), out), "select_array_by_indicator"); return list(out)

    def idx_to_indicator(self, idx, ln):
        out = (Assigned * ln)(); _ck(self.L.h2w_idx_to_indicator(self.p, C.byref(idx), ln, out), "idx_to_indicator"); return list(out)

    def num_to_bits(self, a, range_bits):
        out = (Assigned * range_bits)(); _ck(self.L.h2w_num_to_bits(self.p, C.byref(a), range_bits, out), "num_to_bits"); return list(out)

    def bits_to_num(self, bits):
        o = self._out(); _ck(self.L.h2w_bits_to_num(self.p, _arr(bits), len(bits), C.byref(o)), "bits_to_num"); return o

    def decompose_le(self, num, limb_bits, num_limbs):
        out = (Assigned * num_limbs)(); _ck(self.L.h2w_decompose_le(self.p, C.byref(num), limb_bits, num_limbs, out), "decompose_le"); return list(out)

    def limbs_to_num(self, limbs, limb_bits):
        o = self._out(); _ck(self.L.h2w_limbs_to_num(self.p, _arr(limbs), len(limbs), limb_bits, C.byref(o)), "limbs_to_num"); return o

    def check_less_than_safe(self, a, b):
        _ck(self.L.h2w_check_less_than_safe(self.p, C.byref(a), b), "check_less_than_safe")

    def range_check(self, a, range_bits):
        _ck(self.L.h2w_range_check(self.p, C.byref(a), range_bits), "range_check")

    def assert_equal(self, a, b):
        _ck(self.L.h2w_constrain_equal(self.p, C.byref(a), C.byref(b)), "assert_equal")


class GoldilocksChip:
    """verifier/src/field/goldilocks/base.rs:46-466 (fused level: hints run inside the library)."""

    def __init__(self, native):
        self.native, self.L, self.p = native, native.L, native.p

    def load_constant(self, a):
        o = Assigned(); _ck(self.L.h2w_gl_load_constant(self.p, a, C.byref(o)), "gl.load_constant"); return o

    def load_witness(self, a):
        o = Assigned(); _ck(self.L.h2w_gl_load_witness(self.p, a, C.byref(o)), "gl.load_witness"); return o

    def reduce(self, a):
        o = Assigned(); _ck(self.L.h2w_gl_reduce(self.p, C.byref(a), C.byref(o)), "gl.reduce"); return o

    def _bin(self, fn, name, a, b):
        o = Assigned(); _ck(fn(self.p, C.byref(a), C.byref(b), C.byref(o)), name); return o

    def add(self, a, b): return self._bin(self.L.h2w_gl_add, "gl.add", a, b)
    def sub(self, a, b): return self._bin(self.L.h2w_gl_sub, "gl.sub", a, b)
    def mul(self, a, b): return self._bin(self.L.h2w_gl_mul, "gl.mul", a, b)
    def div(self, a, b): return self._bin(self.L.h2w_gl_div, "gl.div", a, b)

    def mul_add(self, a, b, c):
        o = Assigned(); _ck(self.L.h2w_gl_mul_add(self.p, C.byref(a), C.byref(b), C.byref(c), C.byref(o)), "gl.mul_add"); return o

    def inv(self, a):
        o = Assigned(); _ck(self.L.h2w_gl_inv(self.p, C.byref(a), C.byref(o)), "gl.inv"); return o

    def mul_sub(self, a, b, c):
        o = Assigned(); _ck(self.L.h2w_gl_mul_sub(self.p, C.byref(a), C.byref(b), C.byref(c), C.byref(o)), "gl.mul_sub"); return o

    def neg(self, a):
        o = Assigned(); _ck(self.L.h2w_gl_neg(self.p, C.byref(a), C.byref(o)), "gl.neg"); return o

    def square(self, a):
        o = Assigned(); _ck(self.L.h2w_gl_square(self.p, C.byref(a), C.byref(o)), "gl.square"); return o

    def exp_power_of_2(self, a, power_log):
        o = Assigned(); _ck(self.L.h2w_gl_exp_power_of_2(self.p, C.byref(a), power_log, C.byref(o)), "gl.exp_power_of_2"); return o


class Plan:
    """Shape-compiled batched hot path (h2w_plan_* / h2w_fri_witness_batch)."""

    def __init__(self, shape, consts, device_id=0, _handle=None):
        self.L = lib()
        self.shape, self.consts, self.device_id = shape, consts, device_id
        self.p = _handle if _handle is not None else self.L.h2w_plan_compile(C.byref(shape), C.byref(consts), device_id)
        if not self.p:
            raise H2WError("h2w_plan_compile: " + last_error())
        self.num_cells = int(self.L.h2w_plan_num_cells(self.p))
        self.proof_words = int(self.L.h2w_plan_proof_words(self.p))
        self.num_records = int(self.L.h2w_plan_num_records(self.p))
        self.num_record_cells = int(self.L.h2w_plan_num_record_cells(self.p))
        self.num_chain_cells = int(self.L.h2w_plan_num_chain_cells(self.p))

    @classmethod
    def from_trace(cls, ctx, proof_words, parallel_scopes=("verify_query_round", "verify_proof_to_cap_with_cap_index"), device_id=0):
        """h2w_plan_from_trace: the tape `ctx` recorded (Context.trace_begin, then ONE run through the level-1 / level-2 calls) as a plan that
        h2w_fri_witness_batch replays on other proofs of the shape.  parallel_scopes: the #[count] scopes whose instances are independent
        (fri/mod.rs:488-501, merkle/mod.rs:57-78); the library checks the claim on the tape."""
        L = lib()
        names = (C.c_char_p * len(parallel_scopes))(*[s.encode() for s in parallel_scopes])
        h = L.h2w_plan_from_trace(ctx.p, proof_words, names, len(parallel_scopes), device_id)
        if not h:
            raise H2WError("h2w_plan_from_trace: " + last_error())
        return cls(None, None, device_id, _handle=h)

    def close(self):
        if getattr(self, "p", None):
            self.L.h2w_plan_free(self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def workspace_bytes(self, n):
        return int(self.L.h2w_plan_workspace_bytes(self.p, n))

    def run(self, proofs_ptr, n, advice_ptr, workspace_ptr, stream=0, emit_stream=None):
        if emit_stream is None:
            _ck(self.L.h2w_fri_witness_batch(self.p, proofs_ptr, n, advice_ptr, workspace_ptr, stream), "h2w_fri_witness_batch")
        else:
            _ck(self.L.h2w_fri_witness_batch2(self.p, proofs_ptr, n, advice_ptr, workspace_ptr, stream, emit_stream), "h2w_fri_witness_batch2")

    def advice_digest(self, advice_ptr, n_cells, digest_ptr, stream=0):
        """32-byte checksum of n_cells cells at advice_ptr into 4 u64 at digest_ptr (device) — h2w_advice_digest."""
        _ck(self.L.h2w_advice_digest(advice_ptr, n_cells, digest_ptr, stream), "h2w_advice_digest")

    def equalities(self):
        """Copy constraints of the plan's cell stream: [(cell a, cell b), ...] (h2w_plan_equalities)."""
        n = int(self.L.h2w_plan_num_equalities(self.p))
        buf = (C.c_uint64 * max(2 * n, 1))()
        _ck(self.L.h2w_plan_equalities(self.p, buf), "h2w_plan_equalities")
        return [(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(n)]

    def const_equalities(self, proof_words=None):
        """Constant equalities [(cell, value), ...] (h2w_plan_const_equalities); proof_words: needed with Goldilocks-Poseidon caps."""
        n = int(self.L.h2w_plan_num_const_equalities(self.p))
        cells = (C.c_uint64 * max(n, 1))(); vals = (Fr * max(n, 1))()
        _ck(self.L.h2w_plan_const_equalities(self.p, proof_words, cells, vals), "h2w_plan_const_equalities")
        return [(int(cells[i]), vals[i].to_int()) for i in range(n)]

    def strand_layout(self):
        """(prologue cells, cells of query block 0, cells of a later query block, cells per proof) — h2w_plan_strand_layout."""
        out = (C.c_uint64 * 4)()
        _ck(self.L.h2w_plan_strand_layout(self.p, out), "h2w_plan_strand_layout")
        return tuple(int(x) for x in out)

    def configure(self, option, value):
        _ck(self.L.h2w_plan_configure(self.p, option, value), "h2w_plan_configure")

    def expand_records(self, n, advice_ptr, workspace_ptr, stream=0):
        """The expansion kernel alone over the records a previous run() left in the workspace (measurement)."""
        _ck(self.L.h2w_fri_expand_records(self.p, n, advice_ptr, workspace_ptr, stream), "h2w_fri_expand_records")

    def status(self, workspace_ptr, n, stream=0):
        st = (C.c_uint32 * n)()
        _ck(self.L.h2w_plan_status(self.p, workspace_ptr, n, st, stream), "h2w_plan_status")
        return list(st)

    # ---- keygen-side metadata and column layout
    def selectors(self):
        """Context::selector of the shape's cell stream as a bytes bitmap (bit i of byte i // 8 = cell i)."""
        buf = (C.c_uint8 * ((self.num_cells + 7) // 8))()
        _ck(self.L.h2w_plan_selectors(self.p, buf), "h2w_plan_selectors")
        return bytes(buf)

    def lookup_cells(self):
        """Cells registered for the range lookup, in registration order."""
        n = int(self.L.h2w_plan_num_lookups(self.p))
        buf = (C.c_uint64 * max(n, 1))()
        _ck(self.L.h2w_plan_lookup_cells(self.p, buf), "h2w_plan_lookup_cells")
        return list(buf[:n])

    def break_points(self, k, unusable_rows=9):
        sel = self.selectors()
        n = C.c_uint64()
        sb = (C.c_uint8 * len(sel)).from_buffer_copy(sel)
        _ck(self.L.h2w_break_points(sb, self.num_cells, k, unusable_rows, None, 0, C.byref(n)), "h2w_break_points")
        out = (C.c_uint64 * max(n.value, 1))()
        _ck(self.L.h2w_break_points(sb, self.num_cells, k, unusable_rows, out, n.value, C.byref(n)), "h2w_break_points")
        return list(out[:n.value])

    def layout_columns(self, advice_ptr, n, break_points, k, columns_ptr, stream=0, proof_stride=None):
        bp = (C.c_uint64 * max(len(break_points), 1))(*break_points)
        _ck(self.L.h2w_layout_columns(advice_ptr, self.num_cells, self.num_cells if proof_stride is None else proof_stride, n, bp, len(break_points), k, columns_ptr, stream), "h2w_layout_columns")

    def run_shard(self, proofs_ptr, n, advice_ptr, workspace_ptr, rank, world, stream=0):
        """This rank's (proof, query) units of the batch (h2w_fri_witness_batch_shard)."""
        _ck(self.L.h2w_fri_witness_batch_shard(self.p, proofs_ptr, n, advice_ptr, workspace_ptr, stream, rank, world), "h2w_fri_witness_batch_shard")

    def run_shard_compact(self, proofs_ptr, n, shard_advice_ptr, workspace_ptr, rank, world, stream=0):
        """The same blocks packed into a buffer of shard_cells(n, rank, world) cells (h2w_fri_witness_batch_shard_compact)."""
        _ck(self.L.h2w_fri_witness_batch_shard_compact(self.p, proofs_ptr, n, shard_advice_ptr, workspace_ptr, stream, rank, world), "h2w_fri_witness_batch_shard_compact")

    def shard_cells(self, n, rank, world):
        return int(self.L.h2w_plan_shard_cells(self.p, n, rank, world))

    def shard_workspace_bytes(self, n, rank, world):
        """Scratch bytes of a sharded call of this rank (the unit buffers hold the rank's own units only)."""
        return int(self.L.h2w_plan_shard_workspace_bytes(self.p, n, rank, world))

    def shard_block(self, rank, world, proof, query):
        """(local cell, cells, global cell) of a block in rank's packed buffer; None when another rank owns it.  query < 0: the prologue block."""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        rc = self.L.h2w_plan_shard_block(self.p, rank, world, proof, query, C.byref(a), C.byref(b), C.byref(c))
        if rc == 1:
            return None
        _ck(rc, "h2w_plan_shard_block")
        return int(a.value), int(b.value), int(c.value)

    def run_columns(self, proofs_ptr, n, break_points, k, columns_ptr, workspace_ptr, stream=0):
        """Batched hot path writing the FlexGate column layout directly (h2w_fri_witness_batch_columns)."""
        bp = (C.c_uint64 * max(len(break_points), 1))(*break_points)
        _ck(self.L.h2w_fri_witness_batch_columns(self.p, proofs_ptr, n, bp, len(break_points), k, columns_ptr, workspace_ptr, stream), "h2w_fri_witness_batch_columns")

    def num_lookup_columns(self, k, unusable_rows=9):
        n = C.c_uint64()
        _ck(self.L.h2w_layout_lookup_columns(self.p, None, 0, 0, k, unusable_rows, None, C.byref(n), None), "h2w_layout_lookup_columns")
        return int(n.value)

    def layout_lookup_columns(self, advice_ptr, n, k, out_ptr, unusable_rows=9, stream=0, proof_stride=None):
        nc = C.c_uint64()
        _ck(self.L.h2w_layout_lookup_columns(self.p, advice_ptr, self.num_cells if proof_stride is None else proof_stride, n, k, unusable_rows, out_ptr, C.byref(nc), stream), "h2w_layout_lookup_columns")
        return int(nc.value)

    def check_constraints(self, advice_ptr, n, stream=0, proof_stride=None):
        """(failed gates, out-of-range lookups) over n advice streams, checked on the device."""
        bad = (C.c_uint64 * 2)()
        _ck(self.L.h2w_check_constraints(self.p, advice_ptr, self.num_cells if proof_stride is None else proof_stride, n, bad, stream), "h2w_check_constraints")
        return int(bad[0]), int(bad[1])

    def check_equalities(self, advice_ptr, n, equalities, const_equalities, stream=0, proof_stride=None):
        """(violated copy constraints, violated constant equalities) over n advice streams; lists from Context.equalities() /
        Context.const_equalities() of an eager keygen context of the same shape."""
        npairs, nconst = len(equalities), len(const_equalities)
        pairs = (C.c_uint64 * max(2 * npairs, 1))(*[x for ab in equalities for x in ab])
        cc = (C.c_uint64 * max(nconst, 1))(*[c for c, _ in const_equalities])
        cv = (Fr * max(nconst, 1))()
        for i, (_, v) in enumerate(const_equalities):
            for j in range(4):
                cv[i].l[j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
        bad = (C.c_uint64 * 2)()
        _ck(self.L.h2w_check_equalities(advice_ptr, self.num_cells, self.num_cells if proof_stride is None else proof_stride, n, pairs, npairs, cc, cv, nconst, bad, stream), "h2w_check_equalities")
        return int(bad[0]), int(bad[1])

    def timing(self, back=0):
        """(prologue ms, strands ms, BN254-unit ms, expansion-kernel ms, total ms) of the batch call `back` calls before
        the last, from HIP events recorded on the call's stream."""
        ms = (C.c_float * 5)()
        _ck(self.L.h2w_plan_timing(self.p, back, ms), "h2w_plan_timing")
        return tuple(ms)

    def timing_ex(self, back=0):
        """Per kernel, ms: (prologue values, permutation records, glue strands, chain values | one-pass chains, chain emission, expansion, whole
        call) and the chain passes the call ran with."""
        ms = (C.c_float * 8)()
        _ck(self.L.h2w_plan_timing_ex(self.p, back, ms), "h2w_plan_timing_ex")
        return tuple(float(x) for x in ms)

    def last_timing(self):
        ms = (C.c_float * 5)()
        _ck(self.L.h2w_plan_last_timing(self.p, ms), "h2w_plan_last_timing")
        return tuple(ms)

    def event_gap(self, back_a, which_a, back_b, which_b):
        """ms from event which_a of the call back_a before the last to event which_b of the call back_b before the last (H2W_EV_*)."""
        ms = C.c_float()
        _ck(self.L.h2w_plan_event_gap(self.p, back_a, which_a, back_b, which_b, C.byref(ms)), "h2w_plan_event_gap")
        return float(ms.value)


class Prover:
    """Synthetic valid FRI instances generated on the GPU (h2w_prover_* / h2w_prove_fri): the step before the
    path.  The reference gets its proofs from starky's prover (stark/mod.rs:405-426); the committed polynomials are inputs."""

    def __init__(self, shape, consts, device_id=0):
        self.L = lib()
        self.shape, self.consts = shape, consts
        self.p = self.L.h2w_prover_new(C.byref(shape), C.byref(consts), device_id)
        if not self.p:
            raise H2WError("h2w_prover_new: " + last_error())
        self.num_polys = int(self.L.h2w_prover_num_polys(self.p))
        self.proof_words = int(self.L.h2w_prover_proof_words(self.p))

    def close(self):
        if getattr(self, "p", None):
            self.L.h2w_prover_free(self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def prove(self, coeffs_ptr, public_inputs, proof_ptr, stream=0):
        """coeffs_ptr: device [num_polys][2^degree_bits] u64; public_inputs: sequence of n_pis ints; proof_ptr: device proof_words u64."""
        pis = (C.c_uint64 * max(len(public_inputs), 1))(*public_inputs)
        _ck(self.L.h2w_prove_fri(self.p, coeffs_ptr, pis, proof_ptr, stream), "h2w_prove_fri")

    def prove_batch(self, coeffs_ptr, public_inputs, proofs_ptr, n, stream=0):
        """n proofs in lockstep: coeffs [n][num_polys][2^degree_bits] (device), public_inputs flat [n][n_pis], proofs [n][proof_words] (device)."""
        pis = (C.c_uint64 * max(len(public_inputs), 1))(*public_inputs)
        _ck(self.L.h2w_prove_fri_batch(self.p, coeffs_ptr, pis, proofs_ptr, n, stream), "h2w_prove_fri_batch")

    def timing(self):
        ms = (C.c_float * 7)()
        _ck(self.L.h2w_prover_timing(self.p, ms), "h2w_prover_timing")
        return dict(zip(("lde", "commit", "openings_quotient", "fri_commit", "pow", "queries", "total_wall"), [float(x) for x in ms]))


def advice_digest_reference(cells_bytes, chunk_cells=1 << 22):
    """The checksum h2w_advice_digest computes (include/h2w.h), restated with numpy over a host copy of a cell stream: limb j of cell i
    is weighted by ((i + 1) * 0x9E3779B97F4A7C15 | 1) + 2 j, sums mod 2^64 per limb position."""
    import numpy as np
    c = cells_bytes if isinstance(cells_bytes, np.ndarray) else np.frombuffer(cells_bytes, dtype=np.uint64).reshape(-1, 4)
    acc = [0, 0, 0, 0]
    with np.errstate(over="ignore"):
        for lo in range(0, c.shape[0], chunk_cells):
            part = c[lo:lo + chunk_cells]
            m = (np.arange(lo + 1, lo + part.shape[0] + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) | np.uint64(1)
            for j in range(4):
                acc[j] = (acc[j] + int((part[:, j] * (m + np.uint64(2 * j))).sum(dtype=np.uint64))) & 0xFFFFFFFFFFFFFFFF
    return acc
