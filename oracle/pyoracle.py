"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY (see oracle/oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("oracle.c", "oracle.h", "oracle_field.h", "prover.inc")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src if os.path.exists(s)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


class Fr(C.Structure):
    _fields_ = [("l", C.c_uint64 * 4)]

    @staticmethod
    def from_int(x):
        f = Fr()
        for i in range(4):
            f.l[i] = (x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF
        return f

    def to_int(self):
        return sum(int(self.l[i]) << (64 * i) for i in range(4))


class Shape(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "degree_bits", "rate_bits", "cap_height", "num_queries", "pow_bits", "num_challenges",
        "arity_bits", "final_poly_bits", "n_cols", "n_perm_z", "n_quotient", "n_pis",
        "perm_batch_size", "hash_mode", "lookup_bits", "witness_load_range_check")]


def fibonacci_shape(degree_bits, num_queries, rate_bits=1, cap_height=4, hash_mode=1, lookup_bits=21,
                    witness_load_range_check=1):
    """Fibonacci STARK (test_util/fibonacci_stark.rs:60-61,129-131) under StarkConfig::standard_fast_config
     with the given FRI overrides."""
    return Shape(degree_bits=degree_bits, rate_bits=rate_bits, cap_height=cap_height, num_queries=num_queries,
                 pow_bits=16, num_challenges=2, arity_bits=4, final_poly_bits=5, n_cols=4, n_perm_z=2,
                 n_quotient=2, n_pis=3, perm_batch_size=1, hash_mode=hash_mode, lookup_bits=lookup_bits,
                 witness_load_range_check=witness_load_range_check)


class Consts(C.Structure):
    _fields_ = [
        ("all_round_constants", C.c_uint64 * 360),
        ("mds_circ", C.c_uint64 * 12),
        ("mds_diag", C.c_uint64 * 12),
        ("fast_partial_first_round_constant", C.c_uint64 * 12),
        ("fast_partial_round_constants", C.c_uint64 * 22),
        ("fast_partial_round_initial_matrix", (C.c_uint64 * 11) * 11),
        ("fast_partial_round_w_hats", (C.c_uint64 * 11) * 22),
        ("fast_partial_round_vs", (C.c_uint64 * 11) * 22),
        ("bn_c", Fr * 88),
        ("bn_s", Fr * 392),
        ("bn_m", (Fr * 4) * 4),
        ("bn_p", (Fr * 4) * 4),
    ]


class AV(C.Structure):
    _fields_ = [("v", Fr), ("cell", C.c_int64)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB)
    vp = C.c_void_p
    L.orc_ctx_new.restype = vp
    L.orc_ctx_new.argtypes = [C.c_int, C.c_int, C.c_int]
    L.orc_ctx_free.argtypes = [vp]
    L.orc_ctx_new_streaming.restype = vp
    L.orc_ctx_new_streaming.argtypes = [C.c_int]
    L.orc_digest.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.orc_ctx_reserve.argtypes = [vp, C.c_uint64]
    L.orc_num_cells.restype = C.c_uint64
    L.orc_num_cells.argtypes = [vp]
    L.orc_advice.restype = C.POINTER(Fr)
    L.orc_advice.argtypes = [vp]
    L.orc_error.restype = C.c_char_p
    L.orc_error.argtypes = [vp]
    L.orc_mock_prover.restype = C.c_int
    L.orc_mock_prover.argtypes = [vp] + [C.POINTER(C.c_uint64)] * 4
    L.orc_scope_dump.restype = C.c_size_t
    L.orc_scope_dump.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.orc_watch_cell.argtypes = [vp, C.c_uint64]
    L.orc_watch_path.restype = C.c_char_p
    L.orc_watch_path.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.orc_synth_consts.argtypes = [C.POINTER(Consts), C.c_uint64]
    L.orc_proof_words.restype = C.c_size_t
    L.orc_proof_words.argtypes = [C.POINTER(Shape)]
    L.orc_synth_proof.argtypes = [C.POINTER(Shape), C.c_uint64, C.POINTER(C.c_uint64)]
    L.orc_num_gates.restype = C.c_uint64; L.orc_num_gates.argtypes = [vp]
    L.orc_selector_bitmap.argtypes = [vp, vp]
    L.orc_num_equalities.restype = C.c_uint64; L.orc_num_equalities.argtypes = [vp]
    L.orc_equalities.argtypes = [vp, vp]
    L.orc_num_const_equalities.restype = C.c_uint64; L.orc_num_const_equalities.argtypes = [vp]
    L.orc_const_equalities.argtypes = [vp, vp, vp]
    L.orc_num_lookups.restype = C.c_uint64; L.orc_num_lookups.argtypes = [vp]
    L.orc_lookup_cells.argtypes = [vp, vp]
    L.orc_break_points.restype = C.c_uint64; L.orc_break_points.argtypes = [vp, C.c_int, C.c_int, vp, C.c_uint64]
    L.orc_layout_columns.argtypes = [vp, vp, C.c_uint64, C.c_int, vp]
    L.orc_layout_lookup_columns.restype = C.c_uint64; L.orc_layout_lookup_columns.argtypes = [vp, C.c_int, C.c_int, vp]
    L.orc_prove_fri.restype = C.c_int
    L.orc_prove_fri.argtypes = [C.POINTER(Shape), C.POINTER(Consts), C.c_uint64, C.POINTER(C.c_uint64)]
    L.orc_prove_fri_inputs.restype = C.c_size_t
    L.orc_prove_fri_inputs.argtypes = [C.POINTER(Shape), C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.orc_prove_fri_coef.restype = C.c_int
    L.orc_prove_fri_coef.argtypes = [C.POINTER(Shape), C.POINTER(Consts), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.orc_nv_gl_permute.argtypes = [C.POINTER(Consts), C.POINTER(C.c_uint64)]
    L.orc_nv_bn_permute.argtypes = [C.POINTER(Consts), C.POINTER(Fr)]
    L.orc_nv_hash_or_noop.argtypes = [C.POINTER(Consts), C.c_int, C.POINTER(C.c_uint64), C.c_int, C.POINTER(C.c_uint64)]
    L.orc_verify_stark.restype = C.c_int
    L.orc_verify_stark.argtypes = [vp, C.POINTER(Shape), C.POINTER(Consts), C.POINTER(C.c_uint64)]
    frp = C.POINTER(Fr)
    avp = C.POINTER(AV)
    sigs = {
        "orc_load_witness": (AV, [vp, frp]), "orc_load_constant": (AV, [vp, frp]), "orc_load_zero": (AV, [vp]),
        "orc_add": (AV, [vp, AV, AV]), "orc_mul": (AV, [vp, AV, AV]), "orc_mul_add": (AV, [vp, AV, AV, AV]),
        "orc_select": (AV, [vp, AV, AV, AV]), "orc_select_from_idx": (AV, [vp, avp, C.c_int, AV]),
        "orc_idx_to_indicator": (None, [vp, AV, C.c_int, avp]),
        "orc_select_array_by_indicator": (None, [vp, avp, C.c_int, C.c_int, avp, avp]),
        "orc_num_to_bits": (None, [vp, AV, C.c_int, avp]), "orc_bits_to_num": (AV, [vp, avp, C.c_int]),
        "orc_decompose_le": (None, [vp, AV, C.c_int, C.c_int, avp]),
        "orc_limbs_to_num": (AV, [vp, avp, C.c_int, C.c_int]),
        "orc_check_less_than_safe": (None, [vp, AV, C.c_uint64]), "orc_range_check": (None, [vp, AV, C.c_int]),
        "orc_constrain_equal": (None, [vp, AV, AV]),
        "orc_gl_load_witness": (AV, [vp, C.c_uint64]), "orc_gl_load_constant": (AV, [vp, C.c_uint64]),
        "orc_gl_reduce": (AV, [vp, AV]), "orc_gl_add": (AV, [vp, AV, AV]), "orc_gl_sub": (AV, [vp, AV, AV]),
        "orc_gl_mul": (AV, [vp, AV, AV]), "orc_gl_mul_add": (AV, [vp, AV, AV, AV]),
        "orc_gl_mul_sub": (AV, [vp, AV, AV, AV]), "orc_gl_div": (AV, [vp, AV, AV]), "orc_gl_inv": (AV, [vp, AV]),
        "orc_gl_exp_from_bits_const_base": (AV, [vp, C.c_uint64, avp, C.c_int]),
        "orc_gl_exp_power_of_2": (AV, [vp, AV, C.c_int]),
        "orc_ext_mul": (None, [vp, avp, avp, avp]), "orc_ext_inv": (None, [vp, avp, avp]),
        "orc_ext_div": (None, [vp, avp, avp, avp]),
        "orc_gl_poseidon_permute": (None, [vp, C.POINTER(Consts), avp, avp]),
        "orc_bn_poseidon_permute": (None, [vp, C.POINTER(Consts), avp, avp]),
        "orc_hash_no_pad": (None, [vp, C.POINTER(Consts), C.c_int, avp, C.c_int, avp]),
        "orc_two_to_one": (None, [vp, C.POINTER(Consts), C.c_int, avp, avp, avp]),
        "orc_merkle_verify": (None, [vp, C.POINTER(Consts), C.c_int, avp, C.c_int, avp, C.c_int, AV, avp, C.c_int, avp, C.c_int]),
        "orc_glf_mul": (C.c_uint64, [C.c_uint64, C.c_uint64]), "orc_glf_inv": (C.c_uint64, [C.c_uint64]),
        "orc_glf_exp": (C.c_uint64, [C.c_uint64, C.c_uint64]),
        "orc_glf_primitive_root_of_unity": (C.c_uint64, [C.c_int]),
        "orc_fr_mul": (None, [frp, frp, frp]), "orc_fr_inv": (None, [frp, frp]), "orc_fr_modulus": (None, [frp]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


class Ctx:
    """An oracle context (halo2-base Context restated)."""

    def __init__(self, lookup_bits=21, witness_gen_only=True, track_scopes=False, streaming=False):
        self.L = lib()
        # streaming: for streams too long for the host - only a ring of the last cells is kept, the stream is summed into the checksum of h2w_advice_digest
        self.p = self.L.orc_ctx_new_streaming(lookup_bits) if streaming else self.L.orc_ctx_new(lookup_bits, 1 if witness_gen_only else 0, 1 if track_scopes else 0)

    def digest(self):
        out = (C.c_uint64 * 4)()
        self.L.orc_digest(self.p, out)
        return [int(x) for x in out]

    def close(self):
        if self.p:
            self.L.orc_ctx_free(self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reserve(self, ncells):
        self.L.orc_ctx_reserve(self.p, ncells)

    def num_cells(self):
        return int(self.L.orc_num_cells(self.p))

    def advice_bytes(self):
        n = self.num_cells()
        return C.string_at(self.L.orc_advice(self.p), n * 32)

    def advice_array(self):
        """The advice as an (n, 4) uint64 numpy view of the context's own memory (no copy; valid until the context changes)."""
        import numpy as np
        n = self.num_cells()
        return np.ctypeslib.as_array(C.cast(self.L.orc_advice(self.p), C.POINTER(C.c_uint64)), shape=(n, 4))

    def error(self):
        return self.L.orc_error(self.p).decode()

    def watch_cell(self, cell):
        """Before a run (track_scopes=True): remember the #[count] call stack that appends `cell`."""
        self.L.orc_watch_cell(self.p, cell)

    def watch_path(self):
        off = C.c_uint64()
        return self.L.orc_watch_path(self.p, C.byref(off)).decode(), int(off.value)

    def scopes(self):
        n = self.L.orc_scope_dump(self.p, None, 0)
        buf = C.create_string_buffer(n + 1)
        self.L.orc_scope_dump(self.p, buf, n + 1)
        out = {}
        for line in buf.value.decode().splitlines():
            path, cells = line.rsplit(" ", 1)
            out[path] = int(cells)
        return out

    # keygen metadata / column layout (witness_gen_only=False contexts)
    def selectors(self):
        buf = (C.c_uint8 * ((self.num_cells() + 7) // 8))()
        self.L.orc_selector_bitmap(self.p, buf)
        return bytes(buf)

    def lookup_cells(self):
        n = int(self.L.orc_num_lookups(self.p))
        buf = (C.c_uint64 * max(n, 1))()
        self.L.orc_lookup_cells(self.p, buf)
        return list(buf[:n])

    def equalities(self):
        n = int(self.L.orc_num_equalities(self.p))
        buf = (C.c_uint64 * max(2 * n, 1))()
        self.L.orc_equalities(self.p, buf)
        return [(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(n)]

    def const_equalities(self):
        n = int(self.L.orc_num_const_equalities(self.p))
        cells = (C.c_uint64 * max(n, 1))(); vals = (Fr * max(n, 1))()
        self.L.orc_const_equalities(self.p, cells, vals)
        return [(int(cells[i]), vals[i].to_int()) for i in range(n)]

    def gate_cells(self):
        import numpy as np
        sel = np.unpackbits(np.frombuffer(self.selectors(), dtype=np.uint8), bitorder="little")
        return [int(i) for i in np.nonzero(sel)[0]]

    def break_points(self, k, unusable_rows=9):
        n = int(self.L.orc_break_points(self.p, k, unusable_rows, None, 0))
        buf = (C.c_uint64 * max(n, 1))()
        self.L.orc_break_points(self.p, k, unusable_rows, buf, n)
        return list(buf[:n])

    def layout_columns(self, bp, k):
        out = (Fr * ((len(bp) + 1) << k))()
        self.L.orc_layout_columns(self.p, (C.c_uint64 * max(len(bp), 1))(*bp), len(bp), k, out)
        return bytes(out)

    def layout_lookup_columns(self, k, unusable_rows=9):
        n = int(self.L.orc_layout_lookup_columns(self.p, k, unusable_rows, None))
        out = (Fr * max(n << k, 1))()
        self.L.orc_layout_lookup_columns(self.p, k, unusable_rows, out)
        return n, bytes(out)[:(n << k) * 32]

    def mock_prover(self):
        g, e, l, s = (C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64())
        bad = self.L.orc_mock_prover(self.p, C.byref(g), C.byref(e), C.byref(l), C.byref(s))
        return dict(bad=bad, gates=g.value, equalities=e.value, lookups=l.value, semantic_failed=s.value)


def synth_consts(seed=0xC0FFEE):
    k = Consts()
    lib().orc_synth_consts(C.byref(k), seed)
    return k


def published_consts(path=None):
    """The published Poseidon parameter sets (tests/golden/poseidon_published.json, made and known-answer-checked by
    tools/gen_poseidon_constants.py) as the oracle's constant block."""
    import json
    path = path or os.path.join(_HERE, "..", "tests", "golden", "poseidon_published.json")
    j = json.load(open(path))
    g, b = j["goldilocks_w12"], j["bn254_t4"]
    as_int = lambda x: int(x, 16) if isinstance(x, str) else int(x)
    k = Consts()
    for name in ("all_round_constants", "mds_circ", "mds_diag", "fast_partial_first_round_constant", "fast_partial_round_constants"):
        for i, v in enumerate(g[name]):
            getattr(k, name)[i] = as_int(v)
    for name in ("fast_partial_round_initial_matrix", "fast_partial_round_w_hats", "fast_partial_round_vs"):
        for i, row in enumerate(g[name]):
            for jx, v in enumerate(row):
                getattr(k, name)[i][jx] = as_int(v)
    for i, v in enumerate(b["C"]):
        k.bn_c[i] = Fr.from_int(as_int(v))
    for i, v in enumerate(b["S"]):
        k.bn_s[i] = Fr.from_int(as_int(v))
    for i in range(4):
        for jx in range(4):
            k.bn_m[i][jx] = Fr.from_int(as_int(b["M"][i][jx]))
            k.bn_p[i][jx] = Fr.from_int(as_int(b["P"][i][jx]))
    return k


def synth_proof(shape, seed):
    n = lib().orc_proof_words(C.byref(shape))
    buf = (C.c_uint64 * n)()
    lib().orc_synth_proof(C.byref(shape), seed, buf)
    return buf


def prove_fri(shape, consts, seed):
    """A valid FRI instance of the shape (oracle/prover.inc), flat proof words."""
    n = lib().orc_proof_words(C.byref(shape))
    buf = (C.c_uint64 * n)()
    rc = lib().orc_prove_fri(C.byref(shape), C.byref(consts), seed, buf)
    if rc != 0:
        raise RuntimeError("orc_prove_fri: unsupported shape")
    return buf


def prove_fri_inputs(shape, seed):
    """The committed polynomials' coefficients ([polynomial][2^degree_bits]) and public inputs orc_prove_fri(seed) uses."""
    n = lib().orc_prove_fri_inputs(C.byref(shape), seed, None, None)
    coefs = (C.c_uint64 * n)(); pis = (C.c_uint64 * max(shape.n_pis, 1))()
    lib().orc_prove_fri_inputs(C.byref(shape), seed, coefs, pis)
    return coefs, pis


def prove_fri_coef(shape, consts, coefs, pis):
    n = lib().orc_proof_words(C.byref(shape))
    buf = (C.c_uint64 * n)()
    if lib().orc_prove_fri_coef(C.byref(shape), C.byref(consts), coefs, pis, buf) != 0:
        raise RuntimeError("orc_prove_fri_coef: unsupported shape")
    return buf


def verify_stark(ctx, shape, consts, proof_words):
    return lib().orc_verify_stark(ctx.p, C.byref(shape), C.byref(consts), proof_words)
