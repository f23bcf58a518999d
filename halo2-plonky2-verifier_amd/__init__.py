"""halo2-plonky2-verifier_amd — MI355X-native witness generation for the plonky2/starky FRI-verifier gadget.

Python is plumbing only: a ctypes binding of the C-ABI shared library (include/h2w.h, csrc/) plus host-side
mirrors of the reference's chip interfaces (chips.py).  All advice cells are produced by HIP kernels; there
is no CPU fallback: loading fails loudly when libh2w.so is missing.

The directory name contains '-', so import it with
    importlib.import_module("halo2-plonky2-verifier_amd")
(tests/conftest.py and bench.py do this and alias it as `h2w_amd`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("H2W_LIB") or os.path.join(_HERE, "libh2w.so")   # H2W_LIB: A/B builds of the same library


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


class Assigned(C.Structure):
    """h2w_assigned_t: mirrors halo2-base AssignedValue{value, cell}."""
    _fields_ = [("value", Fr), ("offset", C.c_uint64), ("ctx_id", C.c_uint32), ("has_cell", C.c_uint32)]

    def int_value(self):
        return self.value.to_int()


class Shape(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "degree_bits", "rate_bits", "cap_height", "num_queries", "pow_bits", "num_challenges",
        "arity_bits", "final_poly_bits", "n_cols", "n_perm_z", "n_quotient", "n_pis",
        "perm_batch_size", "hash_mode", "lookup_bits", "witness_load_range_check")]


class PoseidonConsts(C.Structure):
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


def fibonacci_shape(degree_bits, num_queries, rate_bits=1, cap_height=4, hash_mode=1, lookup_bits=21,
                    witness_load_range_check=1):
    """Fibonacci STARK (reference test_util/fibonacci_stark.rs) under StarkConfig::standard_fast_config."""
    return Shape(degree_bits=degree_bits, rate_bits=rate_bits, cap_height=cap_height, num_queries=num_queries,
                 pow_bits=16, num_challenges=2, arity_bits=4, final_poly_bits=5, n_cols=4, n_perm_z=2,
                 n_quotient=2, n_pis=3, perm_batch_size=1, hash_mode=hash_mode, lookup_bits=lookup_bits,
                 witness_load_range_check=witness_load_range_check)


# every symbol include/h2w.h declares: (restype, argtypes)
_vp = C.c_void_p
_av = C.POINTER(Assigned)
_fr = C.POINTER(Fr)
SYMBOLS = {
    "h2w_abi_version": (C.c_int, []),
    "h2w_poseidon_published": (C.c_int, [C.POINTER(PoseidonConsts)]),
    "h2w_last_error": (C.c_char_p, []),
    "h2w_device_count": (C.c_int, []),
    "h2w_ctx_new": (_vp, [C.c_int, C.c_int, C.c_int]),
    "h2w_ctx_free": (None, [_vp]),
    "h2w_num_cells": (C.c_uint64, [_vp]),
    "h2w_ctx_error": (C.c_int, [_vp]),
    "h2w_push_context": (C.c_int, [_vp, C.c_char_p]),
    "h2w_pop_context": (C.c_int, [_vp]),
    "h2w_context_dump": (C.c_size_t, [_vp, C.c_char_p, C.c_size_t]),
    "h2w_load_constant": (C.c_int, [_vp, _fr, _av]),
    "h2w_load_zero": (C.c_int, [_vp, _av]),
    "h2w_load_constants": (C.c_int, [_vp, _fr, C.c_size_t, _av]),
    "h2w_load_witness": (C.c_int, [_vp, _fr, _av]),
    "h2w_add": (C.c_int, [_vp, _av, _av, _av]),
    "h2w_mul": (C.c_int, [_vp, _av, _av, _av]),
    "h2w_mul_add": (C.c_int, [_vp, _av, _av, _av, _av]),
    "h2w_select": (C.c_int, [_vp, _av, _av, _av, _av]),
    "h2w_select_from_idx": (C.c_int, [_vp, _av, C.c_size_t, _av, _av]),
    "h2w_select_array_by_indicator": (C.c_int, [_vp, _av, C.c_size_t, C.c_size_t, _av, _av]),
    "h2w_idx_to_indicator": (C.c_int, [_vp, _av, C.c_size_t, _av]),
    "h2w_num_to_bits": (C.c_int, [_vp, _av, C.c_size_t, _av]),
    "h2w_bits_to_num": (C.c_int, [_vp, _av, C.c_size_t, _av]),
    "h2w_decompose_le": (C.c_int, [_vp, _av, C.c_size_t, C.c_size_t, _av]),
    "h2w_limbs_to_num": (C.c_int, [_vp, _av, C.c_size_t, C.c_size_t, _av]),
    "h2w_check_less_than_safe": (C.c_int, [_vp, _av, C.c_uint64]),
    "h2w_range_check": (C.c_int, [_vp, _av, C.c_size_t]),
    "h2w_constrain_equal": (C.c_int, [_vp, _av, _av]),
    "h2w_gl_load_constant": (C.c_int, [_vp, C.c_uint64, _av]),
    "h2w_gl_load_witness": (C.c_int, [_vp, C.c_uint64, _av]),
    "h2w_gl_reduce": (C.c_int, [_vp, _av, _av]),
    "h2w_gl_add": (C.c_int, [_vp, _av, _av, _av]),
    "h2w_gl_sub": (C.c_int, [_vp, _av, _av, _av]),
    "h2w_gl_mul": (C.c_int, [_vp, _av, _av, _av]),
    "h2w_gl_mul_add": (C.c_int, [_vp, _av, _av, _av, _av]),
    "h2w_gl_div": (C.c_int, [_vp, _av, _av, _av]),
    "h2w_gl_inv": (C.c_int, [_vp, _av, _av]),
    "h2w_gl_mul_sub": (C.c_int, [_vp, _av, _av, _av, _av]),
    "h2w_gl_neg": (C.c_int, [_vp, _av, _av]),
    "h2w_gl_square": (C.c_int, [_vp, _av, _av]),
    "h2w_gl_exp_power_of_2": (C.c_int, [_vp, _av, C.c_size_t, _av]),
    "h2w_gl_ext_inv_witness": (C.c_int, [_vp, _av, _av]),
    "h2w_ctx_trace_begin": (C.c_int, [_vp]),
    "h2w_ctx_reset": (C.c_int, [_vp]),
    "h2w_ctx_footprint": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "h2w_ctx_reserve": (C.c_int, [_vp, C.c_uint64, C.c_uint64]),
    "h2w_trace_input": (C.c_int, [_vp, C.c_uint64, C.c_uint32]),
    "h2w_plan_from_trace": (_vp, [_vp, C.c_uint64, C.POINTER(C.c_char_p), C.c_size_t, C.c_int]),
    "h2w_chip_ext_op": (C.c_int, [_vp, C.c_int, _av, _av, _av, _av]),
    "h2w_chip_gl_exp_from_bits_const_base": (C.c_int, [_vp, C.c_uint64, _av, C.c_size_t, _av]),
    "h2w_chip_gl_poseidon_permute": (C.c_int, [_vp, C.POINTER(PoseidonConsts), _av, _av]),
    "h2w_chip_bn_poseidon_permute": (C.c_int, [_vp, C.POINTER(PoseidonConsts), _av, _av]),
    "h2w_chip_hash_no_pad": (C.c_int, [_vp, C.POINTER(PoseidonConsts), C.c_int, _av, C.c_size_t, _av]),
    "h2w_chip_two_to_one": (C.c_int, [_vp, C.POINTER(PoseidonConsts), C.c_int, _av, _av, _av]),
    "h2w_chip_merkle_verify": (C.c_int, [_vp, C.POINTER(PoseidonConsts), C.c_int, _av, C.c_size_t, _av, C.c_size_t, _av, _av, C.c_size_t, _av, C.c_size_t]),
    "h2w_chip_verify_stark": (C.c_int, [_vp, C.POINTER(Shape), C.POINTER(PoseidonConsts), C.POINTER(C.c_uint64)]),
    "h2w_ctx_advice_device": (C.c_int, [_vp, C.POINTER(_vp)]),
    "h2w_ctx_download": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _fr]),
    "h2w_plan_compile": (_vp, [C.POINTER(Shape), C.POINTER(PoseidonConsts), C.c_int]),
    "h2w_plan_free": (None, [_vp]),
    "h2w_plan_num_cells": (C.c_uint64, [_vp]),
    "h2w_plan_proof_words": (C.c_uint64, [_vp]),
    "h2w_plan_num_records": (C.c_uint64, [_vp]),
    "h2w_plan_workspace_bytes": (C.c_uint64, [_vp, C.c_uint64]),
    "h2w_fri_witness_batch": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, _vp]),
    "h2w_fri_expand_records": (C.c_int, [_vp, C.c_uint64, _vp, _vp, _vp]),
    "h2w_fri_witness_batch2": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, _vp, _vp]),
    "h2w_plan_num_gates": (C.c_uint64, [_vp]),
    "h2w_plan_num_lookups": (C.c_uint64, [_vp]),
    "h2w_plan_selectors": (C.c_int, [_vp, _vp]),
    "h2w_plan_lookup_cells": (C.c_int, [_vp, _vp]),
    "h2w_break_points": (C.c_int, [_vp, C.c_uint64, C.c_int, C.c_int, _vp, C.c_uint64, _vp]),
    "h2w_layout_columns": (C.c_int, [_vp, C.c_uint64, C.c_uint64, C.c_uint64, _vp, C.c_uint64, C.c_int, _vp, _vp]),
    "h2w_fri_witness_batch_columns": (C.c_int, [_vp, _vp, C.c_uint64, _vp, C.c_uint64, C.c_int, _vp, _vp, _vp]),
    "h2w_layout_lookup_columns": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, C.c_int, C.c_int, _vp, _vp, _vp]),
    "h2w_check_constraints": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, _vp, _vp]),
    "h2w_fri_witness_batch_shard": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, _vp, C.c_int, C.c_int]),
    "h2w_advice_to_montgomery": (C.c_int, [_vp, C.c_uint64, _vp]),
    "h2w_ctx_num_gates": (C.c_uint64, [_vp]), "h2w_ctx_gate_cells": (C.c_int, [_vp, _vp]),
    "h2w_ctx_num_lookups": (C.c_uint64, [_vp]), "h2w_ctx_lookup_cells": (C.c_int, [_vp, _vp]),
    "h2w_ctx_num_equalities": (C.c_uint64, [_vp]), "h2w_ctx_equalities": (C.c_int, [_vp, _vp]),
    "h2w_ctx_num_const_equalities": (C.c_uint64, [_vp]), "h2w_ctx_const_equalities": (C.c_int, [_vp, _vp, _vp]),
    "h2w_check_equalities": (C.c_int, [_vp, C.c_uint64, C.c_uint64, C.c_uint64, _vp, C.c_uint64, _vp, _vp, C.c_uint64, _vp, _vp]),
    "h2w_plan_status": (C.c_int, [_vp, _vp, C.c_uint64, C.POINTER(C.c_uint32), _vp]),
    "h2w_advice_digest": (C.c_int, [_vp, C.c_uint64, _vp, _vp]),
    "h2w_plan_last_timing": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "h2w_plan_event_gap": (C.c_int, [_vp, C.c_uint64, C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_float)]),
    "h2w_plan_configure": (C.c_int, [_vp, C.c_int, C.c_int]),
    "h2w_plan_num_chain_cells": (C.c_uint64, [_vp]),
    "h2w_plan_strand_layout": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "h2w_plan_num_equalities": (C.c_uint64, [_vp]),
    "h2w_plan_num_const_equalities": (C.c_uint64, [_vp]),
    "h2w_plan_equalities": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "h2w_plan_const_equalities": (C.c_int, [_vp, _vp, C.POINTER(C.c_uint64), _vp]),
    "h2w_chipbatch_new": (_vp, [C.c_int, C.c_int, C.c_int]),
    "h2w_chipbatch_free": (None, [_vp]),
    "h2w_chipbatch_num_operands": (C.c_uint64, [_vp]),
    "h2w_chipbatch_num_cells": (C.c_uint64, [_vp]),
    "h2w_chipbatch_run": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, _vp]),
    "h2w_comm_unique_id": (C.c_int, [_vp]),
    "h2w_comm_init": (_vp, [_vp, C.c_int, C.c_int, C.c_int]),
    "h2w_comm_free": (None, [_vp]),
    "h2w_comm_rank": (C.c_int, [_vp]),
    "h2w_comm_world": (C.c_int, [_vp]),
    "h2w_comm_broadcast_proofs": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int, _vp]),
    "h2w_comm_allgather_digests": (C.c_int, [_vp, _vp, _vp, _vp]),
    "h2w_plan_timing": (C.c_int, [_vp, C.c_uint64, C.POINTER(C.c_float)]),
    "h2w_plan_timing_ex": (C.c_int, [_vp, C.c_uint64, C.POINTER(C.c_float)]),
    "h2w_plan_shard_cells": (C.c_uint64, [_vp, C.c_uint64, C.c_int, C.c_int]),
    "h2w_plan_shard_workspace_bytes": (C.c_uint64, [_vp, C.c_uint64, C.c_int, C.c_int]),
    "h2w_plan_shard_block": (C.c_int, [_vp, C.c_int, C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "h2w_fri_witness_batch_shard_compact": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, _vp, C.c_int, C.c_int]),
    "h2w_plan_num_record_cells": (C.c_uint64, [_vp]),
    "h2w_prover_new": (_vp, [C.POINTER(Shape), C.POINTER(PoseidonConsts), C.c_int]),
    "h2w_prover_free": (None, [_vp]),
    "h2w_prover_num_polys": (C.c_uint64, [_vp]),
    "h2w_prover_proof_words": (C.c_uint64, [_vp]),
    "h2w_prove_fri": (C.c_int, [_vp, _vp, C.POINTER(C.c_uint64), _vp, _vp]),
    "h2w_prove_fri_batch": (C.c_int, [_vp, _vp, C.POINTER(C.c_uint64), _vp, C.c_uint64, _vp]),
    "h2w_prover_timing": (C.c_int, [_vp, C.POINTER(C.c_float)]),
}

_lib = None


class H2WError(RuntimeError):
    pass


def _preload_hip_runtime():
    """One process must hold ONE HIP runtime.  PyTorch-ROCm bundles its own libamdhip64.so (soname
    libamdhip64.so.7, same as /opt/rocm's); if libh2w.so pulled in /opt/rocm's copy first, a later `import torch`
    would bring a second runtime and neither would see the GPU.  So when torch is installed, map its copy first:
    libh2w.so's DT_NEEDED libamdhip64.so.7 then binds to it by soname, and device pointers / streams are shared."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load libh2w.so (built in-tree by build.sh / __graft_entry__.build()).  Fails loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise H2WError(f"{LIB_PATH} not found: build the HIP extension first (./build.sh). There is no CPU fallback.")
    _preload_hip_runtime()
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def last_error():
    return lib().h2w_last_error().decode()


def _ck(rc, what):
    if rc != 0:
        raise H2WError(f"{what}: {last_error()}")


def published_consts():
    """The published Poseidon parameter sets (plonky2 Goldilocks width 12; circomlib / plonky2x BN254 t = 4) the reference links in
    from its dependencies (hash/poseidon/permutation.rs:2-7, hash/poseidon_bn254/permutation.rs:7-11): h2w_poseidon_published."""
    k = PoseidonConsts()
    _ck(lib().h2w_poseidon_published(C.byref(k)), "h2w_poseidon_published")
    return k
