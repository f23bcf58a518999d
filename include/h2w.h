/*
 * h2w.h — C ABI of the MI355X-native halo2 witness-generation engine for the plonky2/starky
 * FRI-verifier gadget (drop-in for ONE hot path of shuklaayush/halo2-plonky2-verifier).
 *
 * The boundary sits below the reference's NativeChip (verifier/src/field/native.rs:11-194) /
 * ContextWrapper.ctx (verifier/src/util/context_wrapper.rs:11-26), i.e. where the Rust code calls
 * halo2-base Context / GateChip / RangeChip.  Three API levels, all producing the SAME advice stream:
 *
 *   1. eager NativeChip level  (h2w_load_*, h2w_add ... h2w_range_check)   — 1:1 with field/native.rs
 *   2. fused Goldilocks level  (h2w_gl_*)                                  — field/goldilocks/base.rs ops
 *      (hints — u128 divmod, GL inverse — run inside the library)
 *   3. batched hot path        (h2w_plan_* / h2w_fri_witness_batch)        — whole
 *      load_proof_with_pis + StarkChip::verify_proof (stark/mod.rs:483-508) for a batch of proofs,
 *      values and cells computed on the GPU.
 *
 * Plain C: pointers and sizes only, no C++/torch types.  All functions returning int return 0 on
 * success; on failure h2w_last_error() describes it (the reference panics instead; the Rust shim
 * in INTEGRATION.md converts non-zero to panic!).  Handles are not thread-safe; distinct handles
 * are independent (the reference is single-threaded: one &mut Context) and carry their device: a call
 * makes the handle's device current for its own duration (functions without a handle use the device their
 * device pointer lives on), so one thread can drive plans on several GPUs.
 *
 * Advice cells are BN254 Fr values, 32 bytes each, canonical little-endian (4 x u64 limbs).
 */
#ifndef H2W_H
#define H2W_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define H2W_ABI_VERSION 1

typedef struct { uint64_t l[4]; } h2w_fr_t;

/* mirrors halo2-base AssignedValue{value, cell: Option<ContextCell{context_id, offset}>} */
typedef struct {
    h2w_fr_t value;
    uint64_t offset;     /* row in Context::advice */
    uint32_t ctx_id;
    uint32_t has_cell;   /* 0 = value only (QuantumCell::Witness/Constant not yet assigned) */
} h2w_assigned_t;

/* STARK / FRI shape: starky StarkConfig + the Stark trait facts the verifier needs
 * (stark/mod.rs:145-200, challenger/mod.rs:168-222, fri/mod.rs:447-502). */
typedef struct {
    int32_t degree_bits, rate_bits, cap_height, num_queries, pow_bits, num_challenges;
    int32_t arity_bits, final_poly_bits;          /* FriReductionStrategy::ConstantArityBits */
    int32_t n_cols, n_perm_z, n_quotient, n_pis;  /* Fibonacci: 4, 2, 2, 3 */
    int32_t perm_batch_size;
    int32_t hash_mode;                            /* 0 = Goldilocks Poseidon Merkle, 1 = PoseidonBN254 Merkle */
    int32_t lookup_bits;                          /* RangeChip lookup_bits (= k-1) */
    int32_t witness_load_range_check;             /* 1 = witness/mod.rs:49-51 (28 cells / GL element) */
} h2w_shape_t;

/* Poseidon constants are caller inputs (plonky2 `hash::poseidon`, plonky2x `poseidon_bn128_constants`);
 * indexing as hash/poseidon/permutation.rs:55-68,97-104,120-130,143-171,184-192,229-235 and
 * hash/poseidon_bn254/permutation.rs:86-109,138-159,163-170. */
typedef struct {
    uint64_t all_round_constants[360];
    uint64_t mds_circ[12];
    uint64_t mds_diag[12];
    uint64_t fast_partial_first_round_constant[12];
    uint64_t fast_partial_round_constants[22];
    uint64_t fast_partial_round_initial_matrix[11][11];
    uint64_t fast_partial_round_w_hats[22][11];
    uint64_t fast_partial_round_vs[22][11];
    h2w_fr_t bn_c[88];
    h2w_fr_t bn_s[392];
    h2w_fr_t bn_m[4][4];
    h2w_fr_t bn_p[4][4];
} h2w_poseidon_consts_t;

typedef struct h2w_ctx h2w_ctx;
typedef struct h2w_plan h2w_plan;

/* ------------------------------------------------------------------ library */
int         h2w_abi_version(void);
const char *h2w_last_error(void);
int         h2w_device_count(void);              /* number of visible HIP devices (0 = none: every compute call fails) */
/* The published parameter sets the reference links in from its dependencies: plonky2's Goldilocks Poseidon (width 12:
 * ALL_ROUND_CONSTANTS, MDS_MATRIX_CIRC/_DIAG, FAST_PARTIAL_*; hash/poseidon/permutation.rs:2-7) and plonky2x's
 * circomlib t = 4 PoseidonBN254 tables C/S/M/P (hash/poseidon_bn254/permutation.rs:7-11).  Pure data (no device needed);
 * a caller may pass any other tables of the same layout instead. */
int         h2w_poseidon_published(h2w_poseidon_consts_t *out);

/* ------------------------------------------------------------------ 1. eager NativeChip level
 * replaces halo2-base Context + GateChip + RangeChip as used by field/native.rs.
 * Values are computed on the host at call time (the reference reads AssignedValue::value() for hints:
 * base.rs:27-35,349,382); cells are materialised on the GPU from compact records when the advice is
 * requested (h2w_ctx_advice_device / h2w_ctx_download). */
h2w_ctx *h2w_ctx_new(int lookup_bits, int witness_gen_only, int device_id);   /* base_test().k(k): lookup_bits = k-1 */
void     h2w_ctx_free(h2w_ctx *);
int      h2w_ctx_reset(h2w_ctx *);                                              /* the context as new, its host memory kept (the next proof's run does not fault its pages in again) */
int      h2w_ctx_footprint(const h2w_ctx *, uint64_t out[2]);                    /* what the run so far appended: out[0] block records, out[1] literal cells (32 B each) */
int      h2w_ctx_reserve(h2w_ctx *, uint64_t n_records, uint64_t n_literal_cells); /* Vec::with_capacity for a NEW context: its host vectors sized and mapped ahead (a cfg-3 PoseidonBN254
                                                                                  * proof appends 350 MB; on a fresh context two thirds of a level-1 run are first-touch faults and
                                                                                  * reallocation copies).  Sizes: h2w_ctx_footprint of an earlier run of the same shape */
uint64_t h2w_num_cells(const h2w_ctx *);                                        /* util/context_wrapper.rs:24-26 */
int      h2w_ctx_error(const h2w_ctx *);                                        /* sticky error flag of the context */
/* scoped cell counters: what the reference's #[count] proc-macro (macro/src/lib.rs:9-61) drives through
 * ContextWrapper::push_context / pop_context (util/context_wrapper.rs:28-34, util/context_tree.rs) */
int      h2w_push_context(h2w_ctx *, const char *name);
int      h2w_pop_context(h2w_ctx *);
size_t   h2w_context_dump(const h2w_ctx *, char *buf, size_t cap);              /* "a;b;c <inclusive cells>\n" lines; returns bytes needed */
/* field/native.rs:28-46 */
int h2w_load_constant(h2w_ctx *, const h2w_fr_t *c, h2w_assigned_t *out);
int h2w_load_zero(h2w_ctx *, h2w_assigned_t *out);
int h2w_load_constants(h2w_ctx *, const h2w_fr_t *c, size_t n, h2w_assigned_t *out);
int h2w_load_witness(h2w_ctx *, const h2w_fr_t *w, h2w_assigned_t *out);
/* field/native.rs:48-92 */
int h2w_add(h2w_ctx *, const h2w_assigned_t *a, const h2w_assigned_t *b, h2w_assigned_t *out);
int h2w_mul(h2w_ctx *, const h2w_assigned_t *a, const h2w_assigned_t *b, h2w_assigned_t *out);
int h2w_mul_add(h2w_ctx *, const h2w_assigned_t *a, const h2w_assigned_t *b, const h2w_assigned_t *c, h2w_assigned_t *out);
int h2w_select(h2w_ctx *, const h2w_assigned_t *a, const h2w_assigned_t *b, const h2w_assigned_t *sel, h2w_assigned_t *out);
/* field/native.rs:95-148 */
int h2w_select_from_idx(h2w_ctx *, const h2w_assigned_t *arr, size_t n, const h2w_assigned_t *idx, h2w_assigned_t *out);
int h2w_select_array_by_indicator(h2w_ctx *, const h2w_assigned_t *array2d /* [len][w] row-major */, size_t len, size_t w,
                                  const h2w_assigned_t *indicator, h2w_assigned_t *out /* [w] */);
int h2w_idx_to_indicator(h2w_ctx *, const h2w_assigned_t *idx, size_t len, h2w_assigned_t *out);
int h2w_num_to_bits(h2w_ctx *, const h2w_assigned_t *a, size_t range_bits, h2w_assigned_t *out);
int h2w_bits_to_num(h2w_ctx *, const h2w_assigned_t *bits, size_t n, h2w_assigned_t *out);
/* field/native.rs:150-193 */
int h2w_decompose_le(h2w_ctx *, const h2w_assigned_t *num, size_t limb_bits, size_t num_limbs, h2w_assigned_t *out);
int h2w_limbs_to_num(h2w_ctx *, const h2w_assigned_t *limbs, size_t n, size_t limb_bits, h2w_assigned_t *out);
int h2w_check_less_than_safe(h2w_ctx *, const h2w_assigned_t *a, uint64_t b);
int h2w_range_check(h2w_ctx *, const h2w_assigned_t *a, size_t range_bits);
int h2w_constrain_equal(h2w_ctx *, const h2w_assigned_t *a, const h2w_assigned_t *b); /* Context::constrain_equal: no cells */

/* ------------------------------------------------------------------ 2. fused Goldilocks level
 * field/goldilocks/base.rs: one call = the whole cell block of the op, hint included. */
int h2w_gl_load_constant(h2w_ctx *, uint64_t a, h2w_assigned_t *out);            /* base.rs:61-70   (1 cell)  */
int h2w_gl_load_witness(h2w_ctx *, uint64_t a, h2w_assigned_t *out);             /* base.rs:107-119 (28 @L=21) */
int h2w_gl_reduce(h2w_ctx *, const h2w_assigned_t *a, h2w_assigned_t *out);      /* base.rs:346-368 (61) */
int h2w_gl_add(h2w_ctx *, const h2w_assigned_t *a, const h2w_assigned_t *b, h2w_assigned_t *out);     /* :251-260 (65) */
int h2w_gl_sub(h2w_ctx *, const h2w_assigned_t *a, const h2w_assigned_t *b, h2w_assigned_t *out);     /* :274-283 (66) */
int h2w_gl_mul(h2w_ctx *, const h2w_assigned_t *a, const h2w_assigned_t *b, h2w_assigned_t *out);     /* :296-305 (65) */
int h2w_gl_mul_add(h2w_ctx *, const h2w_assigned_t *a, const h2w_assigned_t *b, const h2w_assigned_t *c, h2w_assigned_t *out); /* :319-329 */
int h2w_gl_div(h2w_ctx *, const h2w_assigned_t *a, const h2w_assigned_t *b, h2w_assigned_t *out);     /* :371-393 (93); error if b == 0 (:379) */
int h2w_gl_inv(h2w_ctx *, const h2w_assigned_t *a, h2w_assigned_t *out);                              /* :395-399 (94) */
int h2w_gl_mul_sub(h2w_ctx *, const h2w_assigned_t *a, const h2w_assigned_t *b, const h2w_assigned_t *c, h2w_assigned_t *out); /* :332-343 (70) */
int h2w_gl_neg(h2w_ctx *, const h2w_assigned_t *a, h2w_assigned_t *out);                              /* :234-238 (1 + 65) */
int h2w_gl_square(h2w_ctx *, const h2w_assigned_t *a, h2w_assigned_t *out);                           /* :401-404 (65) */
int h2w_gl_exp_power_of_2(h2w_ctx *, const h2w_assigned_t *base, size_t power_log, h2w_assigned_t *out); /* :433-445 (65 * power_log) */
/* The hint of GoldilocksQuadExtChip::inv (field/goldilocks/extension.rs:320-340: `a.value().inverse()` loaded as two witnesses, 56 cells at
 * lookup_bits 21): the inverse of a[0] + a[1] X is computed inside; the caller goes on with mul(a, out) and the assert_equal.  Error if a == 0 (:327). */
int h2w_gl_ext_inv_witness(h2w_ctx *, const h2w_assigned_t a[2], h2w_assigned_t out[2]);

/* ------------------------------------------------------------------ 2b. the reference's higher chips over the eager boundary
 * (host side above the C-ABI: csrc/chips.h + csrc/verifier.h instantiated on a backend that only calls the functions above).
 * Hash wires are 4 assigned values (hash_mode 0: PoseidonHashWire.elements) or 1 (hash_mode 1: PoseidonBN254HashWire.value;
 * outputs replicate it 4x). */
int h2w_chip_ext_op(h2w_ctx *, int op, const h2w_assigned_t a[2], const h2w_assigned_t b[2], const h2w_assigned_t c[2], h2w_assigned_t out[2]); /* field/goldilocks/extension.rs: 0 add 1 sub 2 mul 3 square 4 inv 5 div 6 mul_add 7 scalar_mul 8 scalar_div */
int h2w_chip_gl_exp_from_bits_const_base(h2w_ctx *, uint64_t base, const h2w_assigned_t *bits, size_t n, h2w_assigned_t *out);                 /* base.rs:407-430 */
int h2w_chip_gl_poseidon_permute(h2w_ctx *, const h2w_poseidon_consts_t *, const h2w_assigned_t in[12], h2w_assigned_t out[12]);             /* hash/poseidon/permutation.rs:270-284 */
int h2w_chip_bn_poseidon_permute(h2w_ctx *, const h2w_poseidon_consts_t *, const h2w_assigned_t in[4], h2w_assigned_t out[4]);               /* hash/poseidon_bn254/permutation.rs:190-203 */
int h2w_chip_hash_no_pad(h2w_ctx *, const h2w_poseidon_consts_t *, int hash_mode, const h2w_assigned_t *in, size_t n, h2w_assigned_t out[4]); /* HasherChip::hash_no_pad */
int h2w_chip_two_to_one(h2w_ctx *, const h2w_poseidon_consts_t *, int hash_mode, const h2w_assigned_t l[4], const h2w_assigned_t r[4], h2w_assigned_t out[4]); /* HasherChip::two_to_one */
int h2w_chip_merkle_verify(h2w_ctx *, const h2w_poseidon_consts_t *, int hash_mode, const h2w_assigned_t *leaf, size_t n_leaf,
                           const h2w_assigned_t *index_bits, size_t n_bits, const h2w_assigned_t *cap_index,
                           const h2w_assigned_t *cap, size_t n_cap, const h2w_assigned_t *siblings, size_t n_sib);                               /* merkle/mod.rs:57-78 */
int h2w_chip_verify_stark(h2w_ctx *, const h2w_shape_t *, const h2w_poseidon_consts_t *, const uint64_t *proof_words);                         /* stark/mod.rs:483-508 */

/* ------------------------------------------------------------------ 2d. record and replay: the operator API served at GPU speed
 * A context in trace mode records the tape of ONE run driven through nothing but the level-1 / level-2 calls above (for the reference: its unchanged
 * chips over the NativeChip shim); h2w_plan_from_trace lowers the tape to a device program, and h2w_fri_witness_batch replays it on any batch of proofs
 * of the same shape - bit-identical to running the calls on each of them.  What makes a run replayable:
 *   - control flow does not depend on values (true of the reference's gadgets: SURVEY 7);
 *   - a value that depends on the proof enters as a TAGGED proof word: h2w_trace_input(ctx, word, n) just before the h2w_load_witness /
 *     h2w_gl_load_witness / h2w_load_constant / h2w_gl_load_constant that loads it (n = 1: a Goldilocks element, 4: a BN254 hash);
 *   - hints are computed by the library: the two sites where the reference reads AssignedValue::value() for one (base.rs:382, extension.rs:327)
 *     are h2w_gl_div and h2w_gl_ext_inv_witness;  an untagged witness is an error at h2w_plan_from_trace.
 * parallel_scopes: names of h2w_push_context scopes whose instances do not depend on one another (the reference's #[count] scopes
 * "verify_query_round", fri/mod.rs:488-501, and "verify_proof_to_cap_with_cap_index", merkle/mod.rs:57-78): each instance becomes a lane of the
 * device program.  The library VERIFIES the claim on the tape (an instance may read what its enclosing scopes computed before it and nothing else;
 * nothing outside reads what it computes) and fails otherwise.  The plan supports h2w_plan_num_cells / _proof_words / _num_records /
 * _workspace_bytes / _status, h2w_fri_witness_batch and h2w_plan_free. */
int h2w_ctx_trace_begin(h2w_ctx *);                                   /* on a fresh context */
int h2w_trace_input(h2w_ctx *, uint64_t word, uint32_t n_words);      /* no-op on a context that is not tracing */
h2w_plan *h2w_plan_from_trace(h2w_ctx *, uint64_t proof_words, const char *const *parallel_scopes, size_t n_scopes, int device_id);

/* ------------------------------------------------------------------ advice hand-off (eager contexts) */
/* Expands all pending records on the GPU; *dev_ptr receives a device pointer to num_cells*32 bytes
 * owned by the context (valid until the next h2w_* call on it). */
int h2w_ctx_advice_device(h2w_ctx *, void **dev_ptr);
/* Copies cells [first, first+count) to host memory (expanding on the GPU first). */
int h2w_ctx_download(h2w_ctx *, uint64_t first, uint64_t count, h2w_fr_t *host_dst);

/* ------------------------------------------------------------------ 3. batched hot path */
/* Shape compile: replays the gadget once (host) to lay out every cell block of
 * load_proof_with_pis + verify_proof for this shape; uploads tables + constants to `device_id`. */
h2w_plan *h2w_plan_compile(const h2w_shape_t *, const h2w_poseidon_consts_t *, int device_id);
void      h2w_plan_free(h2w_plan *);
uint64_t  h2w_plan_num_cells(const h2w_plan *);         /* advice cells per proof */
uint64_t  h2w_plan_proof_words(const h2w_plan *);       /* u64 words per flat proof (layout: INTEGRATION.md) */
uint64_t  h2w_plan_num_records(const h2w_plan *);
/* Scratch bytes the batch call needs on the device for n_proofs (records, challenge blocks, traces). */
uint64_t  h2w_plan_workspace_bytes(const h2w_plan *, uint64_t n_proofs);
/* proofs_dev:  n_proofs * proof_words u64, device memory.
 * advice_dev:  n_proofs * num_cells * 32 bytes, device memory (canonical LE Fr).
 * workspace_dev: h2w_plan_workspace_bytes bytes, device memory.
 * stream: hipStream_t (NULL = default stream).  Asynchronous: returns after enqueueing. */
int h2w_fri_witness_batch(h2w_plan *, const uint64_t *proofs_dev, uint64_t n_proofs,
                          void *advice_dev, void *workspace_dev, void *stream);
/* Same, with the HBM-bound expansion kernel issued on `emit_stream` (ordered after the value strands by an event; `stream`
 * waits for it).  Lets a caller give the latency-bound value strands and the streaming kernel differently CU-masked streams. */
int h2w_fri_witness_batch2(h2w_plan *, const uint64_t *proofs_dev, uint64_t n_proofs,
                           void *advice_dev, void *workspace_dev, void *stream, void *emit_stream);
/* The expansion kernel alone: re-expands the block records a previous h2w_fri_witness_batch* call (same n_proofs) left in
 * workspace_dev into advice_dev (flat layout).  Cells the value kernels write directly are not touched.  For measurement. */
int h2w_fri_expand_records(h2w_plan *, uint64_t n_proofs, void *advice_dev, void *workspace_dev, void *stream);
/* ---- keygen-side bookkeeping of an eager context created with witness_gen_only == 0 (halo2-base Context when
 * witness_gen_only is false; semantics [R]): gate cells (selector on), range-lookup registrations (in order),
 * copy constraints (advice_equalities: pairs of cell offsets) and constant equalities (cell, constant). */
uint64_t h2w_ctx_num_gates(h2w_ctx *);           int h2w_ctx_gate_cells(h2w_ctx *, uint64_t *cells);
uint64_t h2w_ctx_num_lookups(h2w_ctx *);         int h2w_ctx_lookup_cells(h2w_ctx *, uint64_t *cells);
uint64_t h2w_ctx_num_equalities(h2w_ctx *);      int h2w_ctx_equalities(h2w_ctx *, uint64_t *pairs /* 2 per equality */);
uint64_t h2w_ctx_num_const_equalities(h2w_ctx *); int h2w_ctx_const_equalities(h2w_ctx *, uint64_t *cells, h2w_fr_t *values);

/* ---- the consumer-side format of the advice stream -------------------------------------------
 * The cell stream of a shape is static, so its keygen metadata is too.  Restates (halo2-lib `community-edition`, not in
 * /root/reference; semantics [R]): Context::selector (one bit per cell: a vertical gate starts there),
 * the range-lookup registrations (RangeChip::range_check -> cells_to_lookup, in registration order), the FlexGate
 * break points (gates/flex_gate/threads: assign_with_constraints) and the column assignment of the witness
 * (assign_witnesses) incl. the lookup-advice columns, and the copy manager's lists (advice_equalities, constant_equalities). */
uint64_t h2w_plan_num_gates(h2w_plan *);
uint64_t h2w_plan_num_lookups(h2w_plan *);
int h2w_plan_selectors(h2w_plan *, uint8_t *bitmap /* (num_cells + 7) / 8 bytes, bit i = cell i */);
int h2w_plan_lookup_cells(h2w_plan *, uint64_t *cells /* num_lookups */);
/* Copy constraints (pairs of cell offsets: Context::constrain_equal and the copies of Existing cells into gates, field/native.rs:185-193)
 * and constant equalities (cell, constant) of the plan's cell stream - what the halo2-base Context the reference hands to MockProver
 * carries (stark/mod.rs:483-518).  Built on first use by replaying the shape through an eager keygen context on the host. */
uint64_t h2w_plan_num_equalities(h2w_plan *);
uint64_t h2w_plan_num_const_equalities(h2w_plan *);
int h2w_plan_equalities(h2w_plan *, uint64_t *pairs /* 2 per equality */);
/* proof_words (host, h2w_plan_proof_words of them) may be NULL with PoseidonBN254 caps; with Goldilocks-Poseidon caps the reference loads
 * every hash wire of the proof as a CONSTANT (hash/poseidon/hash.rs:86-96), so those constants are the proof's own words. */
int h2w_plan_const_equalities(h2w_plan *, const uint64_t *proof_words, uint64_t *cells, h2w_fr_t *values);
/* break_points[c] = last used row of column c (the cell there is repeated at row 0 of column c + 1); max_rows = 2^k - unusable_rows.
 * out may be NULL to query *n_out. */
int h2w_break_points(const uint8_t *selectors, uint64_t n_cells, int k, int unusable_rows,
                     uint64_t *out, uint64_t cap, uint64_t *n_out);
/* flat advice (device) -> columns[n_proofs][n_bp + 1][2^k] (device, 32-byte canonical Fr, unassigned rows zero) */
int h2w_layout_columns(const void *advice_dev, uint64_t n_cells, uint64_t proof_stride_cells, uint64_t n_proofs,
                       const uint64_t *break_points, uint64_t n_bp, int k, void *columns_dev, void *stream);
/* The same result as h2w_fri_witness_batch + h2w_layout_columns without materialising the flat stream: every kernel writes its
 * cells straight to columns_dev[n_proofs][n_bp + 1][2^k] (a piecewise-constant address shift per column; a small fix-up kernel repeats
 * the boundary cells and zeroes the unused rows).  k >= 12.  Workspace as for h2w_fri_witness_batch. */
int h2w_fri_witness_batch_columns(h2w_plan *, const uint64_t *proofs_dev, uint64_t n_proofs, const uint64_t *break_points, uint64_t n_bp,
                                  int k, void *columns_dev, void *workspace_dev, void *stream);
/* lookup advice columns: out[n_proofs][*n_cols_out][2^k]; out_dev may be NULL to query *n_cols_out */
int h2w_layout_lookup_columns(h2w_plan *, const void *advice_dev, uint64_t proof_stride_cells, uint64_t n_proofs,
                              int k, int unusable_rows, void *out_dev, uint64_t *n_cols_out, void *stream);

/* Device-side constraint check of advice streams (restated MockProver gate + lookup checks): bad[0] = vertical
 * gates a[i] + a[i+1]*a[i+2] != a[i+3] over the selector-enabled cells, bad[1] = looked-up cells >= 2^lookup_bits, summed over
 * the n_proofs streams.  Synchronises the stream.  Covers every cell of full-size streams without the CPU oracle. */
int h2w_check_constraints(h2w_plan *, const void *advice_dev, uint64_t proof_stride_cells, uint64_t n_proofs,
                          uint64_t bad[2], void *stream);

/* (proof, query) sharding across the GPUs of a node (north star: "independent FRI queries and Merkle paths sharded across
 * the 8 GPUs"; the loop being dealt out is fri/mod.rs:488-501): rank r of `world` generates, at their global offsets in its own
 * advice_dev[n_proofs][num_cells], the prologue block of the proofs it owns (proof % world == r: witness load, challenger, PoW, reduced
 * openings) and the query blocks of the units (proof * num_queries + query) % world == r; the other blocks are not touched.  A rank
 * LAUNCHES only what it owns: its strand kernels run over its own units, its expansion kernel walks its own record ranges, so the
 * chain / glue / expansion time of a rank is 1 / world of the unsharded launch's.  The one exception is the values of the prologues
 * (one wavefront per proof, a fixed latency whatever the number of proofs): every rank needs every proof's challenges and computes
 * them itself - cheaper than a collective in the middle of the call; only the owner writes the block.  No collective on the data path:
 * every rank needs the proofs (one broadcast) and nothing else.  The ranks' blocks are disjoint and their union is the full stream. */
int h2w_fri_witness_batch_shard(h2w_plan *, const uint64_t *proofs_dev, uint64_t n_proofs, void *advice_dev, void *workspace_dev,
                                void *stream, int rank, int world);
/* The same blocks PACKED: shard_advice_dev holds only this rank's blocks, h2w_plan_shard_cells(plan, n_proofs, rank, world) cells, back to
 * back in (proof, block) order (prologue block first, then the owned query blocks by query number; every query block takes a slot of the
 * larger of the two query block sizes) - a rank's buffer is 1 / world of the stream, so the batch that fits a GPU grows with the world.
 * h2w_plan_shard_block: where a block lies - *local_cell in the packed buffer, *n_cells long, *global_cell its offset inside its proof's
 * flat stream; query < 0: the prologue block.  Returns 1 (and writes nothing) when the block belongs to another rank. */
uint64_t h2w_plan_shard_cells(const h2w_plan *, uint64_t n_proofs, int rank, int world);
/* Scratch bytes a sharded call (either form) of this rank needs for n_proofs: the per-proof pieces of h2w_plan_workspace_bytes plus the
 * PoseidonBN254 unit buffers of the rank's own (proof, query) units only (1 / world of them).  A workspace of h2w_plan_workspace_bytes(n_proofs)
 * bytes is always large enough. */
uint64_t h2w_plan_shard_workspace_bytes(const h2w_plan *, uint64_t n_proofs, int rank, int world);
int h2w_plan_shard_block(const h2w_plan *, int rank, int world, uint64_t proof, int query, uint64_t *local_cell, uint64_t *n_cells, uint64_t *global_cell);
int h2w_fri_witness_batch_shard_compact(h2w_plan *, const uint64_t *proofs_dev, uint64_t n_proofs, void *shard_advice_dev, void *workspace_dev,
                                        void *stream, int rank, int world);
/* The block structure of a proof's cell stream that the sharding above follows (static per shape; no device needed):
 * out[0] = cells of the prologue block [0, out[0]); out[1] = cells of query block 0 (it holds the Context's one cached load_zero
 * cell when that falls into a query); out[2] = cells of every later query block; out[3] = cells per proof
 * (= out[0] + out[1] + (num_queries - 1) * out[2]). */
int h2w_plan_strand_layout(const h2w_plan *, uint64_t out[4]);
/* The rest of the restated MockProver: copy constraints (bad[0] = pairs whose two cells differ) and constant equalities (bad[1])
 * over device advice streams.  The lists are static per shape: h2w_plan_equalities / h2w_plan_const_equalities (or an eager keygen
 * context of the same shape: h2w_ctx_equalities / h2w_ctx_const_equalities).  Synchronises the stream. */
int h2w_check_equalities(const void *advice_dev, uint64_t n_cells, uint64_t proof_stride_cells, uint64_t n_proofs,
                         const uint64_t *pairs, uint64_t n_pairs, const uint64_t *const_cells, const h2w_fr_t *const_values,
                         uint64_t n_const, uint64_t bad[2], void *stream);

/* Per-proof device status words.  0 = ok.  1 = GoldilocksChip::div by zero (reference asserts, base.rs:379), 2 = extension
 * inverse of zero (extension.rs:327), 3 = challenger input buffer overflow, 4 = the proof holds a word outside its field's
 * canonical range (a Goldilocks word >= p or a BN254 hash >= r: not representable by the reference's types; the cells are
 * still produced — identical to the reference's arithmetic for Goldilocks words, unreduced for the hash).  The first
 * condition met wins. */
int h2w_plan_status(h2w_plan *, const void *workspace_dev, uint64_t n_proofs, uint32_t *host_status, void *stream);
/* 32-byte checksum of n_cells cells (for streamed configurations, whose advice buffers are re-used before anything could read them
 * back): digest[j] = sum over cells i of limb_j(cell i) * ((((i + 1) * 0x9E3779B97F4A7C15) | 1) + 2 j)  mod 2^64 - position-dependent,
 * order-independent in its evaluation.  digest4_dev: 4 x u64, device.  One call per proof. */
int h2w_advice_digest(const void *advice_dev, uint64_t n_cells, uint64_t *digest4_dev, void *stream);
/* ---- Multi-GPU ingest: one process per GPU, no data-path collective.  The only exchange is ONE broadcast of the flat proof block
 * from the ingest rank (RCCL over xGMI); after it every rank runs h2w_fri_witness_batch on its own proofs or h2w_fri_witness_batch_shard on
 * its (proof, query) units.  Optionally the ranks gather each other's h2w_advice_digest words.  RCCL is loaded at the first call here
 * (dlopen), so the rest of the library works without it.  Replaces, for a caller without torch.distributed, what bench.py does with
 * dist.broadcast; the reference has no counterpart (single process, fri/mod.rs:488-501 is the loop being sharded).
 * id: H2W_COMM_ID_BYTES bytes made by h2w_comm_unique_id on one rank and handed to the others by any out-of-band channel. */
#define H2W_COMM_ID_BYTES 128
typedef struct h2w_comm h2w_comm;
int h2w_comm_unique_id(void *id128);
h2w_comm *h2w_comm_init(const void *id128, int rank, int world, int device_id);      /* collective: every rank calls it */
void h2w_comm_free(h2w_comm *);
int h2w_comm_rank(const h2w_comm *);
int h2w_comm_world(const h2w_comm *);
int h2w_comm_broadcast_proofs(h2w_comm *, uint64_t *proofs_dev, uint64_t n_words, int root, void *stream);      /* in place */
int h2w_comm_allgather_digests(h2w_comm *, const uint64_t *digest4_dev, uint64_t *all_dev /* [world][4] */, void *stream);
/* Output format: the stream is canonical little-endian Fr (what Fr::from_repr / to_repr use).  For a consumer that copies cells into
 * halo2curves' in-memory representation (Montgomery form, R = 2^256) this converts n_cells cells in place on the device. */
int h2w_advice_to_montgomery(void *cells_dev, uint64_t n_cells, void *stream);
/* Scheduling options of a plan.  H2W_OPT_FORK_CHAINS (default 1): with PoseidonBN254 Merkle caps the chain kernel of a batch call
 * runs on a library-owned side stream beside the query-glue and expansion kernels of the same call (they depend on the prologue
 * only); the caller's stream still completes when the whole advice is written.  0: every kernel on the caller's stream. */
#define H2W_OPT_FORK_CHAINS 1
/* H2W_OPT_SERIAL_EXPAND (default 1): the expansion kernel of a batch call
 * waits for the expansion kernel of the plan's previous batch call, whatever stream that was issued on, so that the latency-bound
 * strands of the other calls in flight find free CUs; the cells written are the same either way. */
#define H2W_OPT_SERIAL_EXPAND 2
/* H2W_OPT_CHAIN_PASSES (PoseidonBN254 caps): how the Merkle paths (merkle/mod.rs:57-78) are generated.
 * 1: one quad per path walks it and emits as it goes - the least arithmetic per cell, serial in the path's depth (18 permutations at 2^20
 *    rows, ~7 ms): right when many paths are in flight (large batches, several launches pipelined), which hide that latency.
 * 2: a values pass walks every path serially, then one quad per permutation emits its cells - all levels of all paths side by side.  The
 *    emission is a streaming kernel whose time is proportional to the work (1 / world of it in a sharded launch), at the price of evaluating
 *    every permutation twice: right for small or sharded launches, which the depth of a path would otherwise bound.
 * 0 (default): 2 for launches of at most 512 (proof, query) units of this rank, else 1.  The cells are the same either way. */
#define H2W_OPT_CHAIN_PASSES 3
/* H2W_OPT_VALUES_FORM (PoseidonBN254 caps, two-pass paths): how the values pass walks a path.  1: four lanes per path (a Montgomery product on one
 * lane); 2: one wavefront per path, a product spread over the 16 lanes of a row, one 29-bit limb per lane (a path in less than half the time, six
 * times the instructions per path: right while there is a SIMD for nearly every path).  0 (default): 2 for launches of at most 784 paths of this
 * rank (4 proofs of 2^20 rows x 28 queries), else 1: alone on the chip form 2 wins up to ~2,500 paths (profiles/r04_values_forms.jsonl), with other launches
 * in flight its instructions are not free - a caller that runs one launch at a time sets 2.  The cells are the same either way. */
#define H2W_OPT_VALUES_FORM 4
int h2w_plan_configure(h2w_plan *, int option, int value);
/* Kernel timing of a batch call, in ms, from HIP events the library records on the streams it launches on:
 * ms[0] = prologue strands (values; with PoseidonBN254 caps also the records of their permutations), ms[1] = query glue strands (with
 * Goldilocks-Poseidon caps also the Merkle strands' values and the records of every listed permutation), ms[2] = PoseidonBN254 Merkle
 * chains, values + emission (0 with Goldilocks-Poseidon caps), ms[3] = expansion kernel, ms[4] = whole call.
 * `back` = how many batch calls before the last one (a ring of the last 64 is kept).  Blocks until that batch finished. */
int h2w_plan_timing(h2w_plan *, uint64_t back, float ms[5]);
/* The same per kernel: ms[0] k_prologue_values (alone: the witness-load kernel k_prologue_load runs BEHIND it, beside the Merkle chains - nothing waits for its
 * cells - and is part of ms[6] only), ms[1] k_glp_emit (records of the listed Goldilocks-Poseidon permutations), ms[2] k_strands
 * (+ k_merkle_gl_values), ms[3] k_merkle_bn_values (one pass: k_merkle_bn_fused), ms[4] k_merkle_bn_emit, ms[5] expansion kernel, ms[6] whole call,
 * ms[7] = the H2W_OPT_CHAIN_PASSES the call ran with (PoseidonBN254 caps; else 0). */
int h2w_plan_timing_ex(h2w_plan *, uint64_t back, float ms[8]);
int h2w_plan_last_timing(h2w_plan *, float ms[5]);
/* Elapsed ms from event `which_a` of the batch call `back_a` calls before the last one to event `which_b` of the call `back_b` before
 * the last one (negative when b came first): how the kernels of different calls lie against each other on the device. */
#define H2W_EV_CALL_START 0
#define H2W_EV_PROLOGUE_END 1
#define H2W_EV_GLUE_START 2
#define H2W_EV_GLUE_END 3
#define H2W_EV_CHAINS_START 4
#define H2W_EV_CHAINS_END 5
#define H2W_EV_EXPAND_START 6
#define H2W_EV_EXPAND_END 7
#define H2W_EV_CALL_END 8
#define H2W_EV_COUNT 9
int h2w_plan_event_gap(h2w_plan *, uint64_t back_a, int which_a, uint64_t back_b, int which_b, float *ms);
/* Advice cells per proof written by the expansion kernel (the rest are written directly by the value kernels). */
uint64_t h2w_plan_num_record_cells(const h2w_plan *);
/* Advice cells per proof of the MerkleTreeChip::verify_proof_to_cap_with_cap_index calls (merkle/mod.rs:57-78): with PoseidonBN254
 * caps these are the cells the chain kernel writes itself (its permutations' 4,032 cells each, the selects, the cap lookup). */
uint64_t h2w_plan_num_chain_cells(const h2w_plan *);

/* ------------------------------------------------------------------ 2c. batched chip ops on the device
 * n independent instances of ONE GoldilocksChip / GoldilocksQuadExtChip operation, each in a fresh Context: the operands are loaded
 * as witnesses (GoldilocksChip::load_witness, 28 cells each at lookup_bits 21; extension elements as two), then the op - the shape
 * of the reference's chip tests (field/goldilocks/base.rs:476-495, extension.rs:473-493).  One lane per instance computes the
 * values, the expansion kernel writes advice_dev[n][num_cells].  status_dev[i] = 0, or 1 / 2 where the reference panics
 * (GoldilocksChip::div by zero, base.rs:379; extension inverse of zero, extension.rs:327: the cells are then those of the op on
 * the substituted operand 1, as in the batched verifier path).  operands_dev: [n][num_operands] canonical Goldilocks words. */
#define H2W_OP_GL_ADD 0      /* a, b       base.rs:251-260 */
#define H2W_OP_GL_SUB 1      /* a, b       base.rs:274-283 */
#define H2W_OP_GL_MUL 2      /* a, b       base.rs:296-305 */
#define H2W_OP_GL_MUL_ADD 3  /* a, b, c    base.rs:319-329 */
#define H2W_OP_GL_DIV 4      /* a, b       base.rs:371-393 */
#define H2W_OP_GL_INV 5      /* a          base.rs:395-399 */
#define H2W_OP_EXT_MUL 6     /* a0 a1 b0 b1  extension.rs:211-234 */
#define H2W_OP_EXT_INV 7     /* a0 a1        extension.rs:320-340 */
#define H2W_OP_EXT_DIV 8     /* a0 a1 b0 b1  extension.rs:237-246 */
typedef struct h2w_chipbatch h2w_chipbatch;
h2w_chipbatch *h2w_chipbatch_new(int op, int lookup_bits, int device_id);
void           h2w_chipbatch_free(h2w_chipbatch *);
uint64_t       h2w_chipbatch_num_operands(const h2w_chipbatch *);   /* u64 words per instance */
uint64_t       h2w_chipbatch_num_cells(const h2w_chipbatch *);      /* advice cells per instance */
int            h2w_chipbatch_run(h2w_chipbatch *, const uint64_t *operands_dev, uint64_t n, void *advice_dev, uint32_t *status_dev, void *stream);

/* ------------------------------------------------------------------ the step BEFORE the path
 * Synthetic VALID FRI instances generated on the GPU (SURVEY 8(d) variant (A)): low-degree extension (Goldilocks NTT on the coset
 * 7<w>), Merkle commitments with the selected hash, the starky / plonky2 Fiat-Shamir transcript (stark/mod.rs:167-222 order, as
 * ChallengerChip replays it: challenger/mod.rs:168-222), openings at zeta and g*zeta, the alpha-batched quotient, the FRI commit
 * phase (arity 2^arity_bits coefficient folding), proof-of-work grinding and the query openings - written in the flat proof layout
 * h2w_fri_witness_batch consumes, so proofs never visit the host.  The reference obtains such proofs from starky's
 * prover (stark/mod.rs:405-426, test_util/fibonacci_stark.rs); as there, the committed polynomials are inputs: the STARK constraint
 * identity is not part of the verifier gadget (stark/mod.rs:243-321 is commented out), any polynomials of degree < 2^degree_bits do.
 * The serial sponge runs on the host between device phases (a few dozen permutations per proof). */
typedef struct h2w_prover h2w_prover;
h2w_prover *h2w_prover_new(const h2w_shape_t *, const h2w_poseidon_consts_t *, int device_id);
void        h2w_prover_free(h2w_prover *);
uint64_t    h2w_prover_num_polys(const h2w_prover *);      /* n_cols + n_perm_z + n_quotient */
uint64_t    h2w_prover_proof_words(const h2w_prover *);    /* = h2w_plan_proof_words of the same shape */
/* coeffs_dev: [num_polys][2^degree_bits] canonical Goldilocks coefficients (trace columns, permutation Zs, quotient polynomials), device.
 * public_inputs: n_pis words, host.  proof_dev: proof_words u64, device.  Synchronises `stream` several times (transcript). */
int h2w_prove_fri(h2w_prover *, const uint64_t *coeffs_dev, const uint64_t *public_inputs, uint64_t *proof_dev, void *stream);
/* n_proofs (<= 2048) instances in lockstep: coeffs_dev [n_proofs][num_polys][2^degree_bits], public_inputs [n_proofs][n_pis] (host),
 * proofs_dev [n_proofs][proof_words].  Every kernel covers the whole batch and the host transcript serves all proofs at each of its
 * round trips, so the latency-bound pieces (upper tree levels, small fold steps) are paid once per batch.  Device scratch is
 * (re)allocated inside the handle for the largest batch seen (cfg 3: about 0.65 GB per proof). */
int h2w_prove_fri_batch(h2w_prover *, const uint64_t *coeffs_dev, const uint64_t *public_inputs, uint64_t *proofs_dev, uint64_t n_proofs, void *stream);
/* ms[0] = LDE (NTTs), ms[1] = Merkle commitments of the oracles, ms[2] = openings + batched quotient, ms[3] = FRI commit phase,
 * ms[4] = proof of work, ms[5] = query openings, ms[6] = whole call (wall clock incl. the host transcript) of the last h2w_prove_fri / h2w_prove_fri_batch */
int h2w_prover_timing(h2w_prover *, float ms[7]);

#ifdef __cplusplus
}
#endif
#endif
