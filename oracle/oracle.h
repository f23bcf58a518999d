/*
 * oracle.h — CPU restatement ("oracle") of the halo2 FRI-verifier gadget's witness generation.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (halo2-plonky2-verifier_amd/, include/) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker / CPU baseline.
 *
 * PARITY STATUS: "parity unpinned" at the third-party boundary.  The reference
 * (/root/reference, Rust nightly) cannot be compiled here (no cargo/rustc, un-vendored git deps:
 * halo2-lib `community-edition`, plonky2/starky, succinctx, all unpinned: verifier/Cargo.toml:14-20)
 * and its tests hold no literal vectors.  What IS pinned: the exact per-call-stack advice-cell counts of
 * verifier/profile/{gl,bn254}.svg (tests/golden/svg_frames_*.json), which this oracle reproduces
 * frame by frame, plus mathematical identities, a restated MockProver (constraint checker), and self-consistency on VALID
 * FRI instances: prover.inc (a native prover following plonky2's prover conventions) produces proofs on which the restated
 * gadget satisfies every constraint including the chip-level asserts (tests/test_oracle_valid_proof.py).
 *
 * Every function cites the reference file:line it follows.
 */
#ifndef H2W_ORACLE_H
#define H2W_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[4]; } ofr_t; /* BN254 Fr, canonical, little-endian 64-bit limbs */

/* Shape of the STARK / FRI instance (starky StarkConfig + Stark trait facts). */
typedef struct {
    int32_t degree_bits;      /* log2(trace rows) */
    int32_t rate_bits;        /* FriConfig.rate_bits */
    int32_t cap_height;       /* FriConfig.cap_height */
    int32_t num_queries;      /* FriConfig.num_query_rounds */
    int32_t pow_bits;         /* FriConfig.proof_of_work_bits */
    int32_t num_challenges;   /* StarkConfig.num_challenges */
    int32_t arity_bits;       /* ConstantArityBits(arity_bits, final_poly_bits) */
    int32_t final_poly_bits;
    int32_t n_cols;           /* S::COLUMNS (Fibonacci: 4) */
    int32_t n_perm_z;         /* num_permutation_batches (Fibonacci: 2), 0 = no permutation args */
    int32_t n_quotient;       /* quotient_degree_factor * num_challenges (Fibonacci: 2) */
    int32_t n_pis;            /* S::PUBLIC_INPUTS (3) */
    int32_t perm_batch_size;  /* stark.permutation_batch_size() (1) */
    int32_t hash_mode;        /* 0 = Goldilocks-Poseidon Merkle, 1 = PoseidonBN254 Merkle */
    int32_t lookup_bits;      /* halo2-base RangeChip lookup_bits (= k-1) */
    int32_t witness_load_range_check; /* 1 = current source (28 cells / GL element, witness/mod.rs:49-51);
                                         0 = SVG-era loader (1 cell / element), used only to match total_samples */
} oshape_t;

/* Poseidon constants are INPUTS (the Rust caller has them from plonky2 / plonky2x; they are not
 * on this box).  Indexing exactly as hash/poseidon/permutation.rs and hash/poseidon_bn254/permutation.rs. */
typedef struct {
    uint64_t all_round_constants[360];
    uint64_t mds_circ[12];
    uint64_t mds_diag[12];
    uint64_t fast_partial_first_round_constant[12];
    uint64_t fast_partial_round_constants[22];
    uint64_t fast_partial_round_initial_matrix[11][11];
    uint64_t fast_partial_round_w_hats[22][11];
    uint64_t fast_partial_round_vs[22][11];
    ofr_t bn_c[88];      /* C_CONSTANTS */
    ofr_t bn_s[392];     /* S_CONSTANTS */
    ofr_t bn_m[4][4];    /* M_MATRIX */
    ofr_t bn_p[4][4];    /* P_MATRIX */
} oconsts_t;

typedef struct octx octx_t;

/* --- context --- */
octx_t *orc_ctx_new(int lookup_bits, int witness_gen_only, int track_scopes);
/* streams too long for the host: a ring of the last cells + the checksum of include/h2w.h's h2w_advice_digest (orc_advice is then meaningless) */
octx_t *orc_ctx_new_streaming(int lookup_bits);
void orc_digest(const octx_t *, uint64_t out[4]);
void    orc_ctx_free(octx_t *);
uint64_t orc_num_cells(const octx_t *);
void orc_ctx_reserve(octx_t *, uint64_t ncells);   /* capacity hint (avoids realloc copies) */
const ofr_t *orc_advice(const octx_t *);
const char *orc_error(const octx_t *);
/* restated MockProver: returns 0 if all gates / internal equalities / constant equalities / lookups hold.
 * semantic_failed receives the number of chip-level assert_equal constraints that do not hold
 * (expected non-zero for random, i.e. invalid, synthetic proofs). */
int orc_mock_prover(const octx_t *, uint64_t *gates, uint64_t *equalities, uint64_t *lookups, uint64_t *semantic_failed);
/* scope tree dump: writes "path cells\n" lines (inclusive counts) into buf; returns bytes needed */
size_t orc_scope_dump(const octx_t *, char *buf, size_t cap);
/* which #[count] call stack appends a given cell (set before the run; the context must track scopes) */
void orc_watch_cell(octx_t *, uint64_t cell);
const char *orc_watch_path(const octx_t *, uint64_t *offset_in_call);

/* --- keygen metadata + FlexGate column layout (needs witness_gen_only = 0); halo2-lib semantics [R], see oracle.c --- */
uint64_t orc_num_gates(const octx_t *);
void orc_selector_bitmap(const octx_t *, uint8_t *out /* (num_cells + 7) / 8 */);
uint64_t orc_num_equalities(const octx_t *);
void orc_equalities(const octx_t *, uint64_t *pairs /* 2 per equality: copy constraints incl. chip-level assert_equal */);
uint64_t orc_num_const_equalities(const octx_t *);
void orc_const_equalities(const octx_t *, uint64_t *cells, ofr_t *values);
uint64_t orc_num_lookups(const octx_t *);
void orc_lookup_cells(const octx_t *, uint64_t *out);
uint64_t orc_break_points(const octx_t *, int k, int unusable_rows, uint64_t *out, uint64_t cap);
void orc_layout_columns(const octx_t *, const uint64_t *bp, uint64_t n_bp, int k, ofr_t *out);
uint64_t orc_layout_lookup_columns(const octx_t *, int k, int unusable_rows, ofr_t *out);

/* --- synthetic inputs (splitmix64-seeded) --- */
void orc_synth_consts(oconsts_t *out, uint64_t seed);
size_t orc_proof_words(const oshape_t *);                      /* number of u64 words in the flat proof */
void orc_synth_proof(const oshape_t *, uint64_t seed, uint64_t *words);

/* a VALID FRI instance of the shape (native value-domain prover following plonky2's prover conventions, prover.inc);
 * on it the restated MockProver must report semantic_failed == 0.  Small shapes only (lde_bits <= 20). */
int orc_prove_fri(const oshape_t *, const oconsts_t *, uint64_t seed, uint64_t *words);
size_t orc_prove_fri_inputs(const oshape_t *, uint64_t seed, uint64_t *coefs, uint64_t *pis);   /* returns the number of coefficient words */
int orc_prove_fri_coef(const oshape_t *, const oconsts_t *, const uint64_t *coefs, const uint64_t *pis, uint64_t *words);
/* native (no-cell) twins of the hash gadgets, for cross-checks */
void orc_nv_gl_permute(const oconsts_t *, uint64_t st[12]);
void orc_nv_bn_permute(const oconsts_t *, ofr_t st[4]);
void orc_nv_hash_or_noop(const oconsts_t *, int hash_mode, const uint64_t *in, int n, uint64_t out[4]);

/* --- the path: load_proof_with_pis + StarkChip::verify_proof (stark/mod.rs:483-508) --- */
int orc_verify_stark(octx_t *, const oshape_t *, const oconsts_t *, const uint64_t *proof_words);

/* --- unit-level drivers mirroring the reference's unit tests (used for op-level parity) --- */
/* each appends cells to ctx; handles are cell indices (or -1) with the value kept inside ctx */
typedef struct { ofr_t v; int64_t cell; } oav_t;
oav_t orc_load_witness(octx_t *, const ofr_t *v);
oav_t orc_load_constant(octx_t *, const ofr_t *v);
oav_t orc_load_zero(octx_t *);
oav_t orc_add(octx_t *, oav_t a, oav_t b);
oav_t orc_mul(octx_t *, oav_t a, oav_t b);
oav_t orc_mul_add(octx_t *, oav_t a, oav_t b, oav_t c);
oav_t orc_select(octx_t *, oav_t a, oav_t b, oav_t sel);
oav_t orc_select_from_idx(octx_t *, const oav_t *arr, int n, oav_t idx);
void  orc_idx_to_indicator(octx_t *, oav_t idx, int len, oav_t *out);
void  orc_select_array_by_indicator(octx_t *, const oav_t *arr2d, int len, int w, const oav_t *ind, oav_t *out);
void  orc_num_to_bits(octx_t *, oav_t a, int bits, oav_t *out);
oav_t orc_bits_to_num(octx_t *, const oav_t *bits, int n);
void  orc_decompose_le(octx_t *, oav_t a, int limb_bits, int n, oav_t *out);
oav_t orc_limbs_to_num(octx_t *, const oav_t *limbs, int n, int limb_bits);
void  orc_check_less_than_safe(octx_t *, oav_t a, uint64_t b);
void  orc_range_check(octx_t *, oav_t a, int bits);
void  orc_constrain_equal(octx_t *, oav_t a, oav_t b);

oav_t orc_gl_load_witness(octx_t *, uint64_t a);
oav_t orc_gl_load_constant(octx_t *, uint64_t a);
oav_t orc_gl_reduce(octx_t *, oav_t a);
oav_t orc_gl_add(octx_t *, oav_t a, oav_t b);
oav_t orc_gl_sub(octx_t *, oav_t a, oav_t b);
oav_t orc_gl_mul(octx_t *, oav_t a, oav_t b);
oav_t orc_gl_mul_add(octx_t *, oav_t a, oav_t b, oav_t c);
oav_t orc_gl_mul_sub(octx_t *, oav_t a, oav_t b, oav_t c);
oav_t orc_gl_div(octx_t *, oav_t a, oav_t b);
oav_t orc_gl_inv(octx_t *, oav_t a);
oav_t orc_gl_exp_from_bits_const_base(octx_t *, uint64_t base, const oav_t *bits, int n);
oav_t orc_gl_exp_power_of_2(octx_t *, oav_t base, int power_log);
void  orc_ext_mul(octx_t *, const oav_t a[2], const oav_t b[2], oav_t out[2]);
void  orc_ext_inv(octx_t *, const oav_t a[2], oav_t out[2]);
void  orc_ext_div(octx_t *, const oav_t a[2], const oav_t b[2], oav_t out[2]);
void  orc_gl_poseidon_permute(octx_t *, const oconsts_t *, const oav_t in[12], oav_t out[12]);
void  orc_bn_poseidon_permute(octx_t *, const oconsts_t *, const oav_t in[4], oav_t out[4]);
/* hash_no_pad / two_to_one for both hashers; hash wires are 4 GL wires (mode 0) or 1 Fr wire in out[0] (mode 1) */
void  orc_hash_no_pad(octx_t *, const oconsts_t *, int hash_mode, const oav_t *in, int n, oav_t out[4]);
void  orc_two_to_one(octx_t *, const oconsts_t *, int hash_mode, const oav_t l[4], const oav_t r[4], oav_t out[4]);
/* merkle/mod.rs:57-78; leaf GL wires, index bits (bool wires), cap (n_cap hash wires), siblings */
void  orc_merkle_verify(octx_t *, const oconsts_t *, int hash_mode, const oav_t *leaf, int n_leaf,
                        const oav_t *bits, int n_bits, oav_t cap_index,
                        const oav_t *cap, int n_cap, const oav_t *siblings, int n_sib);

/* native (value-domain) helpers exposed for tests */
uint64_t orc_glf_mul(uint64_t a, uint64_t b);
uint64_t orc_glf_inv(uint64_t a);
uint64_t orc_glf_exp(uint64_t a, uint64_t e);
uint64_t orc_glf_primitive_root_of_unity(int bits);
void orc_fr_mul(const ofr_t *a, const ofr_t *b, ofr_t *out);
void orc_fr_inv(const ofr_t *a, ofr_t *out);
void orc_fr_modulus(ofr_t *out);

#ifdef __cplusplus
}
#endif
#endif
