// chips.h — the verifier gadget stack (layers L2-L5 of the reference), written once and instantiated over a
// backend `B` that stands in for NativeChip / halo2-base:
//
//   * ValBackend<DevSink>  (device)  — computes Goldilocks / Fr values natively and emits block records + direct cells
//   * ValBackend<PlanSink> (host)    — same code with a counting sink: the shape compiler (cell layout, record metas)
//   * AbiBackend           (host)    — drives the eager C-ABI of include/h2w.h (the drop-in boundary itself)
//
// Each class mirrors one reference chip: names, argument meaning and call order follow the Rust source so the
// advice stream is cell-for-cell the reference's.  File:line citations are to /root/reference/verifier/src.
#pragma once
#include "field.h"

namespace h2w {

constexpr int SPONGE_WIDTH = 12, SPONGE_RATE = 8, HALF_N_FULL_ROUNDS = 4, N_PARTIAL_ROUNDS = 22, NUM_HASH_OUT_ELTS = 4;
constexpr int BN_WIDTH = 4, BN_RATE = 3, BN_FULL_ROUNDS = 8, BN_PARTIAL_ROUNDS = 56;
constexpr int MAX_STEPS = 8, MAX_ARITY = 16, MAX_CAP = 64, MAX_BATCH_POLYS = 16, MAX_FINAL_POLY = 128;      // MAX_CAP: cap_height <= 6 (sizes only the host-side / Goldilocks-caps indicator arrays)
// Constants of the FRI gadgets that depend on the shape only (fri/mod.rs:222-322 recomputes them at every call: the two-adic subgroup of each
// arity, its barycentric weights - an inversion each -, the inverse generator, the generator of the LDE domain).  Built once per plan on the host
// (fri_tab_build); a backend without one (fri_tab() == nullptr: the eager ABI driver, the shape compiler's replay) computes them as the reference does.
constexpr int FRI_TAB_BITS = 4;
static_assert((1 << FRI_TAB_BITS) == MAX_ARITY, "FriTab is indexed by arity_bits <= log2 MAX_ARITY");
struct FriTab { uint64_t dom[FRI_TAB_BITS + 1][MAX_ARITY], bw[FRI_TAB_BITS + 1][MAX_ARITY], g_inv[FRI_TAB_BITS + 1], root_lde; int64_t lde_bits; };
inline void fri_tab_build(FriTab &t, int lde_bits) {
    for (int ab = 0; ab <= FRI_TAB_BITS; ab++) {
        const int n = 1 << ab; const uint64_t g = gl_primitive_root_of_unity(ab);
        uint64_t *dom = t.dom[ab]; dom[0] = 1; for (int i = 1; i < n; i++) dom[i] = gl_mul(dom[i - 1], g);
        for (int i = 0; i < MAX_ARITY; i++) { if (i >= n) { dom[i] = 0; t.bw[ab][i] = 0; continue; } uint64_t pr = 1; for (int j = 0; j < n; j++) if (j != i) pr = gl_mul(pr, gl_sub(dom[i], dom[j])); t.bw[ab][i] = gl_inv(pr); }
        t.g_inv[ab] = gl_exp(g, (uint64_t)n - 1);
    }
    t.root_lde = gl_primitive_root_of_unity(lde_bits); t.lde_bits = lde_bits;
}
enum { PRE_NONE = 0, PRE_A = 1, PRE_B = 2 };

// A small array that is indexed with run-time indices WITHOUT being addressed: an element is picked by a chain of selects over constant indices.
// One access through a run-time offset keeps a WHOLE local object in scratch memory (the compiler splits an object into registers only when every
// access to it has a constant offset) - and with the Verifier in memory so are the backend and the sink it points to: every counter the strand kernels
// bump was a scratch round trip behind their record stores (round 4: 2.4 k cycles around every permutation of the prologue).
template <class T, int N> struct SelArr {
    T v[N];
    HF T operator[](int i) const {
        T r = v[0];
#pragma unroll
        for (int j = 1; j < N; j++) { const T e = v[j]; r = i == j ? e : r; }      // (every element READ, then picked: a load under the condition is merged with its neighbours into one load at a selected offset - a run-time offset again)
        return r;
    }
    HF void set(int i, T x) {
#pragma unroll
        for (int j = 0; j < N; j++) v[j] = i == j ? x : v[j];
    }
};
// shape-derived quantities (plonky2 FriParams)
struct Derived {
    int lde_bits, n_steps; SelArr<int, MAX_STEPS> arity; int final_poly_len, cap_size, n_oracles; SelArr<int, 3> oracle_polys;
};
HF Derived derive_shape(const h2w_shape_t &s) {
    Derived d; d.lde_bits = s.degree_bits + s.rate_bits; d.n_steps = 0;
    int db = s.degree_bits;
#pragma unroll
    for (int i = 0; i < MAX_STEPS; i++) {      // (once the condition fails it stays failed: db no longer changes)
        const bool more = db > s.final_poly_bits && db + s.rate_bits - s.arity_bits >= s.cap_height;
        d.arity.v[i] = more ? s.arity_bits : 0;
        if (more) { d.n_steps = i + 1; db -= s.arity_bits; }
    }
    d.final_poly_len = 1 << db; d.cap_size = 1 << s.cap_height;
    d.oracle_polys.v[0] = s.n_cols; d.oracle_polys.v[1] = s.n_perm_z > 0 ? s.n_perm_z : s.n_quotient; d.oracle_polys.v[2] = s.n_perm_z > 0 ? s.n_quotient : 0;
    d.n_oracles = s.n_perm_z > 0 ? 3 : 2;
    return d;
}
// Flat proof layout = WitnessChip load order (witness/mod.rs:236-294); every hash is 4 u64 words
// (GL mode: 4 field elements; BN254 mode: canonical LE limbs of the Fr hash, poseidon_bn254/hash.rs:19-21).
struct ProofLayout {
    uint64_t trace_cap, quotient_cap, openings, perm_cap, pow_witness, final_poly, commit_caps, queries, query_words, pis, total;
    SelArr<uint64_t, 3> init_off;            // within a query: start of oracle o (evals then siblings)
    SelArr<uint64_t, MAX_STEPS> step_off;    // within a query: start of fold step i (evals then siblings)
    SelArr<int, MAX_STEPS> step_sibs; int init_sibs;
};
HF ProofLayout proof_layout(const h2w_shape_t &s, const Derived &d) {
    ProofLayout L; uint64_t w = 0;
    L.trace_cap = w; w += (uint64_t)d.cap_size * 4;
    L.quotient_cap = w; w += (uint64_t)d.cap_size * 4;
    L.openings = w; w += 2ull * (2 * s.n_cols + 2 * s.n_perm_z + s.n_quotient);
    L.perm_cap = w; if (s.n_perm_z > 0) w += (uint64_t)d.cap_size * 4;
    L.pow_witness = w; w += 1;
    L.final_poly = w; w += 2ull * d.final_poly_len;
    L.commit_caps = w; w += (uint64_t)d.n_steps * d.cap_size * 4;
    L.queries = w;
    uint64_t q = 0; L.init_sibs = d.lde_bits - s.cap_height;
#pragma unroll
    for (int o = 0; o < 3; o++) { L.init_off.v[o] = q; if (o < d.n_oracles) q += (uint64_t)d.oracle_polys.v[o] + (uint64_t)L.init_sibs * 4; }
    int bits = d.lde_bits;
#pragma unroll
    for (int i = 0; i < MAX_STEPS; i++) {      // (constant indices: see SelArr)
        const bool on = i < d.n_steps;
        if (on) bits -= d.arity.v[i];
        L.step_off.v[i] = q; L.step_sibs.v[i] = on ? bits - s.cap_height : 0;
        if (on) q += (2ull << d.arity.v[i]) + (uint64_t)L.step_sibs.v[i] * 4;
    }
    L.query_words = q; w += q * (uint64_t)s.num_queries;
    L.pis = w; w += (uint64_t)s.n_pis; L.total = w;
    return L;
}

// =========================================================================== GoldilocksChip (field/goldilocks/base.rs)
template <class B> struct GoldilocksChip {
    typedef typename B::Gl Gl; typedef typename B::Bool Bool; typedef typename B::Big Big;
    B &be;
    HF explicit GoldilocksChip(B &b) : be(b) {}
    HF Gl load_constant(uint64_t a) { return be.gl_const(a); }                         // :61-70
    HF Gl load_zero() { return load_constant(0); }                                     // :72-75
    HF Gl load_one() { return load_constant(1); }                                      // :77-80
    HF Gl load_neg_one() { return load_constant(GL_NEG_ONE); }                         // :82-85
    HF void load_constant_array(const uint64_t *a, int n, Gl *out) { for (int i = 0; i < n; i++) out[i] = load_constant(a[i]); } // :87-94
    HF void load_zero_array(int n, Gl *out) { be.gl_const_run(0, n, out); }            // load_constant_array(&[ZERO; N])
    HF Gl load_witness(uint64_t a) { return be.gl_witness(a); }                        // :107-119 (range check included)
    HF Gl select(Gl a, Gl b, Bool sel) { return be.select(a, b, sel); }                // :138-148
    HF void select_array(const Gl *a, const Gl *b, int n, Bool sel, Gl *out) { for (int i = 0; i < n; i++) out[i] = be.select(a[i], b[i], sel); } // :150-165
    HF Gl select_from_idx(const Gl *arr, int n, Gl idx) {                              // :168-180
        Bool ind[MAX_CAP]; be.idx_to_indicator(idx, n, ind); return be.select_by_indicator(arr, 1, ind, n);
    }
    HF void select_array_from_idx(const Gl *arr /*[len][w]*/, int len, int w, Gl idx, Gl *out) { // :182-207
        Bool ind[MAX_CAP]; be.idx_to_indicator(idx, len, ind);
        for (int j = 0; j < w; j++) out[j] = be.select_by_indicator(arr + j, w, ind, len);
    }
    HF void num_to_bits(Gl a, int range_bits, Bool *out) { be.num_to_bits(a, range_bits, out); }  // :209-220
    HF Gl bits_to_num(const Bool *bits, int n) { return be.bits_to_num(bits, n); }                // :222-232
    HF Gl neg(Gl a) { Gl m1 = load_neg_one(); return mul(a, m1); }                                 // :234-238
    HF Big add_no_reduce(Gl a, Gl b) { return be.gl_gate(PRE_NONE, b, be.gl_lit(1), a); }          // :240-249  [a, b, 1, a+b]
    HF Gl add(Gl a, Gl b) { return be.glop(PRE_NONE, b, be.gl_lit(1), a); }                         // :251-260
    HF Big sub_no_reduce(Gl a, Gl b) { return be.gl_gate(PRE_B, b, be.gl_lit(GL_NEG_ONE), a); }    // :262-272  [p-1][a, b, p-1, b(p-1)+a]
    HF Gl sub(Gl a, Gl b) { return be.glop(PRE_B, b, be.gl_lit(GL_NEG_ONE), a); }                   // :274-283
    HF Big mul_no_reduce(Gl a, Gl b) { return be.gl_gate(PRE_NONE, a, b, be.gl_lit(0)); }          // :285-294  [0, a, b, ab]
    HF Gl mul(Gl a, Gl b) { return be.glop(PRE_NONE, a, b, be.gl_lit(0)); }                         // :296-305
    HF Gl mul_add(Gl a, Gl b, Gl c) { return be.glop(PRE_NONE, a, b, c); }                          // :319-329
    HF Gl reduce(Big a) { return be.gl_reduce(a); }                                                 // :346-368
    // load_constant(K) immediately followed by an op that takes the constant wire (one fused record):
    HF Gl const_add(Gl x, uint64_t k) { return be.glop(PRE_A, be.gl_lit(k), be.gl_lit(1), x); }     // c = load_constant(k); add(x, c)
    HF Gl const_mul(uint64_t k, Gl x) { return be.glop(PRE_A, be.gl_lit(k), x, be.gl_lit(0)); }     // c = load_constant(k); mul(c, x)
    HF Gl const_mul_add(uint64_t k, Gl x, Gl acc) { return be.glop(PRE_A, be.gl_lit(k), x, acc); }  // c = load_constant(k); mul_add(c, x, acc)
    HNI Gl div(Gl a, Gl b) {                                                                         // :371-393
        if constexpr (B::kHintOps) return be.gl_div(a, b);      // the hint (:382) and its cells are the boundary's: one level-2 call (h2w_gl_div), replayable
        uint64_t bv = be.gl_val(b), av = be.gl_val(a);
        if (bv == 0) { be.fail(1); bv = 1; }                 // reference: assert!(b != 0) (:379)
        Gl res = load_witness(gl_mul(av, gl_inv(bv)));
        Gl prod = mul(b, res); be.assert_equal(a, prod);      // assert_equal adds no cells (a copy constraint; keygen bookkeeping only)
        return res;
    }
    HF Gl inv(Gl a) { Gl one = load_one(); return div(one, a); }                                    // :395-399
    HF Gl square(Gl a) { return mul(a, a); }                                                        // :401-404
    HNI Gl exp_from_bits_const_base(uint64_t base, const Bool *bits, int n) {                        // :407-430
        Gl product = load_one(); uint64_t bp = base;          // base^(2^i)
        for (int i = 0; i < n; i++) {
            Gl a = const_mul(gl_sub(bp, 1), product);
            product = mul_add(a, be.bool_as_gl(bits[i]), product);
            bp = gl_mul(bp, bp);
        }
        return product;
    }
    HF Gl exp_power_of_2(Gl base, int power_log) { Gl p = base; for (int i = 0; i < power_log; i++) p = square(p); return p; } // :433-445
};

// =========================================================================== GoldilocksQuadExtChip (field/goldilocks/extension.rs)
template <class B> struct ExtW { typename B::Gl e[2]; };
template <class B> struct QuadExtChip {
    typedef typename B::Gl Gl; typedef typename B::Bool Bool; typedef typename B::Big Big; typedef ExtW<B> Ex;
    B &be; GoldilocksChip<B> gl;
    HF explicit QuadExtChip(B &b) : be(b), gl(b) {}
    HF gle_t value(const Ex &a) { gle_t v; v.c[0] = be.gl_val(a.e[0]); v.c[1] = be.gl_val(a.e[1]); return v; }  // :19-25
    HF Ex load_constant(gle_t a) { Ex r; r.e[0] = gl.load_constant(a.c[0]); r.e[1] = gl.load_constant(a.c[1]); return r; } // :52-64
    HF Ex load_zero() { gle_t z; z.c[0] = 0; z.c[1] = 0; return load_constant(z); }   // :66-69
    HF Ex load_one() { gle_t o; o.c[0] = 1; o.c[1] = 0; return load_constant(o); }    // :71-74
    HF Ex load_witness(gle_t a) { Ex r; be.gl_witness2(a.c[0], a.c[1], r.e); return r; }   // :85-97
    HF Ex select_from_idx(const Ex *arr, int n, Gl idx) {                             // :99-118
        Gl a0[MAX_ARITY], a1[MAX_ARITY];
        for (int i = 0; i < n; i++) { a0[i] = arr[i].e[0]; a1[i] = arr[i].e[1]; }
        Ex r; r.e[0] = gl.select_from_idx(a0, n, idx); r.e[1] = gl.select_from_idx(a1, n, idx); return r;
    }
    HF Ex load_base(Gl a) { Ex r; r.e[1] = gl.load_zero(); r.e[0] = a; return r; }    // :120-128
    HNI Ex add(const Ex &a, const Ex &b) {                                             // :146-155 (add_no_reduce x2, reduce x2)
        Big s0 = gl.add_no_reduce(a.e[0], b.e[0]), s1 = gl.add_no_reduce(a.e[1], b.e[1]);
        Ex r; r.e[0] = gl.reduce(s0); r.e[1] = gl.reduce(s1); return r;
    }
    HNI Ex sub(const Ex &a, const Ex &b) {                                             // :173-182
        Big d0 = gl.sub_no_reduce(a.e[0], b.e[0]), d1 = gl.sub_no_reduce(a.e[1], b.e[1]);
        Ex r; r.e[0] = gl.reduce(d0); r.e[1] = gl.reduce(d1); return r;
    }
    HNI Ex mul(const Ex &a, const Ex &b) {                                             // :211-234
        Gl w = gl.load_constant(7);
        Gl a0b0 = gl.mul(a.e[0], b.e[0]), a1b1 = gl.mul(a.e[1], b.e[1]), wa1b1 = gl.mul(w, a1b1);
        Ex r; r.e[0] = gl.add(a0b0, wa1b1);
        Gl a0b1 = gl.mul(a.e[0], b.e[1]), a1b0 = gl.mul(a.e[1], b.e[0]);
        r.e[1] = gl.add(a0b1, a1b0); return r;
    }
    HNI Ex square(const Ex &a) {                                                       // :248-268
        Gl w = gl.load_constant(7);
        Gl a0a0 = gl.square(a.e[0]), a1a1 = gl.square(a.e[1]), wa1a1 = gl.mul(w, a1a1);
        Ex r; r.e[0] = gl.add(a0a0, wa1a1);
        Gl a0a1 = gl.mul(a.e[0], a.e[1]);
        r.e[1] = gl.add(a0a1, a0a1); return r;
    }
    HF Ex mul_add(const Ex &a, const Ex &b, const Ex &c) { Ex ab = mul(a, b); return add(ab, c); }   // :284-294
    HF Ex mul_sub(const Ex &a, const Ex &b, const Ex &c) { Ex ab = mul(a, b); return sub(ab, c); }   // :308-318
    HNI Ex inv(const Ex &a) {                                                          // :320-340
        Ex i;
        if constexpr (B::kHintOps) be.ext_inv_witness(a.e, i.e);      // the hint (:327) is the boundary's (h2w_gl_ext_inv_witness)
        else {
            gle_t av = value(a);
            if (av.c[0] == 0 && av.c[1] == 0) { be.fail(2); av.c[0] = 1; }
            i = load_witness(gle_inv(av));
        }
        Ex pr = mul(a, i), one = load_one();                  // assert_equal(product, one): no cells
        be.assert_equal(pr.e[0], one.e[0]); be.assert_equal(pr.e[1], one.e[1]);
        return i;
    }
    HF Ex div(const Ex &a, const Ex &b) { Ex bi = inv(b); return mul(a, bi); }        // :237-246
    HF Ex scalar_mul(const Ex &a, Gl s) { Ex r; r.e[0] = gl.mul(a.e[0], s); r.e[1] = gl.mul(a.e[1], s); return r; }   // :342-353
    HF Ex scalar_div(const Ex &a, Gl s) { Ex r; r.e[0] = gl.div(a.e[0], s); r.e[1] = gl.div(a.e[1], s); return r; }   // :355-366
    HF Ex exp_u64(const Ex &base, uint64_t e) {                                       // :382-407
        if (e == 0) return load_one();
        if (e == 1) return base;
        if (e == 2) return mul(base, base);
        Ex cur = base, prod = load_one();
        int nb = 0; for (uint64_t t = e; t; t >>= 1) nb++;
        for (int j = 0; j < nb; j++) { if (j != 0) cur = square(cur); if ((e >> j) & 1) prod = mul(prod, cur); }
        return prod;
    }
    HF Ex exp_power_of_2(const Ex &base, int power_log) { Ex c = base; for (int i = 0; i < power_log; i++) c = square(c); return c; } // :410-422
    template <class TermFn> HF Ex reduce_with_powers(int n, TermFn term, const Ex &scalar) {   // :424-437 (terms[i] via callback)
        Ex sum = load_zero();
        for (int i = n - 1; i >= 0; i--) { sum = mul(sum, scalar); sum = add(sum, term(i)); }
        return sum;
    }
};

// =========================================================================== Goldilocks Poseidon (hash/poseidon/permutation.rs)
template <class B> struct PoseidonPermutationChip {
    typedef typename B::Gl Gl;
    B &be; GoldilocksChip<B> gl; const h2w_poseidon_consts_t *k;
    HF PoseidonPermutationChip(B &b, const h2w_poseidon_consts_t *kk) : be(b), gl(b), k(kk) {}
    HNI Gl mds_row_shf(int r, const Gl *v) {                                           // :43-71
        Gl res = gl.load_constant(0);
        for (int i = 0; i < SPONGE_WIDTH; i++) res = gl.const_mul_add(k->mds_circ[i], v[(i + r) % SPONGE_WIDTH], res);
        res = gl.const_mul_add(k->mds_diag[r], v[r], res);
        return res;
    }
    HF void mds_layer(Gl *st) {                                                       // :73-87
        Gl res[SPONGE_WIDTH]; gl.load_zero_array(SPONGE_WIDTH, res);
        for (int r = 0; r < SPONGE_WIDTH; r++) res[r] = mds_row_shf(r, st);
        for (int i = 0; i < SPONGE_WIDTH; i++) st[i] = res[i];
    }
    HF void partial_first_constant_layer(Gl *st) {                                    // :89-106
        for (int i = 0; i < SPONGE_WIDTH; i++) st[i] = gl.const_add(st[i], k->fast_partial_first_round_constant[i]);
    }
    HNI void mds_partial_layer_init(Gl *st) {                                          // :108-132
        Gl res[SPONGE_WIDTH]; gl.load_zero_array(SPONGE_WIDTH, res);
        res[0] = st[0];
        for (int r = 1; r < SPONGE_WIDTH; r++) for (int c = 1; c < SPONGE_WIDTH; c++)
            res[c] = gl.const_mul_add(k->fast_partial_round_initial_matrix[r - 1][c - 1], st[r], res[c]);
        for (int i = 0; i < SPONGE_WIDTH; i++) st[i] = res[i];
    }
    HNI void mds_partial_layer_fast(Gl *st, int r) {                                   // :134-173
        Gl d = gl.const_mul(k->mds_circ[0] + k->mds_diag[0], st[0]);
        for (int i = 1; i < SPONGE_WIDTH; i++) d = gl.const_mul_add(k->fast_partial_round_w_hats[r][i - 1], st[i], d);
        Gl res[SPONGE_WIDTH]; gl.load_zero_array(SPONGE_WIDTH, res);
        res[0] = d;
        for (int i = 1; i < SPONGE_WIDTH; i++) res[i] = gl.const_mul_add(k->fast_partial_round_vs[r][i - 1], st[0], st[i]);
        for (int i = 0; i < SPONGE_WIDTH; i++) st[i] = res[i];
    }
    HF void constant_layer(Gl *st, int round_ctr) {                                   // :175-193
        for (int i = 0; i < SPONGE_WIDTH; i++) st[i] = gl.const_add(st[i], k->all_round_constants[i + SPONGE_WIDTH * round_ctr]);
    }
    HNI Gl sbox_monomial(Gl x) { Gl x2 = gl.mul(x, x), x4 = gl.mul(x2, x2), x6 = gl.mul(x4, x2); return gl.mul(x6, x); } // :195-207
    HF void sbox_layer(Gl *st) { for (int i = 0; i < SPONGE_WIDTH; i++) st[i] = sbox_monomial(st[i]); }                 // :209-214
    HF void partial_rounds(Gl *st, int &round_ctr) {                                  // :216-239
        partial_first_constant_layer(st); mds_partial_layer_init(st);
        for (int r = 0; r < N_PARTIAL_ROUNDS; r++) {
            st[0] = sbox_monomial(st[0]);
            st[0] = gl.const_add(st[0], k->fast_partial_round_constants[r]);
            mds_partial_layer_fast(st, r);
        }
        round_ctr += N_PARTIAL_ROUNDS;
    }
    HF void full_rounds(Gl *st, int &round_ctr) {                                     // :241-254
        for (int i = 0; i < HALF_N_FULL_ROUNDS; i++) { constant_layer(st, round_ctr); sbox_layer(st); mds_layer(st); round_ctr += 1; }
    }
    HF void load_zero(Gl *st) { gl.load_zero_array(SPONGE_WIDTH, st); }               // :264-268
    HNI void permute(Gl *st) {                                                        // :270-284
        be.glp_note();
        if constexpr (B::kCoopPoseidon) be.coop_poseidon_permute(st, k);              // one wavefront cooperates on the 12-wide state (coop.h)
        else { int rc = 0; full_rounds(st, rc); partial_rounds(st, rc); full_rounds(st, rc); }
    }
    template <class In> HF void absorb_goldilocks(Gl *st, const In &in, int n) {      // :286-301 (overwrite mode); `in`: anything indexable
        for (int off = 0; off < n; off += SPONGE_RATE) {
            int len = n - off < SPONGE_RATE ? n - off : SPONGE_RATE;
            for (int i = 0; i < len; i++) st[i] = in[off + i];
            permute(st);
        }
    }
};

// =========================================================================== BN254 Poseidon (hash/poseidon_bn254/permutation.rs)
template <class B> struct PoseidonBN254PermutationChip {
    typedef typename B::Gl Gl; typedef typename B::Fr Fr;
    B &be; const h2w_poseidon_consts_t *k;
    HF PoseidonBN254PermutationChip(B &b, const h2w_poseidon_consts_t *kk) : be(b), k(kk) {}
    HNI Fr exp5(Fr x) { Fr x2 = be.fr_mul(x, x), x4 = be.fr_mul(x2, x2); return be.fr_mul(x4, x); }    // :48-55
    HF void exp5_state(Fr *st) { for (int i = 0; i < BN_WIDTH; i++) st[i] = exp5(st[i]); }            // :57-62
    HNI void mix(Fr *st, const Fr *m /*[4][4] row-major*/) {                                           // :64-81
        Fr z = be.fr_load_zero(); Fr ns[BN_WIDTH];
        for (int i = 0; i < BN_WIDTH; i++) { ns[i] = z; for (int j = 0; j < BN_WIDTH; j++) ns[i] = be.fr_mul_add(m[j * 4 + i], st[j], ns[i]); }
        for (int i = 0; i < BN_WIDTH; i++) st[i] = ns[i];
    }
    HNI void ark(Fr *st, int it) {                                                                     // :162-170
        for (int i = 0; i < BN_WIDTH; i++) { Fr c = be.fr_const(k->bn_c[it + i]); st[i] = be.fr_add(st[i], c); }
    }
    HNI void partial_rounds(Fr *st) {                                                                  // :83-110
        for (int i = 0; i < BN_PARTIAL_ROUNDS; i++) {
            st[0] = exp5(st[0]);
            Fr c = be.fr_const(k->bn_c[(BN_FULL_ROUNDS / 2 + 1) * BN_WIDTH + i]);
            st[0] = be.fr_add(st[0], c);
            Fr ns0 = be.fr_load_zero();
            for (int j = 0; j < BN_WIDTH; j++) { Fr s = be.fr_const(k->bn_s[(BN_WIDTH * 2 - 1) * i + j]); ns0 = be.fr_mul_add(s, st[j], ns0); }
            for (int kk = 1; kk < BN_WIDTH; kk++) { Fr s = be.fr_const(k->bn_s[(BN_WIDTH * 2 - 1) * i + BN_WIDTH + kk - 1]); st[kk] = be.fr_mul_add(s, st[0], st[kk]); }
            st[0] = ns0;
        }
    }
    HNI void full_rounds(Fr *st, bool is_first) {                                                      // :112-160
        Fr m[16], p[16];
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) m[i * 4 + j] = be.fr_const(k->bn_m[i][j]);
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) p[i * 4 + j] = be.fr_const(k->bn_p[i][j]);
        for (int i = 0; i < BN_FULL_ROUNDS / 2 - 1; i++) {
            exp5_state(st);
            if (is_first) ark(st, (i + 1) * BN_WIDTH);
            else ark(st, (BN_FULL_ROUNDS / 2 + 1) * BN_WIDTH + BN_PARTIAL_ROUNDS + i * BN_WIDTH);
            mix(st, m);
        }
        exp5_state(st);
        if (is_first) { ark(st, (BN_FULL_ROUNDS / 2) * BN_WIDTH); mix(st, p); } else mix(st, m);
    }
    HF void permute(Fr *st) {                                                                         // :190-203
        if constexpr (B::kBnUnits) { be.bn_perm_unit(st, k); return; }      // device chain strands: the quad's sink computes / emits the permutation unit (coop.h)
        else if (be.bn_perm_unit(st, k)) return;
        be.bn_perm_begin();
        ark(st, 0); full_rounds(st, true); partial_rounds(st); full_rounds(st, false);
        be.bn_perm_end();
    }
    template <class In> HF void absorb_goldilocks(Fr *st, const In &in, int n) {                      // :205-228
        for (int off = 0; off < n; off += BN_RATE * 3) {
            int len = n - off < BN_RATE * 3 ? n - off : BN_RATE * 3;
            for (int j = 0, o = 0; o < len; j++, o += 3) {
                int l3 = len - o < 3 ? len - o : 3; Gl t[3];
                for (int u = 0; u < 3; u++) t[u] = in[off + o + (u < l3 ? u : 0)];
                st[j + 1] = be.limbs_to_num(t, l3);
            }
            permute(st);
        }
    }
};

// =========================================================================== HasherChip (hash/mod.rs:52-127; hash/poseidon/hash.rs; hash/poseidon_bn254/hash.rs)
template <class B> struct HashW { typename B::Gl e[4]; typename B::Fr f; };   // PoseidonHashWire (e) / PoseidonBN254HashWire (f)
template <class B> struct HasherChip {
    typedef typename B::Gl Gl; typedef typename B::Bool Bool; typedef typename B::Fr Fr; typedef HashW<B> H;
    B &be; int mode; GoldilocksChip<B> gl; PoseidonPermutationChip<B> pg; PoseidonBN254PermutationChip<B> pb;
    HF HasherChip(B &b, int hash_mode, const h2w_poseidon_consts_t *k) : be(b), mode(hash_mode), gl(b), pg(b, k), pb(b, k) {}
    // the hash mode: a run-time field of the shape, or fixed by the backend (B::kHashMode >= 0: a kernel that only ever runs one kind of
    // Merkle strand does not carry the other hash's code and stack frames)
    HF int md() const { if constexpr (B::kHashMode >= 0) return B::kHashMode; else return mode; }
    HF int max_goldilocks() const { return md() == 0 ? NUM_HASH_OUT_ELTS : 3; }
    HF H load_witness(const uint64_t *w) {    // poseidon/hash.rs:86-96 (as constants!) ; poseidon_bn254/hash.rs:89-98
        H h;
        if (md() == 0) be.gl_const4(w, h.e);
        else { fr_t v; v.l[0] = w[0]; v.l[1] = w[1]; v.l[2] = w[2]; v.l[3] = w[3]; h.f = be.fr_witness(v); }
        return h;
    }
    template <class In> HF H load_goldilocks_slice(const In &in, int n) {   // poseidon/hash.rs:98-112 ; poseidon_bn254/hash.rs:100-114
        H h;
        if (md() == 0) { gl.load_zero_array(NUM_HASH_OUT_ELTS, h.e); for (int i = 0; i < n; i++) h.e[i] = in[i]; }
        else { Gl t[3]; for (int u = 0; u < 3; u++) t[u] = in[u < n ? u : 0]; h.f = be.limbs_to_num(t, n); }
        return h;
    }
    template <class In> HF H hash_no_pad(const In &in, int n) {  // poseidon/hash.rs:161-184 ; poseidon_bn254/hash.rs:156-179
        H h;
        if (md() == 0) { Gl st[SPONGE_WIDTH]; gl.load_zero_array(SPONGE_WIDTH, st); pg.absorb_goldilocks(st, in, n); for (int i = 0; i < 4; i++) h.e[i] = st[i]; }
        else { Fr st[BN_WIDTH]; be.fr_zero_consts4(st); pb.absorb_goldilocks(st, in, n); h.f = st[0]; }
        return h;
    }
    template <class In> HF H hash_or_noop(const In &in, int n) { return n <= max_goldilocks() ? load_goldilocks_slice(in, n) : hash_no_pad(in, n); } // hash/mod.rs:109-119
    HNI H two_to_one(const H &l, const H &r) {                  // poseidon/hash.rs:187-214 ; poseidon_bn254/hash.rs:182-209
        H h;
        if (md() == 0) {
            Gl st[SPONGE_WIDTH]; gl.load_zero_array(SPONGE_WIDTH, st);
            for (int i = 0; i < 4; i++) { st[i] = l.e[i]; st[4 + i] = r.e[i]; }
            pg.permute(st); for (int i = 0; i < 4; i++) h.e[i] = st[i];
        } else { Fr st[BN_WIDTH]; be.fr_zero_consts4(st); st[2] = l.f; st[3] = r.f; pb.permute(st); h.f = st[0]; }
        return h;
    }
    HNI H select(const H &a, const H &b, Bool sel) {            // poseidon/hash.rs:114-126 ; poseidon_bn254/hash.rs:116-127
        H h;
        if (md() == 0) gl.select_array(a.e, b.e, 4, sel, h.e); else h.f = be.fr_select(a.f, b.f, sel);
        return h;
    }
    template <class CapFn> HF H select_from_idx(int n, CapFn cap, Gl idx) {   // poseidon/hash.rs:128-146 ; poseidon_bn254/hash.rs:129-143
        H h;
        if (md() == 0) {
            Bool ind[MAX_CAP]; be.idx_to_indicator(idx, n, ind);
            for (int j = 0; j < 4; j++) { Gl col[MAX_CAP]; for (int i = 0; i < n; i++) col[i] = cap(i).e[j]; h.e[j] = be.select_by_indicator(col, 1, ind, n); }
        } else h.f = be.fr_select_from_idx_fn(n, [&](int i) { return cap(i).f; }, idx);      // (no per-lane column array on the device)
        return h;
    }
    HF void assert_equal(const H &a, const H &b) {      // poseidon/hash.rs:148-159 ; poseidon_bn254/hash.rs:145-154
        if (md() == 0) { for (int i = 0; i < 4; i++) be.assert_equal(a.e[i], b.e[i]); } else be.assert_equal_fr(a.f, b.f);
    }
    HF int to_goldilocks_vec(const H &h, Gl *out) {            // poseidon/hash.rs:22-30 ; poseidon_bn254/hash.rs:29-44
        if (md() == 0) { for (int i = 0; i < 4; i++) out[i] = h.e[i]; return 4; }
        be.decompose_le_56_5(h.f, out); return 5;
    }
};

// =========================================================================== MerkleTreeChip (merkle/mod.rs:57-78)
template <class B> struct MerkleTreeChip {
    typedef typename B::Gl Gl; typedef typename B::Bool Bool; typedef HashW<B> H;
    B &be; HasherChip<B> hs;
    HF MerkleTreeChip(B &b, int mode, const h2w_poseidon_consts_t *k) : be(b), hs(b, mode, k) {}
    template <class LeafT, class BitsT, class SibFn, class CapFn>      // leaf / bits: anything indexable (arrays, proof-word views, packed index bits)
    HF void verify_proof_to_cap_with_cap_index(const LeafT &leaf, int n_leaf, const BitsT &bits, int n_bits, Gl cap_index,
                                               int n_cap, CapFn cap, int n_sib, SibFn sibling) {
        H node = hs.hash_or_noop(leaf, n_leaf);
        int n = n_sib < n_bits ? n_sib : n_bits;
        for (int i = 0; i < n; i++) {
            if (be.merkle_level_skip(node)) continue;      // (device emission windows: the level is another quad's; no-op elsewhere)
            H sib = sibling(i);
            H left = hs.select(sib, node, bits[i]);
            H right = hs.select(node, sib, bits[i]);
            node = hs.two_to_one(left, right);
        }
        if (be.merkle_tail_skip()) return;             // (device emission windows: the cap lookup is the last unit's quad's)
        H root = hs.select_from_idx(n_cap, cap, cap_index);
        hs.assert_equal(root, node);                 // no cells
    }
};

}  // namespace h2w
