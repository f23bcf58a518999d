// verifier.h — protocol layer (L5) over chips.h: WitnessChip, ChallengerChip, FriChip, StarkChip.
// Same single-source scheme as chips.h.  The top-level flow of the reference test
// (stark/mod.rs:483-508: permutation_chip.load_zero; load_proof_with_pis; StarkChip::verify_proof) is cut into
// *strands* whose cell ranges are independent once the Fiat-Shamir challenges are known:
//     prologue (per proof)   : zero state, witness load, get_stark_challenges, fri_instance_info, PoW, from_os_and_alpha
//     query    (per query)   : FriChip::verify_query_round minus its Merkle proofs
//     merkle   (per query x tree) : MerkleTreeChip::verify_proof_to_cap_with_cap_index
// A sequential backend (plan / eager ABI) runs them in the reference order; the device runs one lane per strand.
#pragma once
#include <type_traits>
#include "chips.h"

namespace h2w {

constexpr int MAX_QUERIES = 128;
constexpr int CH_BUF = (5 * MAX_CAP > 2 * MAX_FINAL_POLY ? 5 * MAX_CAP : 2 * MAX_FINAL_POLY) + 16;      // the sponge's input buffer: a cap of BN254 hashes (5 limbs each) or the final polynomial + the PoW witness
constexpr int MK_KINDS = 3 + MAX_STEPS;   // merkle strand kinds: initial oracle o (0..2), fold step i (3+i)

// wires produced by the prologue and consumed by the query strands (fri/mod.rs:64-69 FriChallengesWire + instance points)
template <class B> struct ChallengeBlock {
    ExtW<B> zeta, zeta_next, fri_alpha, fri_betas[MAX_STEPS], reduced_openings[2];
    typename B::Gl fri_pow_response;
    typename B::Gl fri_query_indices[MAX_QUERIES];
};

// =========================================================================== ChallengerChip (challenger/mod.rs)
// B::kDevSponge: the sponge lives in the backend's sink (the device prologue wavefront keeps the duplex state on its lanes and the input
// buffer in LDS, coop.h); otherwise here, as the reference's struct does.
template <class B> struct ChallengerChip {
    typedef typename B::Gl Gl; typedef HashW<B> H;
    B &be; PoseidonPermutationChip<B> pg; HasherChip<B> &hs;
    struct Buffers { Gl state[SPONGE_WIDTH]; Gl in[CH_BUF]; Gl out[SPONGE_RATE]; }; struct NoBuffers {};
    typename std::conditional<B::kDevSponge, NoBuffers, Buffers>::type m; int n_in, n_out;
    HF ChallengerChip(B &b, HasherChip<B> &h, const h2w_poseidon_consts_t *k) : be(b), pg(b, k), hs(h), n_in(0), n_out(0) {}
    HF void load_zero_state() {                                                                         // stark/mod.rs:497-499 (permutation_chip.load_zero)
        if constexpr (B::kDevSponge) { Gl z[SPONGE_WIDTH]; pg.load_zero(z); be.sponge_init(); } else pg.load_zero(m.state);
    }
    HF void observe_element(Gl t) {                                                                     // :45-50
        if constexpr (B::kDevSponge) be.sponge_observe(t);
        else { n_out = 0; if (n_in < CH_BUF) m.in[n_in++] = t; else be.fail(3); }
    }
    HF void observe_hash(const H &h) { Gl v[5]; int n = hs.to_goldilocks_vec(h, v); for (int i = 0; i < n; i++) observe_element(v[i]); } // :59-63
    HF void observe_extension_element(const ExtW<B> &e) { observe_element(e.e[0]); observe_element(e.e[1]); }   // :76-78
    // n extension elements that are proof words (word(i): first word of element i), in order
    template <class WordFn> HF void observe_ext_words(int n, WordFn word) {
        if constexpr (B::kDevSponge) be.sponge_observe_words(2 * n, [&](int j) { return word(j >> 1) + (uint64_t)(j & 1); });
        else for (int i = 0; i < n; i++) { ExtW<B> e; e.e[0] = be.proof_gl(word(i)); e.e[1] = be.proof_gl(word(i) + 1); observe_extension_element(e); }
    }
    HNI void absorb_buffered_inputs() {                                                                 // :260-277
        if constexpr (!B::kDevSponge) {
            if (n_in == 0) return;
            pg.absorb_goldilocks(m.state, m.in, n_in);
            for (int i = 0; i < SPONGE_RATE; i++) m.out[i] = m.state[i];
            n_out = SPONGE_RATE; n_in = 0;
        }
    }
    HNI Gl get_challenge() {                                                                            // :92-108
        if constexpr (B::kDevSponge) return be.sponge_challenge();
        else {
            absorb_buffered_inputs();
            if (n_out == 0) { pg.permute(m.state); for (int i = 0; i < SPONGE_RATE; i++) m.out[i] = m.state[i]; n_out = SPONGE_RATE; }
            return m.out[--n_out];
        }
    }
    HF ExtW<B> get_extension_challenge() { ExtW<B> r; r.e[0] = get_challenge(); r.e[1] = get_challenge(); return r; }  // :119-126
};

// index bits of a query as the device Merkle strands see them: bit i of the strand = bit lo + i of the query index (no per-lane array)
struct PackedBits { uint64_t x; int lo; HF uint64_t operator[](int i) const { return (x >> (lo + i)) & 1; } };

// Limits of a shape, shared by every entry point that takes one (plan compile, the eager StarkChip driver): the chips size their
// stack arrays by MAX_*; nullptr = fine.
inline const char *shape_check(const h2w_shape_t &s) {
    if (s.lookup_bits < 2 || s.lookup_bits > 28) return "lookup_bits outside [2, 28]";
    if (s.num_queries < 1 || s.num_queries > MAX_QUERIES) return "num_queries outside [1, 128]";
    if (s.cap_height < 0 || (1 << s.cap_height) > MAX_CAP) return "cap_height outside [0, 6]";
    if (s.arity_bits < 1 || (1 << s.arity_bits) > MAX_ARITY) return "arity_bits outside [1, 4]";
    if (s.n_cols < 1 || s.n_perm_z < 0 || s.n_quotient < 1 || s.n_cols + s.n_perm_z + s.n_quotient > MAX_BATCH_POLYS) return "too many / too few committed polynomials";
    if (s.hash_mode < 0 || s.hash_mode > 1) return "hash_mode must be 0 or 1";
    if (s.degree_bits < 0 || s.rate_bits < 0 || s.degree_bits + s.rate_bits > 63 || s.degree_bits + s.rate_bits < s.cap_height) return "degree_bits / rate_bits do not fit the cap height or 63 bits";
    if (s.pow_bits < 0 || s.pow_bits > 63 || s.n_pis < 0 || s.num_challenges < 0 || s.final_poly_bits < 0) return "negative or oversized protocol parameter";
    if (s.n_perm_z > 0 && s.perm_batch_size < 1) return "perm_batch_size < 1";
    if (derive_shape(s).final_poly_len > MAX_FINAL_POLY) return "final polynomial too long";
    return nullptr;
}

// =========================================================================== the verifier, strand by strand
template <class B> struct Verifier {
    typedef typename B::Gl Gl; typedef typename B::Bool Bool; typedef typename B::Fr Fr; typedef ExtW<B> Ex; typedef HashW<B> H;
    B &be; const h2w_shape_t &s; const h2w_poseidon_consts_t *k; Derived d; ProofLayout pl;
    GoldilocksChip<B> gl; QuadExtChip<B> ext; HasherChip<B> hs; MerkleTreeChip<B> mk;
    HF Verifier(B &b, const h2w_shape_t &sh, const h2w_poseidon_consts_t *kk)
        : be(b), s(sh), k(kk), d(derive_shape(sh)), pl(proof_layout(sh, derive_shape(sh))), gl(b), ext(b), hs(b, sh.hash_mode, kk), mk(b, sh.hash_mode, kk) {}

    HF Ex proof_ext(uint64_t w) { Ex e; e.e[0] = be.proof_gl(w); e.e[1] = be.proof_gl(w + 1); return e; }
    // ---- WitnessChip::load_proof_with_pis (witness/mod.rs:267-294), in flat-layout order
    HF void load_gl(uint64_t w) { if (s.witness_load_range_check) be.load_proof_gl(w); else be.load_proof_gl_nocheck(w); }   // :48-51
    HNI void load_proof_with_pis() {
        if (be.coop_load_proof()) return;      // device prologue wavefront: items striped over the 64 lanes (coop.h)
        uint64_t w = 0;
        for (int i = 0; i < 2 * d.cap_size; i++, w += 4) be.load_proof_hash(w);                 // trace_cap, quotient_polys_cap (:245-246)
        uint64_t n_open = 2ull * (2 * s.n_cols + 2 * s.n_perm_z + s.n_quotient);
        for (uint64_t i = 0; i < n_open; i++, w++) load_gl(w);                                  // load_openings_set (:129-147)
        if (s.n_perm_z > 0) for (int i = 0; i < d.cap_size; i++, w += 4) be.load_proof_hash(w); // permutation_zs_cap (:251-254)
        load_gl(w++);                                                                           // pow_witness (:155)
        for (int i = 0; i < 2 * d.final_poly_len; i++, w++) load_gl(w);                         // final_poly (:157-164)
        for (int i = 0; i < d.n_steps * d.cap_size; i++, w += 4) be.load_proof_hash(w);         // commit_phase_merkle_caps (:166-170)
        for (int q = 0; q < s.num_queries; q++) {                                               // query_round_proofs (:172-225)
            for (int o = 0; o < d.n_oracles; o++) {
                for (int i = 0; i < d.oracle_polys[o]; i++, w++) load_gl(w);
                for (int i = 0; i < pl.init_sibs; i++, w += 4) be.load_proof_hash(w);
            }
            for (int st = 0; st < d.n_steps; st++) {
                for (int i = 0; i < (2 << d.arity[st]); i++, w++) load_gl(w);
                for (int i = 0; i < pl.step_sibs[st]; i++, w += 4) be.load_proof_hash(w);
            }
        }
        for (int i = 0; i < s.n_pis; i++, w++) load_gl(w);                                      // public inputs (:285-288)
    }
    HF void observe_cap(ChallengerChip<B> &ch, uint64_t w0) {                                       // challenger/mod.rs:65-74
        if constexpr (B::kDevSponge) be.sponge_observe_cap(w0, d.cap_size);      // (the cells of the BN254 hashes' limb decompositions: the load kernel)
        else for (int i = 0; i < d.cap_size; i++) { be.note_cap_hash(w0 + 4ull * i); ch.observe_hash(be.proof_hash(w0 + 4ull * i)); }
    }
    // openings in to_fri_openings() order (stark/mod.rs:48-69): zeta batch = local, perm_zs, quotient ; zeta_next batch = next, perm_zs_next
    HF uint64_t zeta_word(int i) const {
        uint64_t o = pl.openings;
        if (i < s.n_cols) return o + 2ull * i;
        i -= s.n_cols; if (i < s.n_perm_z) return o + 2ull * (2 * s.n_cols + i);
        i -= s.n_perm_z; return o + 2ull * (2 * s.n_cols + 2 * s.n_perm_z + i);
    }
    HF uint64_t zeta_next_word(int i) const {
        uint64_t o = pl.openings;
        if (i < s.n_cols) return o + 2ull * (s.n_cols + i);
        i -= s.n_cols; return o + 2ull * (2 * s.n_cols + s.n_perm_z + i);
    }
    // ---- prologue strand
#if defined(H2W_EXP_GLP_CLOCK) && defined(__HIP_DEVICE_COMPILE__)      // experiment: where the prologue wavefront's cycles go (proof 0, lane 0 prints)
#define H2W_CLK_MARK(what) do { if constexpr (B::kDevSponge) { if (be.sink.dbg_k < 24) { be.sink.dbg_t[be.sink.dbg_k] = (long long)clock64() - clk0; be.sink.dbg_c[be.sink.dbg_k] = be.sink.dbg_cycles; be.sink.dbg_m[be.sink.dbg_k++] = be.sink.dbg_n; } } } while (0)
#else
#define H2W_CLK_MARK(what) do { } while (0)
#endif
    HF void prologue(ChallengeBlock<B> &cb) {
#if defined(H2W_EXP_GLP_CLOCK) && defined(__HIP_DEVICE_COMPILE__)
        const long long clk0 = clock64();
#endif
        ChallengerChip<B> ch(be, hs, k);
        ch.load_zero_state();                                        // stark/mod.rs:497-499
        load_proof_with_pis();                                       // stark/mod.rs:506
        H2W_CLK_MARK("load_proof");
        // ChallengerChip::get_stark_challenges (challenger/mod.rs:167-222)
        observe_cap(ch, pl.trace_cap);
        if (s.n_perm_z > 0) {
            for (int i = 0; i < s.perm_batch_size * s.num_challenges * 2; i++) ch.get_challenge();   // get_n_permutation_challenge_sets (:224-256)
            observe_cap(ch, pl.perm_cap);
        }
        for (int i = 0; i < s.num_challenges; i++) ch.get_challenge();                               // stark_alphas (:203)
        observe_cap(ch, pl.quotient_cap);
        H2W_CLK_MARK("caps and alphas");
        const Ex zeta = ch.get_extension_challenge(); cb.zeta = zeta;                               // :206 (what the strand goes on to use is kept in locals: the block is device memory behind a generic reference - every read of it a flat load behind the record stores)
        const int nz = s.n_cols + s.n_perm_z + s.n_quotient, nzn = s.n_cols + s.n_perm_z;
        ch.observe_ext_words(nz, [&](int i) { return zeta_word(i); });                               // observe_openings (:208)
        ch.observe_ext_words(nzn, [&](int i) { return zeta_next_word(i); });
        H2W_CLK_MARK("zeta and openings");
        // get_fri_challenges (:128-165)
        const Ex fri_alpha = ch.get_extension_challenge(); cb.fri_alpha = fri_alpha;
        for (int i = 0; i < d.n_steps; i++) { observe_cap(ch, pl.commit_caps + (uint64_t)i * d.cap_size * 4); H2W_CLK_MARK("commit cap observed"); cb.fri_betas[i] = ch.get_extension_challenge(); H2W_CLK_MARK("beta"); }
        H2W_CLK_MARK("fri alpha and betas");
        ch.observe_ext_words(d.final_poly_len, [&](int i) { return pl.final_poly + 2ull * i; });
        ch.observe_element(be.proof_gl(pl.pow_witness));
        const Gl pow_response = ch.get_challenge(); cb.fri_pow_response = pow_response;
        H2W_CLK_MARK("final poly and pow");
        for (int i = 0; i < s.num_queries; i++) cb.fri_query_indices[i] = ch.get_challenge();
        H2W_CLK_MARK("query indices");
        // verify_proof_with_challenges: fri_instance_info (stark/mod.rs:144-200): zeta_next = g * zeta
        {   // (the generator of the trace domain is the LDE domain's, squared rate_bits times: with the shape's table at hand that replaces an exponentiation - 33 k cycles of the prologue wavefront)
            const FriTab *ft = be.fri_tab();
            uint64_t gd;
            if (ft && ft->lde_bits == s.degree_bits + s.rate_bits) { gd = ft->root_lde; for (int i = 0; i < s.rate_bits; i++) gd = gl_mul(gd, gd); }
            else gd = gl_primitive_root_of_unity(s.degree_bits);
            gle_t gv; gv.c[0] = gd; gv.c[1] = 0; Ex g = ext.load_constant(gv); cb.zeta_next = ext.mul(g, zeta);
        }
        H2W_CLK_MARK("zeta_next");
        // FriChip::verify_fri_proof (fri/mod.rs:446-502): PoW (:130-145), from_os_and_alpha (:45-62)
        be.range_check(pow_response, 64 - s.pow_bits);
        H2W_CLK_MARK("pow range check");
        if constexpr (B::kDevSponge) {      // (the device prologue wavefront: the opening words side by side into the idle input buffer, not one dependent load each)
            static_assert(CH_BUF >= 4 * MAX_BATCH_POLYS, "the openings fit the sponge's input buffer");
            be.stage_words(0, 2 * nz, [&](int j) { return zeta_word(j >> 1) + (uint64_t)(j & 1); });
            be.stage_words(2 * nz, 2 * nzn, [&](int j) { return zeta_next_word(j >> 1) + (uint64_t)(j & 1); });
            cb.reduced_openings[0] = ext.reduce_with_powers(nz, [&](int i) { Ex e; e.e[0] = be.staged_word(2 * i); e.e[1] = be.staged_word(2 * i + 1); return e; }, fri_alpha);
            H2W_CLK_MARK("reduced openings 0");
            cb.reduced_openings[1] = ext.reduce_with_powers(nzn, [&](int i) { Ex e; e.e[0] = be.staged_word(2 * nz + 2 * i); e.e[1] = be.staged_word(2 * nz + 2 * i + 1); return e; }, fri_alpha);
        } else {
            cb.reduced_openings[0] = ext.reduce_with_powers(nz, [&](int i) { return proof_ext(zeta_word(i)); }, fri_alpha);
            cb.reduced_openings[1] = ext.reduce_with_powers(nzn, [&](int i) { return proof_ext(zeta_next_word(i)); }, fri_alpha);
        }
        H2W_CLK_MARK("reduced openings");
    }
    // ---- merkle strand: kind < 3: initial oracle `kind`; kind >= 3: fold step kind-3.  bits/cap_index are wires of the query.
    HF uint64_t query_word(int q) const { return pl.queries + (uint64_t)q * pl.query_words; }
    HF uint64_t initial_cap_word(int o) const {   // merkle_caps = [trace, perm_zs?, quotient] (stark/mod.rs:323-326)
        const uint64_t t = pl.trace_cap, p = pl.perm_cap, q = pl.quotient_cap;      // (read, THEN picked: loads under the conditions are merged into one load at a selected offset, and an object read at a run-time offset stays in scratch memory - chips.h SelArr)
        return o == 0 ? t : s.n_perm_z > 0 && o == 1 ? p : q;
    }
    // proof words as an indexable view (a leaf is read once, word by word: no per-lane copy of it)
    struct ProofGl { B *be; uint64_t base; HF Gl operator[](int i) const { return be->proof_gl(base + (uint64_t)i); } };
    template <class BitsT> HF void merkle_strand(int q, int kind, const BitsT &bits, int n_bits, Gl cap_index) {
        const uint64_t qw = query_word(q);
        if (kind < 3) {
            const int o = kind; const uint64_t base = qw + pl.init_off[o]; const int ne = d.oracle_polys[o];
            const uint64_t capw = initial_cap_word(o), sibw = base + ne;
            mk.verify_proof_to_cap_with_cap_index(ProofGl{&be, base}, ne, bits, n_bits, cap_index, d.cap_size,
                [&](int i) { return be.proof_hash(capw + 4ull * i); }, pl.init_sibs, [&](int i) { return be.proof_hash(sibw + 4ull * i); });
        } else {
            const int st = kind - 3; const uint64_t base = qw + pl.step_off[st]; const int ne = 2 << d.arity[st];
            const uint64_t capw = pl.commit_caps + (uint64_t)st * d.cap_size * 4, sibw = base + ne;
            mk.verify_proof_to_cap_with_cap_index(ProofGl{&be, base}, ne, bits, n_bits, cap_index, d.cap_size,
                [&](int i) { return be.proof_hash(capw + 4ull * i); }, pl.step_sibs[st], [&](int i) { return be.proof_hash(sibw + 4ull * i); });
        }
    }
    template <class BitsT> HF void merkle_call(int q, int kind, const BitsT &bits, int n_bits, Gl cap_index) {
        if constexpr (B::kSplitOnly) { be.merkle_split(q, kind); return; }      // glue strands: the cells of this call belong to a merkle strand
        else {
            if (be.merkle_split(q, kind)) return;
            be.merkle_begin(q, kind); merkle_strand(q, kind, bits, n_bits, cap_index); be.merkle_end(q, kind);
        }
    }
    // ---- FriChip pieces (fri/mod.rs)
    HNI Ex combine_initial(int q, const ChallengeBlock<B> &cb, Gl subgroup_x) {                       // :169-220
        const uint64_t qw = query_word(q);
        Ex sx = ext.load_base(subgroup_x);
        Ex sum = ext.load_zero();
        for (int b = 0; b < 2; b++) {
            const int np = b == 0 ? s.n_cols + s.n_perm_z + s.n_quotient : s.n_cols + s.n_perm_z;
            Ex evals[MAX_BATCH_POLYS];
            for (int i = 0; i < np; i++) {     // FriPolynomialInfo order of fri_instance_info (stark/mod.rs:157-196)
                int o, pi, t = i;
                if (t < s.n_cols) { o = 0; pi = t; }
                else { t -= s.n_cols; if (t < s.n_perm_z) { o = 1; pi = t; } else { t -= s.n_perm_z; o = s.n_perm_z > 0 ? 2 : 1; pi = t; } }
                evals[i] = ext.load_base(be.proof_gl(qw + pl.init_off[o] + pi));
            }
            Ex reduced_evals = ext.reduce_with_powers(np, [&](int i) { return evals[i]; }, cb.fri_alpha);
            Ex numerator = ext.sub(reduced_evals, cb.reduced_openings[b]);
            Ex denominator = ext.sub(sx, b == 0 ? cb.zeta : cb.zeta_next);
            Ex denominator_inv = ext.inv(denominator);
            Ex alpha_shift = ext.exp_u64(cb.fri_alpha, (uint64_t)np);
            sum = ext.mul(alpha_shift, sum);
            sum = ext.mul_add(numerator, denominator_inv, sum);
        }
        return sum;
    }
    HNI Ex interpolate_coset(Gl coset_shift, const Ex *values, int n, const Ex &evaluation_point) {   // :222-283
        int arity_bits = 0; while ((1 << arity_bits) < n) arity_bits++;
        Ex shifted = ext.scalar_div(evaluation_point, coset_shift);
        const FriTab *ft = be.fri_tab();                                                   // the shape's constants, when the backend has them tabulated
        uint64_t dom[MAX_ARITY];
        if (ft) { for (int i = 0; i < n; i++) dom[i] = H2W_CLOAD64(&ft->dom[arity_bits][i]); }
        else { const uint64_t g = gl_primitive_root_of_unity(arity_bits); dom[0] = 1; for (int i = 1; i < n; i++) dom[i] = gl_mul(dom[i - 1], g); }      // two_adic_subgroup
        Ex domain[MAX_ARITY]; Gl bw[MAX_ARITY]; Ex wv[MAX_ARITY];
        for (int i = 0; i < n; i++) { gle_t e; e.c[0] = dom[i]; e.c[1] = 0; domain[i] = ext.load_constant(e); }
        for (int i = 0; i < n; i++) {                                                     // barycentric_weights
            uint64_t w;
            if (ft) w = H2W_CLOAD64(&ft->bw[arity_bits][i]);
            else { uint64_t pr = 1; for (int j = 0; j < n; j++) if (j != i) pr = gl_mul(pr, gl_sub(dom[i], dom[j])); w = gl_inv(pr); }
            bw[i] = gl.load_constant(w);
        }
        for (int i = 0; i < n; i++) wv[i] = ext.scalar_mul(values[i], bw[i]);
        Ex eval = ext.load_zero(), tpp = ext.load_one();
        for (int i = 0; i < n; i++) {
            Ex term = ext.sub(shifted, domain[i]);
            Ex next_tpp = ext.mul(tpp, term);
            Ex tmp1 = ext.mul(eval, term), tmp2 = ext.mul(wv[i], tpp);
            eval = ext.add(tmp1, tmp2); tpp = next_tpp;
        }
        return eval;
    }
    HF Ex compute_evaluation(Gl x, const Bool *within_bits, int arity_bits, const Ex *evals_in, const Ex &beta) {  // :285-322
        const int arity = 1 << arity_bits;
        const FriTab *ft = be.fri_tab();
        const uint64_t g_inv = ft ? H2W_CLOAD64(&ft->g_inv[arity_bits]) : gl_exp(gl_primitive_root_of_unity(arity_bits), (uint64_t)arity - 1);
        Ex evals[MAX_ARITY];
        for (int i = 0; i < arity; i++) { int r = 0; for (int b = 0; b < arity_bits; b++) if (i & (1 << b)) r |= 1 << (arity_bits - 1 - b); evals[r] = evals_in[i]; }
        Bool rev[8]; for (int i = 0; i < arity_bits; i++) rev[i] = within_bits[arity_bits - 1 - i];
        Gl start = gl.exp_from_bits_const_base(g_inv, rev, arity_bits);
        Gl coset_start = gl.mul(start, x);
        return interpolate_coset(coset_start, evals, arity, beta);
    }
    // ---- query strand: FriChip::verify_query_round (fri/mod.rs:337-444)
    HF void query_round(int q, const ChallengeBlock<B> &cb) {
        const int n_log = d.lde_bits; const uint64_t qw = query_word(q);
        Bool bits64[64];
        gl.num_to_bits(cb.fri_query_indices[q], 64, bits64);                 // :363
        Bool *x_index_bits = bits64; int nb = n_log;                         // truncate(n_log)
        Gl cap_index = gl.bits_to_num(x_index_bits + nb - s.cap_height, s.cap_height);   // :366-369
        for (int o = 0; o < d.n_oracles; o++) merkle_call(q, o, x_index_bits, nb, cap_index);   // verify_initial_proof (:147-167)
        Gl subgroup_x;
        {   // :379-389
            Gl g = gl.load_constant(7);
            Bool rev[64]; for (int i = 0; i < nb; i++) rev[i] = x_index_bits[nb - 1 - i];
            const FriTab *ft = be.fri_tab();
            Gl phi = gl.exp_from_bits_const_base(ft ? H2W_CLOAD64(&ft->root_lde) : gl_primitive_root_of_unity(n_log), rev, nb);
            subgroup_x = gl.mul(g, phi);
        }
        Ex old_eval = combine_initial(q, cb, subgroup_x);
        for (int i = 0; i < d.n_steps; i++) {                                // :403-438
            const int ab = d.arity[i]; const int ne = 1 << ab; const uint64_t ew = qw + pl.step_off[i];
            Ex evals[MAX_ARITY]; for (int j = 0; j < ne; j++) evals[j] = proof_ext(ew + 2ull * j);
            Bool *coset_index_bits = x_index_bits + ab; const int ncb = nb - ab;
            Gl within = gl.bits_to_num(x_index_bits, ab);
            Ex new_eval = ext.select_from_idx(evals, ne, within);
            be.assert_equal(new_eval.e[0], old_eval.e[0]); be.assert_equal(new_eval.e[1], old_eval.e[1]);       // no cells
            old_eval = compute_evaluation(subgroup_x, x_index_bits, ab, evals, cb.fri_betas[i]);
            merkle_call(q, 3 + i, coset_index_bits, ncb, cap_index);
            subgroup_x = gl.exp_power_of_2(subgroup_x, ab);
            x_index_bits = coset_index_bits; nb = ncb;
        }
        {   // eval_scalar (:324-335)
            Ex point = ext.load_base(subgroup_x);
            Ex eval = ext.reduce_with_powers(d.final_poly_len, [&](int i) { return proof_ext(pl.final_poly + 2ull * i); }, point);
            be.assert_equal(eval.e[0], old_eval.e[0]); be.assert_equal(eval.e[1], old_eval.e[1]);               // no cells
        }
    }
    // bits / cap_index of query q's merkle strand `kind`, recomputed from the challenge value (device merkle lanes)
    HF void run_all(ChallengeBlock<B> &cb) {        // sequential backends: the reference order
        prologue(cb);
        for (int q = 0; q < s.num_queries; q++) { be.query_begin(q); query_round(q, cb); be.query_end(q); }
    }
};

}  // namespace h2w
