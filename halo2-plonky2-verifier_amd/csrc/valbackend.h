// valbackend.h — the value backend: stands in for NativeChip + halo2-base under chips.h / verifier.h.
//
// Wires ARE values here (Goldilocks wire = u64, native wire = canonical Fr, bool wire = 0/1): copy constraints
// are not part of the advice stream.  Every op computes its result natively (Goldilocks mul/reduce on 64-bit
// integer MADs, Montgomery Fr) and hands the *cells* to a Sink either as a 32-byte block record (the big
// Goldilocks templates, expanded later by expand.hip) or as direct cells (small irregular templates and BN254
// Poseidon).  Sinks: DevSink (device memory) and PlanSink (host, counts + metas = the shape compiler).
// Cell templates: SURVEY.md Appendix A; reduce block: field/goldilocks/base.rs:346-368.
#pragma once
#include "records.h"
#include "verifier.h"

namespace h2w {

constexpr int INV_TAB = 96;

struct StrandTable {
    // index 0: query 0, index 1: any query >= 1 (they differ by the one cached load_zero cell)
    uint64_t q_rec0[2], q_cell0[2], q_nrec[2], q_ncell[2];
    uint64_t mk_rec_rel[2][MK_KINDS], mk_cell_rel[2][MK_KINDS], mk_nrec[2][MK_KINDS], mk_ncell[2][MK_KINDS];
    int first_zero_kind;     // merkle kind (of query 0) that emits the Context's first load_zero cell; -1 = none
    uint64_t pro_nrec, pro_ncell, total_rec, total_cell;
    // BN254 permutation units (one unit = the 4,032(+1) cells of one PoseidonBN254 permute call)
    uint64_t q_unit0[2], q_nunit[2], mk_unit_rel[2][MK_KINDS], total_unit; int64_t first_zero_unit;
    uint32_t mk_nunit[MK_KINDS];        // units of a Merkle strand of each kind (the same for every query)
    uint32_t mk_item0[MK_KINDS + 1];    // emission work items of a query: strand kind k owns items [mk_item0[k], mk_item0[k + 1]) - one per unit, one for a unit-less strand
    // Goldilocks-Poseidon permutations in stream order (the proof's permutation list): the prologue's, then query by query the Merkle strands' (Goldilocks caps)
    uint32_t pro_nglp, q_nglp, mk_glp_rel[MK_KINDS], mk_nglp[MK_KINDS], total_glp;
};
HF uint64_t strand_q_unit(const StrandTable &t, int q) { return q == 0 ? t.q_unit0[0] : t.q_unit0[1] + (uint64_t)(q - 1) * t.q_nunit[1]; }
HF uint64_t strand_q_rec(const StrandTable &t, int q) { return q == 0 ? t.q_rec0[0] : t.q_rec0[1] + (uint64_t)(q - 1) * t.q_nrec[1]; }
HF uint64_t strand_q_cell(const StrandTable &t, int q) { return q == 0 ? t.q_cell0[0] : t.q_cell0[1] + (uint64_t)(q - 1) * t.q_ncell[1]; }

// one witness-load item of WitnessChip::load_proof_with_pis (built by the shape compiler; consumed by the cooperative loader)
struct LoadItem { uint32_t word; uint32_t kind; uint64_t rec; uint64_t cell; };   // kind 0: GL load_witness, 1: GL 1-cell, 2: GL hash (4 const cells), 3: BN254 hash (1 cell), 4: limb decomposition of a BN254 cap hash (challenger/mod.rs:65-74)

struct ValCfg {
    const FriTab *fri = nullptr;     // device strands: the plan's table (device memory)
    const uint64_t *proof;           // this proof's flat words
    int mode, L;                     // hash mode, lookup bits
    FrParams P;
    const fr_t *inv_pos, *inv_neg;   // inverses of +-k, k < INV_TAB (Assigned::Rational(1,x) cells of is_zero)
    const StrandTable *st;           // null on the sequential (plan) run
    bool split;                      // true: merkle calls are skipped (their cells belong to merkle strands)
    bool split_bn;                   // true: the sink emits a PoseidonBN254 permutation's cells itself (QuadSink::bn_emit_inline)
    const LoadItem *load_items; uint32_t n_load_items; uint64_t load_nrec, load_ncell;
    uint32_t n_cap_items;            // ... followed by n_cap_items items of kind 4
};

template <class Sink> struct ValBackend {
    typedef uint64_t Gl; typedef uint64_t Bool; typedef fr_t Fr; typedef u128 Big;
    static constexpr bool kCoopPoseidon = Sink::kCoop;
    static constexpr bool kHintOps = false;                    // hints (GoldilocksChip::div, ext inv) are computed by the chips on this backend's values
    HF Gl gl_div(Gl, Gl) { return 0; } HF void ext_inv_witness(const Gl *, Gl *) {}
    static constexpr bool kSplitOnly = Sink::kSplitOnly;       // the backend only ever runs strands whose Merkle proofs are other strands
    static constexpr bool kDevSponge = Sink::kDevSponge;       // the Fiat-Shamir sponge is kept by the sink (device prologue wavefront)
    static constexpr bool kBnUnits = Sink::kBnUnits;           // every PoseidonBN254 permutation of this backend is a unit handled by the sink
    static constexpr int kHashMode = Sink::kHashMode;          // >= 0: the only hash mode this backend is ever run with (-1: the shape's)
    HF int md() const { if constexpr (kHashMode >= 0) return kHashMode; else return cfg.mode; }
    HF const FriTab *fri_tab() const { return cfg.fri; }
    Sink &sink; ValCfg cfg; bool zero_cached; uint32_t status; uint64_t unit_idx = 0;
    HF void coop_poseidon_permute(uint64_t *st, const h2w_poseidon_consts_t *k) { sink.coop_poseidon_permute(st, k); }
    HF void glp_note() { sink.glp_note(); }      // a Goldilocks-Poseidon permutation starts here (the shape compiler counts them)
    HF ValBackend(Sink &s, const ValCfg &c, bool zero_cached_) : sink(s), cfg(c), zero_cached(zero_cached_), status(0) {}

    HF void fail(uint32_t code) { if (!status) status = code; }
    HF Gl gl_lit(uint64_t v) { return v; }
    HF uint64_t gl_val(Gl w) { return w; }
    HF Gl bool_as_gl(Bool b) { return b; }
    HF void cell(const fr_t &v) { sink.cell(v); }
    HF void cell64(uint64_t v) { sink.cell(fr_from_u64(v)); }
    // keygen markers for the NEXT direct cell (no-ops on the device sinks; the shape compiler's PlanSink turns them into the
    // selector / lookup bitmaps of SURVEY §8f): G = a vertical gate starts there, LK = the cell is registered for the range lookup
    HF void assert_equal(Gl, Gl) {}                  // copy constraints are not part of the advice stream
    HF void assert_equal_fr(const Fr &, const Fr &) {}
    HF void G() { sink.gate(); }
    HF void LK() { sink.lookup(); }

    // ---------------------------------------------------------------- Goldilocks block records
    HF Gl gl_const(uint64_t k) { sink.rec(T_CONST1, k, 0, 0, 0); return k; }
    HF void gl_const_run(uint64_t k, int n, Gl *out) {
        if (n == 12) sink.rec(T_REP12, k, 0, 0, 0);
        else if (n == 4) sink.rec(T_CONST4, k, k, k, k);
        else for (int i = 0; i < n; i++) sink.rec(T_CONST1, k, 0, 0, 0);
        for (int i = 0; i < n; i++) out[i] = k;
    }
    HF void gl_const4(const uint64_t *w, Gl *out) { sink.rec(T_CONST4, w[0], w[1], w[2], w[3]); for (int i = 0; i < 4; i++) out[i] = w[i]; }
    HF Gl gl_witness(uint64_t v) { sink.rec(T_LOADW, v, 0, 0, 0); return v; }
    HF void gl_witness2(uint64_t a, uint64_t b, Gl *out) { sink.rec(T_LOADW2, a, b, 0, 0); out[0] = a; out[1] = b; }
    HNI Gl glop(int pre, Gl A, Gl B, Gl C) {
        sink.rec(pre == PRE_NONE ? T_GLOP : pre == PRE_A ? T_KA_GLOP : T_KB_GLOP, A, B, C, 0);
        return gl_reduce128((u128)A * B + C);
    }
    HF Big gl_gate(int pre, Gl A, Gl B, Gl C) { sink.rec(pre == PRE_B ? T_KB_GATE : T_GATE, A, B, C, 0); return (u128)A * B + C; }
    HF Gl gl_reduce(Big v) { sink.rec(T_REDUCE, (uint64_t)v, (uint64_t)(v >> 64), 0, 0); return gl_reduce128(v); }

    // ---------------------------------------------------------------- small native templates: direct cells
    HF fr_t inv_small(const fr_t &x) {
        if ((x.l[1] | x.l[2] | x.l[3]) == 0 && x.l[0] < (uint64_t)INV_TAB) return g_load_fr(cfg.inv_pos + x.l[0]);
        fr_t nx = fr_neg(x);
        if ((nx.l[1] | nx.l[2] | nx.l[3]) == 0 && nx.l[0] < (uint64_t)INV_TAB) return g_load_fr(cfg.inv_neg + nx.l[0]);
        return fr_inv(x, cfg.P);
    }
    HNI Gl select(Gl a, Gl b, Bool sel) {      // [a-b, 1, b, a, b, sel, a-b, out]
        fr_t diff = fr_sub(fr_from_u64(a), fr_from_u64(b)); Gl out = sel ? a : b;
        G(); cell(diff); cell64(1); cell64(b); cell64(a); G(); cell64(b); cell64(sel); cell(diff); cell64(out);     // gates @0, @4
        return out;
    }
    HF Bool is_zero_cells(const fr_t &a) {    // [z, a, inv, 1, 0, a, z, 0]
        bool z = fr_is_zero(a); fr_t inv = z ? fr_from_u64(1) : inv_small(a);
        G(); cell64(z ? 1 : 0); cell(a); cell(inv); cell64(1); G(); cell64(0); cell(a); cell64(z ? 1 : 0); cell64(0);
        return z ? 1 : 0;
    }
    HNI void idx_to_indicator(Gl idx, int len, Bool *out) {       // out may be null (cells only)
        fr_t iv = fr_from_u64(idx);
        for (int i = 0; i < len; i++) {
            Bool z;
            if (i == 0) z = is_zero_cells(iv);
            else { fr_t d = fr_sub(iv, fr_from_u64((uint64_t)i)); G(); cell(d); cell64((uint64_t)i); cell64(1); cell(iv); z = is_zero_cells(d); }
            if (out) out[i] = z;
        }
    }
    HNI Gl select_by_indicator(const Gl *a, int stride, const Bool *ind, int len) {   // [0, a0, ind0, s0, ...]
        u128 sum = 0; if (len > 0) G(); cell64(0);                                   // gates @3i
        for (int i = 0; i < len; i++) { sum += (u128)a[i * stride] * ind[i]; cell64(a[i * stride]); cell64(ind[i]); if (i + 1 < len) G(); sink.cell(fr_from_u128(sum)); }
        return (uint64_t)sum;
    }
    HNI void num_to_bits(Gl a, int nbits, Bool *out) {     // inner_product(bits, 2^i) then assert_bit per bit
        for (int i = 0; i < nbits; i++) out[i] = i < 64 ? (a >> i) & 1 : 0;
        if (nbits > 1) G(); cell64(out[0]);                                          // inner product, short form: gates @0, 3, 6, ...
        for (int i = 1; i < nbits; i++) { cell64(out[i]); sink.cell(fr_pow2(i)); if (i + 1 < nbits) G(); cell64(i >= 63 ? a : (a & ((2ull << i) - 1))); }
        for (int i = 0; i < nbits; i++) { G(); cell64(0); cell64(out[i]); cell64(out[i]); cell64(out[i]); }   // assert_bit
    }
    HF Gl bits_to_num(const Bool *bits, int n) {          // inner_product(bits, [1,2,4,..])
        if (n <= 0) { cell64(0); return 0; }              // inner_product of nothing: the single cell [0] (cap_height 0)
        uint64_t acc = bits[0]; if (n > 1) G(); cell64(bits[0]);
        for (int i = 1; i < n; i++) { acc += bits[i] << i; cell64(bits[i]); sink.cell(fr_pow2(i)); if (i + 1 < n) G(); cell64(acc); }
        return acc;
    }
    HNI void range_check(Gl a, int bits) {                 // RangeChip::range_check on a 64-bit value
        const int L = cfg.L; if (bits == 0) return;
        const int n = (bits + L - 1) / L, rem = bits % L; uint64_t last = a;
        const uint64_t lm = (1ull << L) - 1;
        if (n > 1) {
            G(); LK(); cell64(a & lm);
            for (int j = 1; j < n; j++) {
                uint64_t limb = (j * L >= 64) ? 0 : (a >> (j * L)) & lm;
                LK(); cell64(limb); sink.cell(fr_pow2(j * L)); if (j + 1 < n) G(); cell64((j + 1) * L >= 64 ? a : a & ((1ull << ((j + 1) * L)) - 1));
                last = limb;
            }
        }
        if (rem == 1) { G(); cell64(0); cell64(last); cell64(last); cell64(last); }
        else if (rem > 1) { G(); cell64(0); cell64(last); sink.cell(fr_pow2(L - rem)); LK(); sink.cell(fr_from_u128((u128)last << (L - rem))); }
    }
    // ---------------------------------------------------------------- native Fr templates (BN254 Poseidon): direct cells
    HF Fr fr_const(const fr_t &v) { cell(v); return v; }
    HF Fr fr_witness(const fr_t &v) { cell(v); return v; }
    HF Fr fr_load_zero() { if (!zero_cached) { cell64(0); zero_cached = true; } return fr_zero(); }
    HF void fr_zero_consts4(Fr *st) { for (int i = 0; i < 4; i++) { cell64(0); st[i] = fr_zero(); } }
    HNI Fr fr_add(const Fr &a, const Fr &b) { Fr v = h2w::fr_add(a, b); G(); cell(a); cell(b); cell64(1); cell(v); return v; }
    HNI Fr fr_mul(const Fr &a, const Fr &b) { Fr v = h2w::fr_mul(a, b, cfg.P); G(); cell64(0); cell(a); cell(b); cell(v); return v; }
    HNI Fr fr_mul_add(const Fr &a, const Fr &b, const Fr &c) { Fr v = h2w::fr_add(h2w::fr_mul(a, b, cfg.P), c); G(); cell(c); cell(a); cell(b); cell(v); return v; }
    HNI Fr fr_select(const Fr &a, const Fr &b, Bool sel) {
        Fr diff = fr_sub(a, b); Fr out = sel ? a : b;
        G(); cell(diff); cell64(1); cell(b); cell(a); G(); cell(b); cell64(sel); cell(diff); cell(out);
        return out;
    }
    template <class ColF> HF Fr fr_select_from_idx_fn(int n, ColF col, Gl idx) {      // select_from_idx over col(0..n-1): no indicator / column arrays
        idx_to_indicator(idx, n, nullptr);
        Fr sum = fr_zero(); if (n > 0) G(); cell64(0);
        for (int i = 0; i < n; i++) { const Fr c = col(i); const Bool ind = idx == (uint64_t)i ? 1 : 0; if (ind) sum = h2w::fr_add(sum, c); cell(c); cell64(ind); if (i + 1 < n) G(); cell(sum); }
        return sum;
    }
    HF Fr limbs_to_num(const Gl *in, int n) {             // RangeChip::limbs_to_num(limbs, 64)
        Fr acc = fr_zero(); acc.l[0] = in[0]; if (n > 1) G(); cell64(in[0]);
        for (int i = 1; i < n; i++) { acc.l[i] = in[i]; cell64(in[i]); sink.cell(fr_pow2(64 * i)); if (i + 1 < n) G(); cell(acc); }
        return acc;
    }
    HNI void decompose_le_56_5(const Fr &x, Gl *out) {     // RangeChip::decompose_le(x, 56, 5)
        for (int i = 0; i < 5; i++) out[i] = fr_bits(x, 56 * i, 56);
        G(); cell64(out[0]);
        for (int i = 1; i < 5; i++) {
            cell64(out[i]); sink.cell(fr_pow2(56 * i)); if (i < 4) G();
            Fr acc = x; const int hb = 56 * (i + 1);                 // x mod 2^(56(i+1))
            if (hb < 256) { const int w = hb >> 6, sh = hb & 63; for (int j = w + 1; j < 4; j++) acc.l[j] = 0; acc.l[w] &= sh ? ((1ull << sh) - 1) : 0; }
            cell(acc);
        }
        for (int i = 0; i < 5; i++) range_check(out[i], 56);
    }
    // ---------------------------------------------------------------- proof wires (flat layout, verifier.h ProofLayout)
    HF uint64_t pw(uint64_t w) { return g_load_u64(cfg.proof + w); }
    HF Gl proof_gl(uint64_t w) { return pw(w); }
    HF HashW<ValBackend> proof_hash(uint64_t w) {
        HashW<ValBackend> h;
        if (md() == 0) { for (int i = 0; i < 4; i++) h.e[i] = pw(w + i); h.f = fr_zero(); }
        else { for (int i = 0; i < 4; i++) { h.f.l[i] = pw(w + i); h.e[i] = 0; } }
        return h;
    }
    HF bool coop_load_proof() { return sink.coop_load_proof(cfg); }
    HF void load_proof_gl(uint64_t w) { sink.note_load(w, 0); gl_witness(pw(w)); }
    HF void load_proof_gl_nocheck(uint64_t w) { sink.note_load(w, 1); sink.rec(T_CONST1, pw(w), 0, 0, 0); }
    HF void load_proof_hash(uint64_t w) {
        sink.note_load(w, md() == 0 ? 2 : 3);
        if (md() == 0) sink.rec(T_CONST4, pw(w), pw(w + 1), pw(w + 2), pw(w + 3));
        else { fr_t v; for (int i = 0; i < 4; i++) v.l[i] = pw(w + i); cell(v); }
    }
    // ---------------------------------------------------------------- BN254 permutation units
    HF bool bn_perm_unit(Fr *st, const h2w_poseidon_consts_t *) {
        return cfg.split_bn && sink.bn_emit_inline(st, cfg, zero_cached);      // the strand's own lanes emit the permutation's cells (QuadSink)
    }
    HF void bn_perm_begin() { sink.bn_perm_begin(zero_cached); }
    HF void bn_perm_end() { sink.bn_perm_end(zero_cached); unit_idx++; }
    // ---------------------------------------------------------------- the sponge of a device prologue wavefront (kDevSponge; coop.h)
    HF void note_cap_hash(uint64_t w) { sink.note_cap_hash(w); }      // a cap hash is about to be decomposed into limbs (the shape compiler lists it for the load kernel)
    HF void sponge_init() { sink.sponge_init(); }
    HF void sponge_observe(Gl t) { if (!sink.sponge_observe(t)) fail(3); }
    template <class WordFn> HF void sponge_observe_words(int n, WordFn word) { if (!sink.sponge_observe_words(cfg.proof, n, word)) fail(3); }
    HF void sponge_observe_cap(uint64_t w0, int n) { if (!sink.sponge_observe_cap(cfg.proof, w0, n, md(), cfg.L)) fail(3); }
    HF Gl sponge_challenge() { return sink.sponge_challenge(); }
    template <class WordFn> HF void stage_words(int at, int n, WordFn word) { sink.stage_words(cfg.proof, at, n, word); }      // (the sponge's input buffer, while it is idle)
    HF Gl staged_word(int i) { return sink.staged_word(i); }
    // ---------------------------------------------------------------- strand hooks
    HF bool merkle_split(int q, int kind) {
        if (!cfg.split) return false;
        const int s = q == 0 ? 0 : 1;
        sink.skip(cfg.st->mk_nrec[s][kind], cfg.st->mk_ncell[s][kind]);
        return true;
    }
    HF bool merkle_level_skip(HashW<ValBackend> &node) { if constexpr (kBnUnits) return sink.level_skip(node.f, zero_cached); else return false; }
    HF bool merkle_tail_skip() { if constexpr (kBnUnits) return sink.tail_skip(); else return false; }
    HF void merkle_begin(int q, int kind) { sink.merkle_begin(q, kind, zero_cached, unit_idx); }
    HF void merkle_end(int q, int kind) { sink.merkle_end(q, kind, zero_cached); }
    HF void query_begin(int q) { sink.query_begin(q, unit_idx); }
    HF void query_end(int q) { sink.query_end(q, unit_idx); }
};

// device sink: records into this proof's record array, direct cells into this proof's advice range
template <bool COLS, bool SPLIT_ONLY = false> struct DevSinkT {
    static constexpr bool kCoop = false, kSplitOnly = SPLIT_ONLY, kBnUnits = false, kDevSponge = false; static constexpr int kHashMode = -1;
    HF void coop_poseidon_permute(uint64_t *, const h2w_poseidon_consts_t *) {}
    rec_t *recs; uint64_t nrec; fr_t *out; uint64_t cell_off; const uint16_t *ncells; ColPolicy<COLS> cc;
    HF void rec(int t, uint64_t a, uint64_t b, uint64_t c, uint64_t d) { g_store_rec(recs + nrec, a, b, c, d); nrec++; cell_off += ncells[t]; }
    HF void cell(const fr_t &v) { g_store_fr(out + cc.map(cell_off), v); cell_off++; }
    HF void gate() {}
    HF void lookup() {}
    HF void skip(uint64_t nr, uint64_t nc) { nrec += nr; cell_off += nc; }
    HF void merkle_begin(int, int, bool, uint64_t) {}
    HF void merkle_end(int, int, bool) {}
    HF void query_begin(int, uint64_t) {}
    HF void query_end(int, uint64_t) {}
    HF void bn_perm_begin(bool) {}
    HF void bn_perm_end(bool) {}
    HF void glp_note() {}
    HF void note_load(uint64_t, int) {}
    HF void note_cap_hash(uint64_t) {}
    HF bool coop_load_proof(const ValCfg &) { return false; }
    HF bool bn_emit_inline(fr_t *, const ValCfg &, bool &) { return false; }
};
typedef DevSinkT<false> DevSink;

}  // namespace h2w
