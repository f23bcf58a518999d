// abi_backend.cpp — the gadget stack of chips.h / verifier.h driven through the eager C-ABI itself.
//
// AbiBackend implements the backend concept with nothing but include/h2w.h calls (h2w_gl_*, h2w_add, h2w_select, ...):
// it is what the reference's chips would do after the INTEGRATION.md patch, i.e. the drop-in boundary exercised end to
// end.  The h2w_chip_* entry points mirror the reference's unit tests (hash/poseidon/permutation.rs:325-347 test_permute,
// hash/*/hash.rs test_hash_no_pad / test_hash_two_to_one, merkle/mod.rs:136-265, stark/mod.rs:405-518) so the parity
// tests read like them.  Cells still come only from the GPU expansion kernel.
#include <vector>
#include <map>
#include <string>
#include <cstring>
#include "common.h"
#include "verifier.h"

namespace h2w {

struct AbiBackend {
    typedef h2w_assigned_t Gl; typedef h2w_assigned_t Bool; typedef h2w_assigned_t Fr; typedef h2w_assigned_t Big;
    static constexpr bool kCoopPoseidon = false, kSplitOnly = false, kBnUnits = false, kDevSponge = false; static constexpr int kHashMode = -1;
    static constexpr bool kHintOps = true;      // the two hint sites of the reference (base.rs:382, extension.rs:327) are level-2 calls: no value leaves the library (trace.h)
    Gl gl_div(const Gl &a, const Gl &b) { Gl o; memset(&o, 0, sizeof(o)); const int r = h2w_gl_div(ctx, &a, &b, &o); if (r != 0) { if (b.value.l[0] == 0 && (b.value.l[1] | b.value.l[2] | b.value.l[3]) == 0) fail(1); else ck(r); } return o; }
    void ext_inv_witness(const Gl *a, Gl *out) { memset(out, 0, 2 * sizeof(Gl)); const int r = h2w_gl_ext_inv_witness(ctx, a, out); if (r != 0) { if ((a[0].value.l[0] | a[1].value.l[0]) == 0) fail(2); else ck(r); } }
    h2w_ctx *ctx; int mode; const uint64_t *proof; std::vector<h2w_assigned_t> wires; uint32_t status = 0; int rc = 0;
    AbiBackend(h2w_ctx *c, int hash_mode, const uint64_t *proof_words, size_t n_words) : ctx(c), mode(hash_mode), proof(proof_words), wires(n_words) {}
    void ck(int r) { if (r != 0 && rc == 0) rc = r; }
    void fail(uint32_t code) { if (!status) status = code; }
    static bool is_lit(const Gl &w) { return w.has_cell == 0; }
    Gl gl_lit(uint64_t v) { Gl w; memset(&w, 0, sizeof(w)); w.value.l[0] = v; return w; }
    uint64_t gl_val(const Gl &w) { return w.value.l[0]; }
    Gl bool_as_gl(const Bool &b) { return b; }
    void coop_poseidon_permute(Gl *, const h2w_poseidon_consts_t *) {}
    void glp_note() {}
    // ---- Goldilocks level (h2w_gl_*)
    Gl gl_const(uint64_t k) { Gl o; ck(h2w_gl_load_constant(ctx, k, &o)); return o; }
    void gl_const_run(uint64_t k, int n, Gl *out) { for (int i = 0; i < n; i++) out[i] = gl_const(k); }
    void gl_const4(const uint64_t *w, Gl *out) { for (int i = 0; i < 4; i++) out[i] = gl_const(w[i]); }
    Gl gl_witness(uint64_t v) { Gl o; ck(h2w_gl_load_witness(ctx, v, &o)); return o; }
    void gl_witness2(uint64_t a, uint64_t b, Gl *out) { out[0] = gl_witness(a); out[1] = gl_witness(b); }
    Gl cellify(int pre, int which, const Gl &w) { return (pre == which && is_lit(w)) ? gl_const(w.value.l[0]) : w; }   // load_constant(K) first
    Gl glop(int pre, Gl A, Gl B, Gl C) {       // [pre][C, A, B, A*B+C] + reduce  ==  GoldilocksChip::{add,sub,mul,mul_add}
        A = cellify(pre, PRE_A, A); Gl o;
        if (pre == PRE_B) { ck(h2w_gl_sub(ctx, &C, &A, &o)); return o; }                       // sub(a = C, b = A)
        if (is_lit(B) && B.value.l[0] == 1) { ck(h2w_gl_add(ctx, &C, &A, &o)); return o; }     // add(a = C, b = A)
        if (is_lit(C) && C.value.l[0] == 0) { ck(h2w_gl_mul(ctx, &A, &B, &o)); return o; }     // mul(a = A, b = B)
        ck(h2w_gl_mul_add(ctx, &A, &B, &C, &o)); return o;
    }
    Big gl_gate(int pre, Gl A, Gl B, Gl C) {   // the *_no_reduce forms on NativeChip
        Big o;
        if (pre == PRE_B) { Gl m1 = gl_const(B.value.l[0]); ck(h2w_mul_add(ctx, &A, &m1, &C, &o)); return o; }   // sub_no_reduce
        if (is_lit(B) && B.value.l[0] == 1) { ck(h2w_add(ctx, &C, &A, &o)); return o; }
        if (is_lit(C) && C.value.l[0] == 0) { ck(h2w_mul(ctx, &A, &B, &o)); return o; }
        ck(h2w_mul_add(ctx, &A, &B, &C, &o)); return o;
    }
    Gl gl_reduce(const Big &v) { Gl o; ck(h2w_gl_reduce(ctx, &v, &o)); return o; }
    void assert_equal(const Gl &a, const Gl &b) { ck(h2w_constrain_equal(ctx, &a, &b)); }
    void assert_equal_fr(const Fr &a, const Fr &b) { ck(h2w_constrain_equal(ctx, &a, &b)); }
    // ---- NativeChip level
    Gl select(const Gl &a, const Gl &b, const Bool &sel) { Gl o; ck(h2w_select(ctx, &a, &b, &sel, &o)); return o; }
    void idx_to_indicator(const Gl &idx, int len, Bool *out) { ck(h2w_idx_to_indicator(ctx, &idx, (size_t)len, out)); }
    Gl select_by_indicator(const Gl *a, int stride, const Bool *ind, int len) {
        std::vector<Gl> col((size_t)len); for (int i = 0; i < len; i++) col[i] = a[i * stride];
        Gl o; ck(h2w_select_array_by_indicator(ctx, col.data(), (size_t)len, 1, ind, &o)); return o;
    }
    void num_to_bits(const Gl &a, int nbits, Bool *out) { ck(h2w_num_to_bits(ctx, &a, (size_t)nbits, out)); }
    Gl bits_to_num(const Bool *bits, int n) { Gl o; ck(h2w_bits_to_num(ctx, bits, (size_t)n, &o)); return o; }
    void range_check(const Gl &a, int bits) { ck(h2w_range_check(ctx, &a, (size_t)bits)); }
    Fr fr_const(const fr_t &v) { Fr o; ck(h2w_load_constant(ctx, &v, &o)); return o; }
    Fr fr_witness(const fr_t &v) { Fr o; ck(h2w_load_witness(ctx, &v, &o)); return o; }
    Fr fr_load_zero() { Fr o; ck(h2w_load_zero(ctx, &o)); return o; }
    void fr_zero_consts4(Fr *st) { fr_t z[4]; memset(z, 0, sizeof(z)); ck(h2w_load_constants(ctx, z, 4, st)); }
    Fr fr_add(const Fr &a, const Fr &b) { Fr o; ck(h2w_add(ctx, &a, &b, &o)); return o; }
    Fr fr_mul(const Fr &a, const Fr &b) { Fr o; ck(h2w_mul(ctx, &a, &b, &o)); return o; }
    Fr fr_mul_add(const Fr &a, const Fr &b, const Fr &c) { Fr o; ck(h2w_mul_add(ctx, &a, &b, &c, &o)); return o; }
    Fr fr_select(const Fr &a, const Fr &b, const Bool &sel) { Fr o; ck(h2w_select(ctx, &a, &b, &sel, &o)); return o; }
    const FriTab *fri_tab() const { return nullptr; }      // the eager driver computes the FRI gadgets' constants as the reference does
    template <class ColF> Fr fr_select_from_idx_fn(int n, ColF colf, const Gl &idx) {
        std::vector<Fr> col((size_t)n); for (int i = 0; i < n; i++) col[i] = colf(i);
        Fr o; ck(h2w_select_from_idx(ctx, col.data(), (size_t)n, &idx, &o)); return o;
    }
    Fr limbs_to_num(const Gl *in, int n) { Fr o; ck(h2w_limbs_to_num(ctx, in, (size_t)n, 64, &o)); return o; }
    void decompose_le_56_5(const Fr &x, Gl *out) { ck(h2w_decompose_le(ctx, &x, 56, 5, out)); }
    // ---- proof wires
    Gl proof_gl(uint64_t w) { return wires[w]; }
    HashW<AbiBackend> proof_hash(uint64_t w) { HashW<AbiBackend> h; for (int i = 0; i < 4; i++) h.e[i] = wires[w + (mode == 0 ? i : 0)]; h.f = wires[w]; return h; }
    // proof words enter as tagged inputs (h2w_trace_input: a no-op unless the context is recording a trace)
    void load_proof_gl(uint64_t w) { ck(h2w_trace_input(ctx, w, 1)); wires[w] = gl_witness(proof[w]); }
    void load_proof_gl_nocheck(uint64_t w) { fr_t v = fr_from_u64(proof[w]); h2w_assigned_t o; ck(h2w_trace_input(ctx, w, 1)); ck(h2w_load_witness(ctx, &v, &o)); wires[w] = o; }
    void load_proof_hash(uint64_t w) {
        if (mode == 0) for (int i = 0; i < 4; i++) { ck(h2w_trace_input(ctx, w + i, 1)); wires[w + i] = gl_const(proof[w + i]); }            // PoseidonChip::load_witness loads constants (hash.rs:86-96)
        else { fr_t v; for (int i = 0; i < 4; i++) v.l[i] = proof[w + i]; ck(h2w_trace_input(ctx, w, 4)); wires[w] = fr_witness(v); }
    }
    bool coop_load_proof() { return false; }
    void note_cap_hash(uint64_t) {}
    bool merkle_split(int, int) { return false; }
    bool merkle_level_skip(HashW<AbiBackend> &) { return false; }
    bool merkle_tail_skip() { return false; }
    // the scopes the reference's #[count] macro opens around these functions (macro/src/lib.rs:9-61; fri/mod.rs:337, merkle/mod.rs:56)
    void merkle_begin(int, int) { ck(h2w_push_context(ctx, "verify_proof_to_cap_with_cap_index")); }
    void merkle_end(int, int) { ck(h2w_pop_context(ctx)); }
    void query_begin(int) { ck(h2w_push_context(ctx, "verify_query_round")); }
    void query_end(int) { ck(h2w_pop_context(ctx)); }
    bool bn_perm_unit(Fr *, const h2w_poseidon_consts_t *) { return false; }
    void bn_perm_begin() {}
    void bn_perm_end() {}
};

}  // namespace h2w

using namespace h2w;
typedef h2w_assigned_t Av;

static int finish(AbiBackend &be, const char *fn) {
    if (be.rc != 0) return be.rc;
    if (be.status) { set_error(std::string(fn) + ": gadget failed (reference would panic), code " + std::to_string(be.status)); return -1; }
    return 0;
}

extern "C" {

// GoldilocksQuadExtChip (field/goldilocks/extension.rs): op 0 add, 1 sub, 2 mul, 3 square, 4 inv, 5 div, 6 mul_add (c), 7 scalar_mul (b[0]), 8 scalar_div (b[0])
int h2w_chip_ext_op(h2w_ctx *ctx, int op, const Av a[2], const Av b[2], const Av c[2], Av out[2]) {
    AbiBackend be(ctx, 0, nullptr, 0); QuadExtChip<AbiBackend> ext(be);
    ExtW<AbiBackend> A, B, Cc, R; A.e[0] = a[0]; A.e[1] = a[1];
    if (b) { B.e[0] = b[0]; B.e[1] = b[1]; } if (c) { Cc.e[0] = c[0]; Cc.e[1] = c[1]; }
    switch (op) {
        case 0: R = ext.add(A, B); break; case 1: R = ext.sub(A, B); break; case 2: R = ext.mul(A, B); break; case 3: R = ext.square(A); break;
        case 4: R = ext.inv(A); break; case 5: R = ext.div(A, B); break; case 6: R = ext.mul_add(A, B, Cc); break;
        case 7: R = ext.scalar_mul(A, B.e[0]); break; case 8: R = ext.scalar_div(A, B.e[0]); break;
        default: set_error("h2w_chip_ext_op: bad op"); return -1;
    }
    out[0] = R.e[0]; out[1] = R.e[1];
    return finish(be, "h2w_chip_ext_op");
}
int h2w_chip_gl_exp_from_bits_const_base(h2w_ctx *ctx, uint64_t base, const Av *bits, size_t n, Av *out) {   // base.rs:407-430
    AbiBackend be(ctx, 0, nullptr, 0); GoldilocksChip<AbiBackend> gl(be);
    *out = gl.exp_from_bits_const_base(base, bits, (int)n);
    return finish(be, "h2w_chip_gl_exp_from_bits_const_base");
}
int h2w_chip_gl_poseidon_permute(h2w_ctx *ctx, const h2w_poseidon_consts_t *k, const Av in[12], Av out[12]) {   // hash/poseidon/permutation.rs:270-284
    AbiBackend be(ctx, 0, nullptr, 0); PoseidonPermutationChip<AbiBackend> pg(be, k);
    for (int i = 0; i < 12; i++) out[i] = in[i];
    pg.permute(out);
    return finish(be, "h2w_chip_gl_poseidon_permute");
}
int h2w_chip_bn_poseidon_permute(h2w_ctx *ctx, const h2w_poseidon_consts_t *k, const Av in[4], Av out[4]) {     // hash/poseidon_bn254/permutation.rs:190-203
    AbiBackend be(ctx, 1, nullptr, 0); PoseidonBN254PermutationChip<AbiBackend> pb(be, k);
    for (int i = 0; i < 4; i++) out[i] = in[i];
    pb.permute(out);
    return finish(be, "h2w_chip_bn_poseidon_permute");
}
static void to_hw(int mode, const Av *w, HashW<AbiBackend> &h) { for (int i = 0; i < 4; i++) h.e[i] = w[mode == 0 ? i : 0]; h.f = w[0]; }
static void from_hw(int mode, const HashW<AbiBackend> &h, Av *w) { if (mode == 0) for (int i = 0; i < 4; i++) w[i] = h.e[i]; else for (int i = 0; i < 4; i++) w[i] = h.f; }
int h2w_chip_hash_no_pad(h2w_ctx *ctx, const h2w_poseidon_consts_t *k, int hash_mode, const Av *in, size_t n, Av out[4]) {   // HasherChip::hash_no_pad
    AbiBackend be(ctx, hash_mode, nullptr, 0); HasherChip<AbiBackend> hs(be, hash_mode, k);
    from_hw(hash_mode, hs.hash_no_pad(in, (int)n), out);
    return finish(be, "h2w_chip_hash_no_pad");
}
int h2w_chip_two_to_one(h2w_ctx *ctx, const h2w_poseidon_consts_t *k, int hash_mode, const Av l[4], const Av r[4], Av out[4]) {   // HasherChip::two_to_one
    AbiBackend be(ctx, hash_mode, nullptr, 0); HasherChip<AbiBackend> hs(be, hash_mode, k);
    HashW<AbiBackend> L, R; to_hw(hash_mode, l, L); to_hw(hash_mode, r, R);
    from_hw(hash_mode, hs.two_to_one(L, R), out);
    return finish(be, "h2w_chip_two_to_one");
}
// MerkleTreeChip::verify_proof_to_cap_with_cap_index (merkle/mod.rs:57-78); hash wires: 4 (mode 0) or 1 (mode 1) assigned values each
int h2w_chip_merkle_verify(h2w_ctx *ctx, const h2w_poseidon_consts_t *k, int hash_mode, const Av *leaf, size_t n_leaf, const Av *bits, size_t n_bits,
                           const Av *cap_index, const Av *cap, size_t n_cap, const Av *siblings, size_t n_sib) {
    if (!ctx || !k || !leaf || !bits || !cap_index || !cap || (!siblings && n_sib)) { set_error("h2w_chip_merkle_verify: null argument"); return -1; }
    if (hash_mode < 0 || hash_mode > 1 || n_leaf < 1 || n_cap < 1 || n_cap > (size_t)MAX_CAP || n_bits > 64 || n_sib > 64) {
        set_error("h2w_chip_merkle_verify: out of range (hash_mode 0/1, 1 <= n_cap <= 64, n_bits <= 64, n_sib <= 64, n_leaf >= 1)"); return -1;
    }
    AbiBackend be(ctx, hash_mode, nullptr, 0); MerkleTreeChip<AbiBackend> mk(be, hash_mode, k);
    const int hw = hash_mode == 0 ? 4 : 1;
    mk.verify_proof_to_cap_with_cap_index(leaf, (int)n_leaf, bits, (int)n_bits, *cap_index, (int)n_cap,
        [&](int i) { HashW<AbiBackend> h; to_hw(hash_mode, cap + (size_t)i * hw, h); return h; }, (int)n_sib,
        [&](int i) { HashW<AbiBackend> h; to_hw(hash_mode, siblings + (size_t)i * hw, h); return h; });
    return finish(be, "h2w_chip_merkle_verify");
}
// The whole reference test flow through the eager boundary: permutation_chip.load_zero; WitnessChip::load_proof_with_pis;
// StarkChip::verify_proof (stark/mod.rs:483-508)
int h2w_chip_verify_stark(h2w_ctx *ctx, const h2w_shape_t *shape, const h2w_poseidon_consts_t *k, const uint64_t *proof_words) {
    if (!ctx || !shape || !k || !proof_words) { set_error("h2w_chip_verify_stark: null argument"); return -1; }
    if (const char *why = shape_check(*shape)) { set_error(std::string("h2w_chip_verify_stark: unsupported shape: ") + why); return -1; }
    Derived d = derive_shape(*shape); ProofLayout pl = proof_layout(*shape, d);
    AbiBackend be(ctx, shape->hash_mode, proof_words, pl.total);
    Verifier<AbiBackend> V(be, *shape, k);
    ChallengeBlock<AbiBackend> *cb = new ChallengeBlock<AbiBackend>();
    V.run_all(*cb);
    delete cb;
    return finish(be, "h2w_chip_verify_stark");
}


// ---- copy constraints and constant equalities of a PLAN's cell stream (SURVEY 8f row 2 for the batched path).  The batched kernels
// compute on values, not on cell handles, so the lists come from a replay of the shape through the eager keygen context (whose
// handles carry cell offsets: field/native.rs:185-193 constrain_equal, Context::assign_region's copies) - static per shape, host only,
// built once and cached in the plan.  The proof the replay runs on only has to keep the reference's assertions quiet (non-zero
// denominators): words of a fixed pseudo-random sequence, every one a canonical field element.
static void replay_words(std::vector<uint64_t> &words, uint64_t seed) {
    uint64_t x = seed;
    for (uint64_t &w : words) { x += 0x9E3779B97F4A7C15ull; uint64_t z = x; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; w = (z ^ (z >> 31)) >> 4; }      // < 2^60
}
static int plan_replay(h2w_plan *p, const std::vector<uint64_t> &words, std::vector<uint64_t> *pairs, std::vector<uint64_t> &cells, std::vector<fr_t> &values) {
    const h2w_shape_t &sh = plan_shape(p);
    h2w_ctx *c = h2w_ctx_new(sh.lookup_bits, 0, -1);
    if (!c) return -1;
    int rc = h2w_chip_verify_stark(c, &sh, &plan_consts(p), words.data());
    if (rc == 0 && h2w_num_cells(c) != plan_cells(p)) { set_error("h2w_plan_equalities: internal: replay length mismatch"); rc = -1; }
    if (rc == 0) {
        if (pairs) { pairs->resize(2 * h2w_ctx_num_equalities(c)); if (!pairs->empty() && h2w_ctx_equalities(c, pairs->data()) != 0) rc = -1; }
        cells.resize(h2w_ctx_num_const_equalities(c)); values.resize(cells.size());
        if (rc == 0 && !cells.empty() && h2w_ctx_const_equalities(c, cells.data(), values.data()) != 0) rc = -1;
    }
    h2w_ctx_free(c);
    return rc;
}
static int plan_equalities_build(h2w_plan *p) {
    PlanEqualities &E = plan_equalities(p);
    if (E.ready) return 0;
    const h2w_shape_t &sh = plan_shape(p);
    Derived d = derive_shape(sh); ProofLayout pl = proof_layout(sh, d);
    std::vector<uint64_t> wa(pl.total), wb(pl.total);
    replay_words(wa, 0x9E3779B97F4A7C15ull); replay_words(wb, 0xD1B54A32D192ED03ull);
    std::vector<uint64_t> cells_b; std::vector<fr_t> values_b;
    int rc = plan_replay(p, wa, &E.pairs, E.const_cells, E.const_values);
    // the constants that ARE proof words (the reference loads Goldilocks-Poseidon hash wires as constants, hash/poseidon/hash.rs:86-96):
    // the cells whose constant changes with the proof; a second replay on other words finds them and which word each one holds
    if (rc == 0 && sh.hash_mode == 0) rc = plan_replay(p, wb, nullptr, cells_b, values_b);
    if (rc == 0) {
        E.const_word.assign(E.const_cells.size(), -1);
        if (sh.hash_mode == 0) {
            if (cells_b != E.const_cells) { set_error("h2w_plan_equalities: internal: constant-equality cells depend on the proof"); rc = -1; }
            std::map<uint64_t, int64_t> where;
            for (size_t w = 0; w < wa.size(); w++) where[wa[w]] = (int64_t)w;
            for (size_t i = 0; rc == 0 && i < E.const_cells.size(); i++) {
                if (fr_eq(E.const_values[i], values_b[i])) continue;
                auto it = where.find(E.const_values[i].l[0]);
                if (it == where.end() || (E.const_values[i].l[1] | E.const_values[i].l[2] | E.const_values[i].l[3]) || values_b[i].l[0] != wb[(size_t)it->second]) {
                    set_error("h2w_plan_equalities: internal: a proof-dependent constant is not a proof word"); rc = -1;
                } else E.const_word[i] = it->second;
            }
        }
    }
    if (rc != 0) { E.pairs.clear(); E.const_cells.clear(); E.const_values.clear(); E.const_word.clear(); return -1; }
    E.ready = true;
    return 0;
}
uint64_t h2w_plan_num_equalities(h2w_plan *p) { return p && plan_equalities_build(p) == 0 ? plan_equalities(p).pairs.size() / 2 : 0; }
uint64_t h2w_plan_num_const_equalities(h2w_plan *p) { return p && plan_equalities_build(p) == 0 ? plan_equalities(p).const_cells.size() : 0; }
int h2w_plan_equalities(h2w_plan *p, uint64_t *pairs) {
    if (!p || !pairs) { set_error("h2w_plan_equalities: null argument"); return -1; }
    if (plan_equalities_build(p) != 0) return -1;
    const PlanEqualities &E = plan_equalities(p);
    memcpy(pairs, E.pairs.data(), E.pairs.size() * sizeof(uint64_t)); return 0;
}
int h2w_plan_const_equalities(h2w_plan *p, const uint64_t *proof_words, uint64_t *cells, h2w_fr_t *values) {
    if (!p || !cells || !values) { set_error("h2w_plan_const_equalities: null argument"); return -1; }
    if (plan_equalities_build(p) != 0) return -1;
    const PlanEqualities &E = plan_equalities(p);
    memcpy(cells, E.const_cells.data(), E.const_cells.size() * sizeof(uint64_t)); memcpy(values, E.const_values.data(), E.const_values.size() * sizeof(fr_t));
    for (size_t i = 0; i < E.const_word.size(); i++) if (E.const_word[i] >= 0) {
        if (!proof_words) { set_error("h2w_plan_const_equalities: with Goldilocks-Poseidon caps the hash wires are constants of the circuit (hash/poseidon/hash.rs:86-96): pass the proof"); return -1; }
        values[i] = fr_from_u64(proof_words[E.const_word[i]]);
    }
    return 0;
}

}  // extern "C"
