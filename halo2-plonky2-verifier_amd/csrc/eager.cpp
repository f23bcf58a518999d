// eager.cpp — API levels 1 and 2 of include/h2w.h: the NativeChip-shaped eager context.
//
// Mirrors verifier/src/field/native.rs (every method 1:1) and the fused GoldilocksChip ops of
// verifier/src/field/goldilocks/base.rs.  Values are computed here on the host at call time (the reference
// reads AssignedValue::value() for its hints); the advice CELLS are not: each call appends a compact 32-byte
// record (or, for rare wide/irregular templates, literal cells) and the GPU expansion kernel (expand.hip)
// materialises the stream when the advice is requested.  There is no CPU cell path: without a HIP device
// h2w_ctx_advice_device / h2w_ctx_download fail.
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <cstring>
#include <cstdio>
#include <atomic>
#include "records.h"
#include "common.h"
#include "keygen.h"
#include "trace.h"
#include "host_fr.h"
#include "poseidon_tables.h"

namespace h2w {

thread_local std::string g_last_error;
void set_error(const std::string &s) { g_last_error = s; }
static std::atomic<uint32_t> g_next_ctx_id{1};      // contexts may be created from several threads

}  // namespace h2w

using namespace h2w;

struct h2w_ctx {
    int L, witness_gen_only, device; uint32_t id;
    TemplateTable tt; FrParams P; HostFr H;
    std::vector<uint64_t> meta; std::vector<rec_t> recs; std::vector<fr_t> pool;
    uint64_t ncells = 0;
    bool zero_set = false; uint64_t zero_off = 0;
    int err = 0;
    MetaRecorder mr; bool keygen = false;       // witness_gen_only == 0: selectors / equalities / lookups (keygen.h)
    std::vector<fr_t> inv_pos, inv_neg;
    // ContextTree (util/context_tree.rs:11-123): scoped cell counters
    struct Node { int parent; std::string name; uint64_t cells; std::vector<int> children; };
    std::vector<Node> nodes{Node{-1, "all", 0, {}}}; int cur = 0; std::vector<uint64_t> enter;
    // device side
    void *d_out = nullptr, *d_meta = nullptr, *d_recs = nullptr, *d_pool = nullptr;
    DeviceTables dt;
    bool dt_ready = false; size_t dt_nslots = 0, dt_nconsts = 0, dt_ntmpl = 0;
    Trace *trace = nullptr;      // trace mode (h2w_ctx_trace_begin): the op tape of this run (trace.h)
    explicit h2w_ctx(int L_) : tt(L_) {}
    ~h2w_ctx() { delete trace; }
};

namespace {

typedef h2w_assigned_t Av;
// ---- trace mode: one tape entry per call (trace.h).  TrRec is opened before the call's cells are appended and closed after.
struct TrRec {
    h2w_ctx *c; Trace *t; TraceOp op;
    TrRec(h2w_ctx *c_, uint16_t code, uint16_t sub = 0, uint64_t imm = 0);
    void in(const Av *a) { if (!t) return; if (!a->has_cell || a->ctx_id != ctx_id()) t->fail("trace: an operand is not a cell of this context (a value that bypassed the library cannot be replayed)"); t->ins.push_back(TraceIn{a->offset, 0}); op.n_in++; }
    void lit(uint64_t v) { if (!t) return; t->ins.push_back(TraceIn{v, 1}); op.n_in++; }
    void out(const Av *a) { if (!t) return; t->outs.push_back(a->offset); op.n_out++; }
    void tag_required(const char *fn);
    void done();
    uint32_t ctx_id() const;
};
inline bool fits64(const fr_t &v) { return (v.l[1] | v.l[2] | v.l[3]) == 0; }
inline Av mk(h2w_ctx *c, const fr_t &v, uint64_t off) { Av a; a.value = v; a.offset = off; a.ctx_id = c->id; a.has_cell = 1; return a; }

// the host side of a context streams into three vectors (a PoseidonBN254 proof: 350 MB of literal cells), grown by doubling.  (Asking for huge pages
// on the fresh memory - first-touch faults are a large part of a level-1 call - was tried and lost: with THP in madvise mode the fault compacts
// synchronously, 218 -> 466 ns per h2w_mul.)
template <class T> inline void room(std::vector<T> &v) { if (v.size() == v.capacity()) v.reserve(v.capacity() ? v.capacity() * 2 : (size_t)1 << 16); }
inline void lit(h2w_ctx *c, const fr_t &v) {
    if (!c->recs.empty() && meta_tmpl(c->meta.back()) == T_LITERAL && c->recs.back().a + c->recs.back().b == c->pool.size() &&
        meta_off(c->meta.back()) + c->recs.back().b == c->ncells) {
        c->recs.back().b++;
    } else {
        room(c->meta); room(c->recs);
        c->meta.push_back(meta_pack(T_LITERAL, c->ncells));
        c->recs.push_back(rec_t{(uint64_t)c->pool.size(), 1, 0, 0});
    }
    room(c->pool);
    c->pool.push_back(v); c->ncells++;
}
inline void rec(h2w_ctx *c, int t, uint64_t a, uint64_t b, uint64_t cc, uint64_t d) {
    room(c->meta); room(c->recs);
    c->meta.push_back(meta_pack((uint32_t)t, c->ncells));
    c->recs.push_back(rec_t{a, b, cc, d});
    c->ncells += (uint64_t)c->tt.ncells(t);
}
inline void cell(h2w_ctx *c, const fr_t &v) { if (fits64(v)) rec(c, T_CONST1, v.l[0], 0, 0, 0); else lit(c, v); }
inline fr_t fmul(h2w_ctx *c, const fr_t &a, const fr_t &b) { return host_fr_mul(a, b, c->H); }      // (host_fr.h: the product that bounds the eager level)

fr_t inv_cached(h2w_ctx *c, const fr_t &x) {  // Assigned::Rational(1, x) -> x^{-1}
    const size_t N = 96;
    if (c->inv_pos.empty()) {
        c->inv_pos.resize(N); c->inv_neg.resize(N);
        for (size_t k = 1; k < N; k++) { c->inv_pos[k] = fr_inv(fr_from_u64(k), c->P); c->inv_neg[k] = fr_neg(c->inv_pos[k]); }
    }
    if (fits64(x) && x.l[0] < N) return c->inv_pos[x.l[0]];
    fr_t nx = fr_neg(x);
    if (fits64(nx) && nx.l[0] < N) return c->inv_neg[nx.l[0]];
    return fr_inv(x, c->P);
}

// ---- halo2-base templates; each returns the value/offset the reference returns
Av t_gate(h2w_ctx *c, const fr_t &C, const fr_t &A, const fr_t &B) {  // [C, A, B, A*B+C]
    fr_t v = fr_add(fmul(c, A, B), C);
    if (fits64(A) && fits64(B) && fits64(C)) rec(c, T_GATE, A.l[0], B.l[0], C.l[0], 0);
    else { lit(c, C); lit(c, A); lit(c, B); lit(c, v); }
    return mk(c, v, c->ncells - 1);
}
Av t_sub(h2w_ctx *c, const fr_t &a, const fr_t &b) {  // [a-b, b, 1, a] -> cell -4
    fr_t d = fr_sub(a, b);
    lit(c, d); lit(c, b); lit(c, fr_from_u64(1)); lit(c, a);
    return mk(c, d, c->ncells - 4);
}
Av t_select(h2w_ctx *c, const fr_t &a, const fr_t &b, const fr_t &sel) {
    fr_t diff = fr_sub(a, b), out = fr_add(fmul(c, diff, sel), b);
    lit(c, diff); lit(c, fr_from_u64(1)); lit(c, b); lit(c, a); lit(c, b); lit(c, sel); lit(c, diff); lit(c, out);
    return mk(c, out, c->ncells - 1);
}
Av t_is_zero(h2w_ctx *c, const fr_t &a) {  // [z, a, inv, 1, 0, a, z, 0] -> cell -2
    bool z = fr_is_zero(a);
    fr_t zv = fr_from_u64(z ? 1 : 0), inv = z ? fr_from_u64(1) : inv_cached(c, a);
    lit(c, zv); lit(c, a); lit(c, inv); lit(c, fr_from_u64(1)); lit(c, fr_zero()); lit(c, a); lit(c, zv); lit(c, fr_zero());
    return mk(c, zv, c->ncells - 2);
}
void t_idx_to_indicator(h2w_ctx *c, const fr_t &idx, size_t len, Av *out) {
    for (size_t i = 0; i < len; i++) {
        if (i == 0) out[0] = t_is_zero(c, idx);
        else { Av d = t_sub(c, idx, fr_from_u64(i)); out[i] = t_is_zero(c, d.value); }
    }
}
Av t_select_by_indicator(h2w_ctx *c, const Av *a, size_t stride, const Av *ind, size_t len) {
    fr_t sum = fr_zero(); lit(c, fr_zero());
    for (size_t i = 0; i < len; i++) {
        sum = fr_add(sum, fmul(c, a[i * stride].value, ind[i].value));
        lit(c, a[i * stride].value); lit(c, ind[i].value); lit(c, sum);
    }
    return mk(c, sum, c->ncells - 1);
}
// inner_product(a, consts b): returns acc; cell offsets of a[i]: row (i=0, short form) / row+1+3(i-1)
Av t_inner_product(h2w_ctx *c, const fr_t *a, const fr_t *b, size_t n) {
    fr_t sum; size_t start = 0;
    if (n > 0 && fr_eq(b[0], fr_from_u64(1))) { sum = a[0]; lit(c, a[0]); start = 1; }
    else { sum = fr_zero(); lit(c, fr_zero()); }
    for (size_t i = start; i < n; i++) { sum = fr_add(sum, fmul(c, a[i], b[i])); lit(c, a[i]); lit(c, b[i]); lit(c, sum); }
    return mk(c, sum, c->ncells - 1);
}
void t_assert_bit(h2w_ctx *c, const fr_t &x) { lit(c, fr_zero()); lit(c, x); lit(c, x); lit(c, x); }
void t_range_check(h2w_ctx *c, const Av &a, size_t bits) {
    const int L = c->L;
    if (bits == 0) return;
    if (fits64(a.value) && bits <= 64) {   // compact: one record per range check
        int t = c->tt.rc_template((int)bits);
        if (t >= 0) { if (c->tt.ncells(t) > 0) rec(c, t, a.value.l[0], 0, 0, 0); return; }
    }
    size_t n = (bits + L - 1) / L, rem = bits % L; fr_t last = a.value;
    if (n > 1) {
        std::vector<fr_t> limbs(n), bases(n);
        for (size_t i = 0; i < n; i++) { limbs[i] = fr_from_u64(fr_bits(a.value, (int)(i * L), L)); bases[i] = fr_pow2((int)(i * L)); }
        t_inner_product(c, limbs.data(), bases.data(), n);
        last = limbs[n - 1];
    }
    if (rem == 1) t_assert_bit(c, last);
    else if (rem > 1) t_gate(c, fr_zero(), last, fr_pow2((int)(L - rem)));
}
int bit_length(uint64_t b) { int n = 0; while (b) { n++; b >>= 1; } return n; }
void t_check_less_than_safe(h2w_ctx *c, const Av &a, uint64_t b) {
    const int L = c->L; size_t rb = (size_t)((bit_length(b) + L - 1) / L * L);
    if (b == GL_P && fits64(a.value)) { rec(c, T_CLT_SAFE, a.value.l[0], 0, 0, 0); return; }
    t_range_check(c, a, rb);
    fr_t p2 = fr_pow2((int)rb), sh = fr_add(p2, a.value), bv = fr_from_u64(b), d = fr_sub(sh, bv);
    uint64_t first = c->ncells;
    lit(c, d); lit(c, bv); lit(c, fr_from_u64(1)); lit(c, sh); lit(c, fr_neg(p2)); lit(c, fr_from_u64(1)); lit(c, a.value);
    t_range_check(c, mk(c, d, first), rb);
}

typedef MetaRecorder MR;
inline int64_t off(const h2w_assigned_t *a) { return a && a->has_cell ? (int64_t)a->offset : -1; }
// keygen bookkeeping must stay in step with the cell stream
inline int kg_done(h2w_ctx *c, const char *fn);
bool check(h2w_ctx *c, const char *fn) {
    if (!c) { set_error(std::string(fn) + ": null context"); return false; }
    return true;
}
int fail(h2w_ctx *c, const std::string &msg) { if (c) c->err = 1; set_error(msg); return -1; }
inline int kg_done(h2w_ctx *c, const char *fn) { if (c->keygen && c->mr.row != c->ncells) return fail(c, std::string(fn) + ": internal: keygen bookkeeping out of step with the cell stream"); return 0; }

TrRec::TrRec(h2w_ctx *c_, uint16_t code, uint16_t sub, uint64_t imm) : c(c_), t(c_ ? c_->trace : nullptr) {
    if (!t) return;
    op.code = code; op.sub = sub; op.imm = imm; op.n_in = op.n_out = 0; op.first_in = (uint32_t)t->ins.size(); op.first_out = (uint32_t)t->outs.size();
    op.cell0 = c->ncells; op.ncells = 0; op.tag = t->pending; t->pending = TraceTag();
}
uint32_t TrRec::ctx_id() const { return c->id; }
void TrRec::tag_required(const char *fn) { if (t && op.tag.kind == 0) t->fail(std::string(fn) + ": a witness whose value the library cannot recompute on another proof (tag it with h2w_trace_input, or use the level-2 hint ops)"); }
void TrRec::done() { if (!t) return; op.ncells = (uint32_t)(c->ncells - op.cell0); t->ops.push_back(op); }

}  // namespace

extern "C" {

int h2w_abi_version(void) { return H2W_ABI_VERSION; }
int h2w_poseidon_published(h2w_poseidon_consts_t *out) { if (!out) { set_error("h2w_poseidon_published: null output"); return -1; } *out = H2W_POSEIDON_PUBLISHED; return 0; }
const char *h2w_last_error(void) { return g_last_error.c_str(); }
int h2w_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }

h2w_ctx *h2w_ctx_new(int lookup_bits, int witness_gen_only, int device_id) {
    if (lookup_bits < 2 || lookup_bits > 28) { set_error("h2w_ctx_new: lookup_bits out of range [2,28]"); return nullptr; }
    h2w_ctx *c = new h2w_ctx(lookup_bits);
    c->L = lookup_bits; c->witness_gen_only = witness_gen_only; c->device = device_id; c->id = g_next_ctx_id++;
    c->P = fr_params_init(); { static const HostFr hf = host_fr_init(); c->H = hf; }
    c->keygen = witness_gen_only == 0; c->mr.L = lookup_bits;
    return c;
}
void h2w_ctx_free(h2w_ctx *c) {
    if (!c) return;
    DeviceGuard dg(c->d_out || c->d_meta || c->dt.slots ? c->device : -1);
    if (c->d_out) hipFree(c->d_out);
    if (c->d_meta) hipFree(c->d_meta);
    if (c->d_recs) hipFree(c->d_recs);
    if (c->d_pool) hipFree(c->d_pool);
    c->dt.free();
    delete c;
}
// The context as new, its host memory kept: the next proof's run appends into vectors that are already sized and mapped (a PoseidonBN254 proof streams 350 MB of
// literal cells through them; first-touch page faults and reallocation copies are a third of a level-1 call on a fresh context).  Handles of the old run are void.
int h2w_ctx_reset(h2w_ctx *c) {
    if (!check(c, "h2w_ctx_reset")) return -1;
    c->meta.clear(); c->recs.clear(); c->pool.clear(); c->ncells = 0; c->zero_set = false; c->zero_off = 0; c->err = 0;
    c->id = g_next_ctx_id++;                      // stale handles are recognisably another context's
    { MetaRecorder fresh; fresh.L = c->mr.L; c->mr = fresh; }
    c->nodes.assign(1, h2w_ctx::Node{-1, "all", 0, {}}); c->cur = 0; c->enter.clear();
    delete c->trace; c->trace = nullptr;
    return 0;
}
int h2w_ctx_footprint(const h2w_ctx *c, uint64_t out[2]) {
    if (!c || !out) { set_error("h2w_ctx_footprint: null argument"); return -1; }
    out[0] = c->recs.size(); out[1] = c->pool.size();
    return 0;
}
// A new context's vectors sized AND mapped ahead of a run (the pages are touched here, once each, instead of faulting one by one under the first run's appends)
int h2w_ctx_reserve(h2w_ctx *c, uint64_t n_records, uint64_t n_literal_cells) {
    if (!check(c, "h2w_ctx_reserve")) return -1;
    if (n_records > (1ull << 34) || n_literal_cells > (1ull << 34)) { set_error("h2w_ctx_reserve: more than 2^34 entries"); return -1; }
    auto map = [](auto &v, uint64_t n) {
        if (v.capacity() >= n) return;
        const size_t had = v.capacity();
        v.reserve((size_t)n);
        volatile char *b = reinterpret_cast<volatile char *>(v.data());
        const size_t lo = (had * sizeof(v[0])) & ~(size_t)4095, hi = v.capacity() * sizeof(v[0]);
        for (size_t i = lo < v.size() * sizeof(v[0]) ? v.size() * sizeof(v[0]) : lo; i < hi; i += 4096) b[i] = 0;      // (beyond size(): capacity the appends will fill)
    };
    map(c->meta, n_records); map(c->recs, n_records); map(c->pool, n_literal_cells);
    return 0;
}
uint64_t h2w_num_cells(const h2w_ctx *c) { return c ? c->ncells : 0; }
int h2w_ctx_error(const h2w_ctx *c) { return c ? c->err : 1; }

int h2w_load_constant(h2w_ctx *c, const h2w_fr_t *v, h2w_assigned_t *out) {
    if (!check(c, "h2w_load_constant")) return -1;
    TrRec tr(c, TR_LOAD_CONSTANT); if (tr.t) { tr.op.imm = tr.t->consts.size(); tr.t->consts.push_back(*v); }
    if (c->keygen) c->mr.load_constant(*v);
    cell(c, *v); *out = mk(c, *v, c->ncells - 1); tr.out(out); tr.done(); return kg_done(c, "h2w_load_constant");
}
int h2w_load_witness(h2w_ctx *c, const h2w_fr_t *v, h2w_assigned_t *out) {
    if (!check(c, "h2w_load_witness")) return -1;
    TrRec tr(c, TR_LOAD_WITNESS); tr.tag_required("h2w_load_witness");
    if (c->keygen) c->mr.load_witness();
    cell(c, *v); *out = mk(c, *v, c->ncells - 1); tr.out(out); tr.done(); return kg_done(c, "h2w_load_witness");
}
int h2w_load_zero(h2w_ctx *c, h2w_assigned_t *out) {
    if (!check(c, "h2w_load_zero")) return -1;
    if (!c->zero_set) {
        TrRec tr(c, TR_LOAD_CONSTANT); if (tr.t) { tr.op.imm = tr.t->consts.size(); tr.t->consts.push_back(fr_zero()); }
        if (c->keygen) c->mr.load_constant(fr_zero());
        cell(c, fr_zero()); c->zero_set = true; c->zero_off = c->ncells - 1;
        const h2w_assigned_t z = mk(c, fr_zero(), c->zero_off); tr.out(&z); tr.done();
    }
    *out = mk(c, fr_zero(), c->zero_off); return 0;
}
int h2w_load_constants(h2w_ctx *c, const h2w_fr_t *v, size_t n, h2w_assigned_t *out) {
    if (!check(c, "h2w_load_constants")) return -1;
    for (size_t i = 0; i < n; i++) {
        TrRec tr(c, TR_LOAD_CONSTANT); if (tr.t) { tr.op.imm = tr.t->consts.size(); tr.t->consts.push_back(v[i]); }
        if (c->keygen) c->mr.load_constant(v[i]);
        cell(c, v[i]); out[i] = mk(c, v[i], c->ncells - 1); tr.out(out + i); tr.done();
    }
    return kg_done(c, "h2w_load_constants");
}
int h2w_add(h2w_ctx *c, const h2w_assigned_t *a, const h2w_assigned_t *b, h2w_assigned_t *out) {
    if (!check(c, "h2w_add")) return -1;
    TrRec tr(c, TR_ADD); tr.in(a); tr.in(b);
    if (c->keygen) c->mr.add(MR::EX(off(a)), MR::EX(off(b)));
    *out = t_gate(c, a->value, b->value, fr_from_u64(1)); tr.out(out); tr.done(); return kg_done(c, "h2w_add");      // [a, b, 1, a+b]
}
int h2w_mul(h2w_ctx *c, const h2w_assigned_t *a, const h2w_assigned_t *b, h2w_assigned_t *out) {
    if (!check(c, "h2w_mul")) return -1;
    TrRec tr(c, TR_MUL); tr.in(a); tr.in(b);
    if (c->keygen) c->mr.mul(MR::EX(off(a)), MR::EX(off(b)));
    *out = t_gate(c, fr_zero(), a->value, b->value); tr.out(out); tr.done(); return kg_done(c, "h2w_mul");          // [0, a, b, a*b]
}
int h2w_mul_add(h2w_ctx *c, const h2w_assigned_t *a, const h2w_assigned_t *b, const h2w_assigned_t *cc, h2w_assigned_t *out) {
    if (!check(c, "h2w_mul_add")) return -1;
    TrRec tr(c, TR_MUL_ADD); tr.in(a); tr.in(b); tr.in(cc);
    if (c->keygen) c->mr.mul_add(MR::EX(off(a)), MR::EX(off(b)), MR::EX(off(cc)));
    *out = t_gate(c, cc->value, a->value, b->value); tr.out(out); tr.done(); return kg_done(c, "h2w_mul_add");          // [c, a, b, a*b+c]
}
int h2w_select(h2w_ctx *c, const h2w_assigned_t *a, const h2w_assigned_t *b, const h2w_assigned_t *sel, h2w_assigned_t *out) {
    if (!check(c, "h2w_select")) return -1;
    TrRec tr(c, TR_SELECT); tr.in(a); tr.in(b); tr.in(sel);
    if (c->keygen) c->mr.select(MR::EX(off(a)), MR::EX(off(b)), MR::EX(off(sel)));
    *out = t_select(c, a->value, b->value, sel->value); tr.out(out); tr.done(); return kg_done(c, "h2w_select");
}
int h2w_idx_to_indicator(h2w_ctx *c, const h2w_assigned_t *idx, size_t len, h2w_assigned_t *out) {
    if (!check(c, "h2w_idx_to_indicator")) return -1;
    TrRec tr(c, TR_IDX_TO_INDICATOR, 0, len); tr.in(idx);
    if (c->keygen) { std::vector<int64_t> o(len); c->mr.idx_to_indicator(off(idx), len, o.data()); }
    t_idx_to_indicator(c, idx->value, len, out); for (size_t i = 0; i < len; i++) tr.out(out + i); tr.done(); return kg_done(c, "h2w_idx_to_indicator");
}
int h2w_select_from_idx(h2w_ctx *c, const h2w_assigned_t *arr, size_t n, const h2w_assigned_t *idx, h2w_assigned_t *out) {
    if (!check(c, "h2w_select_from_idx")) return -1;
    std::vector<Av> ind(n);
    if (c->keygen) { std::vector<int64_t> o(n), ao(n); c->mr.idx_to_indicator(off(idx), n, o.data()); for (size_t i = 0; i < n; i++) ao[i] = off(arr + i); c->mr.select_by_indicator(ao.data(), 1, o.data(), n); }
    { TrRec tr(c, TR_IDX_TO_INDICATOR, 0, n); tr.in(idx); t_idx_to_indicator(c, idx->value, n, ind.data()); for (size_t i = 0; i < n; i++) tr.out(&ind[i]); tr.done(); }
    TrRec tr(c, TR_SELECT_BY_INDICATOR, 0, n); for (size_t i = 0; i < n; i++) tr.in(arr + i); for (size_t i = 0; i < n; i++) tr.in(&ind[i]);
    *out = t_select_by_indicator(c, arr, 1, ind.data(), n); tr.out(out); tr.done(); return kg_done(c, "h2w_select_from_idx");
}
int h2w_select_array_by_indicator(h2w_ctx *c, const h2w_assigned_t *arr2d, size_t len, size_t w, const h2w_assigned_t *ind, h2w_assigned_t *out) {
    if (!check(c, "h2w_select_array_by_indicator")) return -1;
    if (c->keygen) { std::vector<int64_t> ao(len * w), io(len); for (size_t i = 0; i < len * w; i++) ao[i] = off(arr2d + i); for (size_t i = 0; i < len; i++) io[i] = off(ind + i); for (size_t j = 0; j < w; j++) c->mr.select_by_indicator(ao.data() + j, w, io.data(), len); }
    for (size_t j = 0; j < w; j++) {
        TrRec tr(c, TR_SELECT_BY_INDICATOR, 0, len); for (size_t i = 0; i < len; i++) tr.in(arr2d + i * w + j); for (size_t i = 0; i < len; i++) tr.in(ind + i);
        out[j] = t_select_by_indicator(c, arr2d + j, w, ind, len); tr.out(out + j); tr.done();
    }
    return kg_done(c, "h2w_select_array_by_indicator");
}
int h2w_num_to_bits(h2w_ctx *c, const h2w_assigned_t *a, size_t range_bits, h2w_assigned_t *out) {
    if (!check(c, "h2w_num_to_bits")) return -1;
    if (range_bits == 0 || range_bits > 253) return fail(c, "h2w_num_to_bits: range_bits out of range");
    std::vector<fr_t> bits(range_bits), bases(range_bits);
    for (size_t i = 0; i < range_bits; i++) { bits[i] = fr_from_u64(fr_bits(a->value, (int)i, 1)); bases[i] = fr_pow2((int)i); }
    TrRec tr(c, TR_NUM_TO_BITS, 0, range_bits); tr.in(a);
    if (c->keygen) { std::vector<int64_t> o(range_bits); c->mr.num_to_bits(off(a), range_bits, o.data()); }
    uint64_t row = c->ncells;
    t_inner_product(c, bits.data(), bases.data(), range_bits);
    out[0] = mk(c, bits[0], row);
    for (size_t i = 1; i < range_bits; i++) out[i] = mk(c, bits[i], row + 1 + 3 * (i - 1));
    for (size_t i = 0; i < range_bits; i++) t_assert_bit(c, bits[i]);
    for (size_t i = 0; i < range_bits; i++) tr.out(out + i);
    tr.done(); return kg_done(c, "h2w_num_to_bits");
}
int h2w_bits_to_num(h2w_ctx *c, const h2w_assigned_t *bits, size_t n, h2w_assigned_t *out) {
    if (!check(c, "h2w_bits_to_num")) return -1;
    std::vector<fr_t> a(n), b(n);
    for (size_t i = 0; i < n; i++) { a[i] = bits[i].value; b[i] = fr_pow2((int)i); }
    TrRec tr(c, TR_BITS_TO_NUM, 0, n); for (size_t i = 0; i < n; i++) tr.in(bits + i);
    if (c->keygen) { std::vector<int64_t> o(n); for (size_t i = 0; i < n; i++) o[i] = off(bits + i); c->mr.bits_or_limbs_to_num(o.data(), n, 1); }
    *out = t_inner_product(c, a.data(), b.data(), n); tr.out(out); tr.done(); return kg_done(c, "h2w_bits_to_num");
}
int h2w_decompose_le(h2w_ctx *c, const h2w_assigned_t *num, size_t limb_bits, size_t num_limbs, h2w_assigned_t *out) {
    if (!check(c, "h2w_decompose_le")) return -1;
    if (limb_bits == 0 || limb_bits > 64) return fail(c, "h2w_decompose_le: limb_bits out of range");
    std::vector<fr_t> limbs(num_limbs), bases(num_limbs);
    for (size_t i = 0; i < num_limbs; i++) { limbs[i] = fr_from_u64(fr_bits(num->value, (int)(i * limb_bits), (int)limb_bits)); bases[i] = fr_pow2((int)(i * limb_bits)); }
    TrRec tr(c, TR_DECOMPOSE_LE, 0, ((uint64_t)limb_bits << 32) | (uint64_t)num_limbs); tr.in(num);
    if (c->keygen) { std::vector<int64_t> o(num_limbs); c->mr.decompose_le(off(num), limb_bits, num_limbs, o.data()); }
    uint64_t row = c->ncells;
    t_inner_product(c, limbs.data(), bases.data(), num_limbs);
    out[0] = mk(c, limbs[0], row);
    for (size_t i = 0; i + 1 < num_limbs; i++) out[i + 1] = mk(c, limbs[i + 1], row + 1 + 3 * i);
    for (size_t i = 0; i < num_limbs; i++) t_range_check(c, out[i], limb_bits);
    for (size_t i = 0; i < num_limbs; i++) tr.out(out + i);
    tr.done(); return kg_done(c, "h2w_decompose_le");
}
int h2w_limbs_to_num(h2w_ctx *c, const h2w_assigned_t *limbs, size_t n, size_t limb_bits, h2w_assigned_t *out) {
    if (!check(c, "h2w_limbs_to_num")) return -1;
    std::vector<fr_t> a(n), b(n);
    for (size_t i = 0; i < n; i++) { a[i] = limbs[i].value; b[i] = fr_pow2((int)(i * limb_bits)); }
    TrRec tr(c, TR_LIMBS_TO_NUM, 0, limb_bits); for (size_t i = 0; i < n; i++) tr.in(limbs + i);
    if (c->keygen) { std::vector<int64_t> o(n); for (size_t i = 0; i < n; i++) o[i] = off(limbs + i); c->mr.bits_or_limbs_to_num(o.data(), n, (int)limb_bits); }
    *out = t_inner_product(c, a.data(), b.data(), n); tr.out(out); tr.done(); return kg_done(c, "h2w_limbs_to_num");
}
int h2w_check_less_than_safe(h2w_ctx *c, const h2w_assigned_t *a, uint64_t b) { if (!check(c, "h2w_check_less_than_safe")) return -1; TrRec tr(c, TR_CLT_SAFE, 0, b); tr.in(a); if (c->keygen) c->mr.check_less_than_safe(off(a), b); t_check_less_than_safe(c, *a, b); tr.done(); return kg_done(c, "h2w_check_less_than_safe"); }
int h2w_range_check(h2w_ctx *c, const h2w_assigned_t *a, size_t range_bits) { if (!check(c, "h2w_range_check")) return -1; TrRec tr(c, TR_RANGE_CHECK, 0, range_bits); tr.in(a); if (c->keygen) c->mr.range_check(off(a), range_bits); t_range_check(c, *a, range_bits); tr.done(); return kg_done(c, "h2w_range_check"); }
int h2w_constrain_equal(h2w_ctx *c, const h2w_assigned_t *a, const h2w_assigned_t *b) { if (!check(c, "h2w_constrain_equal")) return -1; if (c->keygen) c->mr.equal(off(a), off(b)); return 0; }

// ---------------------------------------------------------------- ContextWrapper::{push_context, pop_context} (util/context_wrapper.rs:28-34)
int h2w_push_context(h2w_ctx *c, const char *name) {
    if (!check(c, "h2w_push_context") || !name) return -1;
    if (c->trace) { TrRec tr(c, TR_SCOPE_PUSH, 0, c->trace->names.size()); c->trace->names.push_back(name); tr.done(); }
    int ch = -1;
    for (int k : c->nodes[c->cur].children) if (c->nodes[k].name == name) { ch = k; break; }
    if (ch < 0) { ch = (int)c->nodes.size(); c->nodes.push_back(h2w_ctx::Node{c->cur, name, 0, {}}); c->nodes[c->cur].children.push_back(ch); }
    c->enter.push_back(c->ncells); c->cur = ch; return 0;
}
int h2w_pop_context(h2w_ctx *c) {
    if (!check(c, "h2w_pop_context")) return -1;
    if (c->enter.empty()) return fail(c, "h2w_pop_context: no open context");
    if (c->trace) { TrRec tr(c, TR_SCOPE_POP); tr.done(); }
    c->nodes[c->cur].cells += c->ncells - c->enter.back(); c->enter.pop_back(); c->cur = c->nodes[c->cur].parent; return 0;
}
// collapsed-stack dump "a;b;c <inclusive cells>\n" (util/context_tree.rs:132-152 writes own counts; inclusive is what the SVG frames show)
size_t h2w_context_dump(const h2w_ctx *c, char *buf, size_t cap) {
    if (!c) return 0;
    std::string out;
    std::vector<std::string> path(c->nodes.size());
    for (size_t i = 0; i < c->nodes.size(); i++) {
        path[i] = c->nodes[i].parent < 0 ? c->nodes[i].name : path[(size_t)c->nodes[i].parent] + ";" + c->nodes[i].name;
        out += path[i] + " " + std::to_string(i == 0 ? c->ncells : c->nodes[i].cells) + "\n";
    }
    if (buf && cap) { size_t n = out.size() < cap - 1 ? out.size() : cap - 1; memcpy(buf, out.data(), n); buf[n] = 0; }
    return out.size();
}

// ---------------------------------------------------------------- fused Goldilocks level (field/goldilocks/base.rs)
static inline bool gl_canon(const fr_t &v) { return fits64(v) && v.l[0] < GL_P; }
int h2w_gl_load_constant(h2w_ctx *c, uint64_t a, h2w_assigned_t *out) { fr_t v = fr_from_u64(a); return h2w_load_constant(c, &v, out); }
int h2w_gl_load_witness(h2w_ctx *c, uint64_t a, h2w_assigned_t *out) {
    if (!check(c, "h2w_gl_load_witness")) return -1;
    TrRec tr(c, TR_GL_WITNESS); tr.tag_required("h2w_gl_load_witness");
    if (c->keygen) c->mr.gl_load_witness();
    uint64_t off = c->ncells; rec(c, T_LOADW, a, 0, 0, 0);
    *out = mk(c, fr_from_u64(a), off); tr.out(out); tr.done(); return kg_done(c, "h2w_gl_load_witness");
}
// record for the 61-cell reduce tail of an arbitrary (< 2^128) value; returns remainder wire
static int gl_reduce_impl(h2w_ctx *c, const fr_t &v, h2w_assigned_t *out) {
    if (v.l[2] | v.l[3]) return fail(c, "h2w_gl_reduce: value >= 2^128 (reference only supports up to p*(p-1), base.rs:345)");
    u128 V = ((u128)v.l[1] << 64) | v.l[0];
    uint64_t r = gl_reduce128(V);
    uint64_t off = c->ncells; rec(c, T_REDUCE, v.l[0], v.l[1], 0, 0);
    int nl = c->tt.ncells(T_LOADW);
    *out = mk(c, fr_from_u64(r), off + (uint64_t)nl);    // remainder = second load_witness cell
    return 0;
}
int h2w_gl_reduce(h2w_ctx *c, const h2w_assigned_t *a, h2w_assigned_t *out) { if (!check(c, "h2w_gl_reduce")) return -1; TrRec tr(c, TR_GL_REDUCE); tr.in(a); if (c->keygen && !(a->value.l[2] | a->value.l[3])) c->mr.gl_reduce(off(a)); if (gl_reduce_impl(c, a->value, out) != 0) return -1; tr.out(out); tr.done(); return kg_done(c, "h2w_gl_reduce"); }
static int glop(h2w_ctx *c, int t, uint64_t A, uint64_t B, uint64_t C, h2w_assigned_t *out) {
    uint64_t r = gl_reduce128((u128)A * B + C);
    uint64_t off = c->ncells; rec(c, t, A, B, C, 0);
    int pre = (t == T_GLOP) ? 0 : 1, nl = c->tt.ncells(T_LOADW);
    *out = mk(c, fr_from_u64(r), off + (uint64_t)(pre + 4 + nl));
    return 0;
}
#define GL_ARGS2(fn) if (!check(c, fn)) return -1; if (!fits64(a->value) || !fits64(b->value)) return fail(c, std::string(fn) + ": operand is not a 64-bit Goldilocks wire")
int h2w_gl_add(h2w_ctx *c, const h2w_assigned_t *a, const h2w_assigned_t *b, h2w_assigned_t *out) { GL_ARGS2("h2w_gl_add"); TrRec tr(c, TR_GLOP, T_GLOP); tr.in(b); tr.lit(1); tr.in(a); if (c->keygen) c->mr.gl_reduce(c->mr.add(MR::EX(off(a)), MR::EX(off(b)))); glop(c, T_GLOP, b->value.l[0], 1, a->value.l[0], out); tr.out(out); tr.done(); return kg_done(c, "h2w_gl_add"); }
int h2w_gl_mul(h2w_ctx *c, const h2w_assigned_t *a, const h2w_assigned_t *b, h2w_assigned_t *out) { GL_ARGS2("h2w_gl_mul"); TrRec tr(c, TR_GLOP, T_GLOP); tr.in(a); tr.in(b); tr.lit(0); if (c->keygen) c->mr.gl_reduce(c->mr.mul(MR::EX(off(a)), MR::EX(off(b)))); glop(c, T_GLOP, a->value.l[0], b->value.l[0], 0, out); tr.out(out); tr.done(); return kg_done(c, "h2w_gl_mul"); }
int h2w_gl_sub(h2w_ctx *c, const h2w_assigned_t *a, const h2w_assigned_t *b, h2w_assigned_t *out) {
    GL_ARGS2("h2w_gl_sub");
    TrRec tr(c, TR_GLOP, T_KB_GLOP); tr.in(b); tr.lit(GL_NEG_ONE); tr.in(a);
    if (c->keygen) { const int64_t m1 = c->mr.load_constant(fr_from_u64(GL_NEG_ONE)); c->mr.gl_reduce(c->mr.mul_add(MR::EX(off(b)), MR::EX(m1), MR::EX(off(a)))); }
    glop(c, T_KB_GLOP, b->value.l[0], GL_NEG_ONE, a->value.l[0], out); tr.out(out); tr.done(); return kg_done(c, "h2w_gl_sub");
}
int h2w_gl_mul_add(h2w_ctx *c, const h2w_assigned_t *a, const h2w_assigned_t *b, const h2w_assigned_t *cc, h2w_assigned_t *out) {
    GL_ARGS2("h2w_gl_mul_add"); if (!fits64(cc->value)) return fail(c, "h2w_gl_mul_add: operand is not a 64-bit Goldilocks wire");
    TrRec tr(c, TR_GLOP, T_GLOP); tr.in(a); tr.in(b); tr.in(cc);
    if (c->keygen) c->mr.gl_reduce(c->mr.mul_add(MR::EX(off(a)), MR::EX(off(b)), MR::EX(off(cc))));
    glop(c, T_GLOP, a->value.l[0], b->value.l[0], cc->value.l[0], out); tr.out(out); tr.done(); return kg_done(c, "h2w_gl_mul_add");
}
int h2w_gl_div(h2w_ctx *c, const h2w_assigned_t *a, const h2w_assigned_t *b, h2w_assigned_t *out) {
    GL_ARGS2("h2w_gl_div");
    if (!gl_canon(a->value) || !gl_canon(b->value)) return fail(c, "h2w_gl_div: non-canonical Goldilocks wire");
    if (b->value.l[0] == 0) return fail(c, "h2w_gl_div: division by zero (reference asserts, base.rs:379)");
    uint64_t res = gl_mul(a->value.l[0], gl_inv(b->value.l[0]));
    h2w_assigned_t rw, prod;
    if (c->trace) { if (!a->has_cell || !b->has_cell) c->trace->fail("h2w_gl_div: an operand is not a cell"); c->trace->pending = TraceTag{2, 0, 0, a->offset, b->offset}; }      // the hint a / b is the library's (base.rs:382)
    h2w_gl_load_witness(c, res, &rw);
    TrRec tr(c, TR_GLOP, T_GLOP); tr.in(b); tr.in(&rw); tr.lit(0);
    if (c->keygen) { const int64_t pr = c->mr.gl_reduce(c->mr.mul(MR::EX(off(b)), MR::EX(off(&rw)))); c->mr.equal(off(a), pr); }      // gl.assert_equal(a, b * res)
    glop(c, T_GLOP, b->value.l[0], res, 0, &prod); tr.out(&prod); tr.done();
    *out = rw; return kg_done(c, "h2w_gl_div");
}
// GoldilocksChip::mul_sub (base.rs:332-343): mul_no_reduce, sub_no_reduce (= prod + c*(p-1) with a NEG_ONE constant cell), reduce: 70 cells
int h2w_gl_mul_sub(h2w_ctx *c, const h2w_assigned_t *a, const h2w_assigned_t *b, const h2w_assigned_t *cc, h2w_assigned_t *out) {
    GL_ARGS2("h2w_gl_mul_sub"); if (!fits64(cc->value)) return fail(c, "h2w_gl_mul_sub: operand is not a 64-bit Goldilocks wire");
    {   // a*b + c*(p-1) must stay below 2^128 (the reference's reduce is only specified up to p*(p-1), base.rs:345); checked before any cell is emitted
        const u128 ab = (u128)a->value.l[0] * b->value.l[0], cm = (u128)cc->value.l[0] * GL_NEG_ONE;
        if (ab + cm < ab) return fail(c, "h2w_gl_mul_sub: a*b + c*(p-1) >= 2^128 (outside the range of GoldilocksChip::reduce, base.rs:345)");
    }
    h2w_assigned_t prod, neg_one, diff; const h2w_fr_t m1 = fr_from_u64(GL_NEG_ONE);
    if (h2w_mul(c, a, b, &prod) != 0 || h2w_load_constant(c, &m1, &neg_one) != 0 || h2w_mul_add(c, cc, &neg_one, &prod, &diff) != 0) return -1;
    TrRec tr(c, TR_GL_REDUCE); tr.in(&diff);
    if (c->keygen) c->mr.gl_reduce(off(&diff));
    if (gl_reduce_impl(c, diff.value, out) != 0) return -1;
    tr.out(out); tr.done();
    return kg_done(c, "h2w_gl_mul_sub");
}
// GoldilocksChip::neg (base.rs:234-238): load_neg_one, mul
int h2w_gl_neg(h2w_ctx *c, const h2w_assigned_t *a, h2w_assigned_t *out) {
    if (!check(c, "h2w_gl_neg")) return -1;
    h2w_assigned_t neg_one; if (h2w_gl_load_constant(c, GL_NEG_ONE, &neg_one) != 0) return -1;
    return h2w_gl_mul(c, a, &neg_one, out);
}
// GoldilocksChip::square (base.rs:401-404) and exp_power_of_2 (base.rs:433-445)
int h2w_gl_square(h2w_ctx *c, const h2w_assigned_t *a, h2w_assigned_t *out) { return h2w_gl_mul(c, a, a, out); }
int h2w_gl_exp_power_of_2(h2w_ctx *c, const h2w_assigned_t *base, size_t power_log, h2w_assigned_t *out) {
    if (!check(c, "h2w_gl_exp_power_of_2")) return -1;
    h2w_assigned_t p = *base;
    for (size_t i = 0; i < power_log; i++) { h2w_assigned_t q; if (h2w_gl_mul(c, &p, &p, &q) != 0) return -1; p = q; }
    *out = p; return 0;
}
int h2w_gl_inv(h2w_ctx *c, const h2w_assigned_t *a, h2w_assigned_t *out) {
    if (!check(c, "h2w_gl_inv")) return -1;
    h2w_assigned_t one; h2w_gl_load_constant(c, 1, &one);
    return h2w_gl_div(c, &one, a, out);
}

// GoldilocksQuadExtChip::inv's hint (extension.rs:320-340: `let inverse = a.value().inverse()` then load_witness of its two components): the inverse
// is computed here, so that a traced run (trace.h) can recompute it for another proof.  The caller goes on with mul(a, out) and its assert_equal.
int h2w_gl_ext_inv_witness(h2w_ctx *c, const h2w_assigned_t a[2], h2w_assigned_t out[2]) {
    if (!check(c, "h2w_gl_ext_inv_witness")) return -1;
    if (!gl_canon(a[0].value) || !gl_canon(a[1].value)) return fail(c, "h2w_gl_ext_inv_witness: non-canonical Goldilocks wire");
    gle_t av; av.c[0] = a[0].value.l[0]; av.c[1] = a[1].value.l[0];
    if (av.c[0] == 0 && av.c[1] == 0) return fail(c, "h2w_gl_ext_inv_witness: inverse of zero (reference panics, extension.rs:327)");
    const gle_t iv = gle_inv(av);
    for (int k = 0; k < 2; k++) {
        if (c->trace) { if (!a[0].has_cell || !a[1].has_cell) c->trace->fail("h2w_gl_ext_inv_witness: an operand is not a cell"); c->trace->pending = TraceTag{3 + k, 0, 0, a[0].offset, a[1].offset}; }
        if (h2w_gl_load_witness(c, iv.c[k], out + k) != 0) return -1;
    }
    return 0;
}
// ---- trace mode (trace.h)
int h2w_ctx_trace_begin(h2w_ctx *c) {
    if (!check(c, "h2w_ctx_trace_begin")) return -1;
    if (c->ncells != 0 || c->trace) return fail(c, "h2w_ctx_trace_begin: the context must be fresh");
    c->trace = new Trace(); return 0;
}
int h2w_trace_input(h2w_ctx *c, uint64_t word, uint32_t n_words) {
    if (!check(c, "h2w_trace_input")) return -1;
    if (!c->trace) return 0;                       // not tracing: nothing to tag
    if (n_words != 1 && n_words != 4) return fail(c, "h2w_trace_input: a proof value is 1 word (a Goldilocks element) or 4 (a BN254 hash)");
    c->trace->pending = TraceTag{1, word, n_words, 0, 0}; return 0;
}

// ---------------------------------------------------------------- keygen-side metadata of the eager context (witness_gen_only == 0)
static int kg_check(h2w_ctx *c, const char *fn) { if (!check(c, fn)) return -1; if (!c->keygen) return fail(c, std::string(fn) + ": the context was created with witness_gen_only != 0"); return 0; }
uint64_t h2w_ctx_num_gates(h2w_ctx *c) { return c && c->keygen ? c->mr.sel.size() : 0; }
uint64_t h2w_ctx_num_lookups(h2w_ctx *c) { return c && c->keygen ? c->mr.lookups.size() : 0; }
uint64_t h2w_ctx_num_equalities(h2w_ctx *c) { return c && c->keygen ? c->mr.eq.size() / 2 : 0; }
uint64_t h2w_ctx_num_const_equalities(h2w_ctx *c) { return c && c->keygen ? c->mr.ceq_cell.size() : 0; }
int h2w_ctx_gate_cells(h2w_ctx *c, uint64_t *cells) { if (kg_check(c, "h2w_ctx_gate_cells") != 0 || !cells) return -1; memcpy(cells, c->mr.sel.data(), c->mr.sel.size() * 8); return 0; }
int h2w_ctx_lookup_cells(h2w_ctx *c, uint64_t *cells) { if (kg_check(c, "h2w_ctx_lookup_cells") != 0 || !cells) return -1; memcpy(cells, c->mr.lookups.data(), c->mr.lookups.size() * 8); return 0; }
int h2w_ctx_equalities(h2w_ctx *c, uint64_t *pairs) { if (kg_check(c, "h2w_ctx_equalities") != 0 || !pairs) return -1; memcpy(pairs, c->mr.eq.data(), c->mr.eq.size() * 8); return 0; }
int h2w_ctx_const_equalities(h2w_ctx *c, uint64_t *cells, h2w_fr_t *values) {
    if (kg_check(c, "h2w_ctx_const_equalities") != 0 || !cells || !values) return -1;
    memcpy(cells, c->mr.ceq_cell.data(), c->mr.ceq_cell.size() * 8); memcpy(values, c->mr.ceq_val.data(), c->mr.ceq_val.size() * sizeof(fr_t)); return 0;
}

// ---------------------------------------------------------------- advice hand-off: GPU expansion
static int ensure_expanded(h2w_ctx *c) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(c, "h2w: no HIP device visible — the advice stream is only produced on the GPU (no CPU fallback)");
    if (c->device >= ndev) return fail(c, "h2w: device_id out of range");
    DeviceGuard dg(c->device);
    if (c->d_out) { hipFree(c->d_out); c->d_out = nullptr; }
    if (c->d_meta) { hipFree(c->d_meta); c->d_meta = nullptr; }
    if (c->d_recs) { hipFree(c->d_recs); c->d_recs = nullptr; }
    if (c->d_pool) { hipFree(c->d_pool); c->d_pool = nullptr; }
    if (c->ncells == 0) return 0;
    if (!c->dt_ready || c->dt_nslots != c->tt.slots.size() || c->dt_nconsts != c->tt.consts.size() || c->dt_ntmpl != c->tt.info.size()) {
        if (c->dt.upload(c->tt) != 0) return fail(c, "h2w: template upload failed: " + g_last_error);
        c->dt_ready = true; c->dt_nslots = c->tt.slots.size(); c->dt_nconsts = c->tt.consts.size(); c->dt_ntmpl = c->tt.info.size();
    }
    size_t nrec = c->recs.size();
    H2W_HIP(hipMalloc(&c->d_out, c->ncells * sizeof(fr_t)));
    H2W_HIP(hipMalloc(&c->d_meta, nrec * sizeof(uint64_t)));
    H2W_HIP(hipMalloc(&c->d_recs, nrec * sizeof(rec_t)));
    H2W_HIP(hipMemcpy(c->d_meta, c->meta.data(), nrec * sizeof(uint64_t), hipMemcpyHostToDevice));
    H2W_HIP(hipMemcpy(c->d_recs, c->recs.data(), nrec * sizeof(rec_t), hipMemcpyHostToDevice));
    if (!c->pool.empty()) {
        H2W_HIP(hipMalloc(&c->d_pool, c->pool.size() * sizeof(fr_t)));
        H2W_HIP(hipMemcpy(c->d_pool, c->pool.data(), c->pool.size() * sizeof(fr_t), hipMemcpyHostToDevice));
    }
    ExpandArgs A;
    A.meta = (const uint64_t *)c->d_meta; A.recs = (const rec_t *)c->d_recs; A.nrec = nrec; A.rec_stride = nrec;
    A.out = (fr_t *)c->d_out; A.cell_stride = c->ncells; A.pool = (const fr_t *)c->d_pool;
    c->dt.fill(A); A.rb = c->tt.rb;
    A.tile_ctr = nullptr; A.cm.starts = nullptr; A.cm.ncols = 0; A.cm.k = 0; expand_unsharded(A);
    if (launch_expand(A, 1, 2048, nullptr) != 0) return -1;
    H2W_HIP(hipGetLastError());
    H2W_HIP(hipDeviceSynchronize());
    return 0;
}
int h2w_ctx_advice_device(h2w_ctx *c, void **dev_ptr) {
    if (!check(c, "h2w_ctx_advice_device")) return -1;
    if (ensure_expanded(c) != 0) return -1;
    *dev_ptr = c->d_out; return 0;
}
int h2w_ctx_download(h2w_ctx *c, uint64_t first, uint64_t count, h2w_fr_t *host_dst) {
    if (!check(c, "h2w_ctx_download")) return -1;
    if (first + count > c->ncells) return fail(c, "h2w_ctx_download: range out of bounds");
    if (ensure_expanded(c) != 0) return -1;
    if (count == 0) return 0;
    DeviceGuard dg(c->device);
    H2W_HIP(hipMemcpy(host_dst, (const fr_t *)c->d_out + first, count * sizeof(fr_t), hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"

namespace h2w {
Trace *ctx_trace(h2w_ctx *c) { return c ? c->trace : nullptr; }
int ctx_lookup_bits(const h2w_ctx *c) { return c->L; }
uint64_t ctx_num_cells(const h2w_ctx *c) { return c->ncells; }
}
