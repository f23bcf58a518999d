/*
 * oracle.c — CPU restatement of the reference's FRI-verifier witness generation.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Citations are to /root/reference/verifier/src/...
 * halo2-base templates follow SURVEY.md Appendix A (halo2-lib `community-edition`, not on this box).
 */
#include "oracle.h"
#include "oracle_field.h"
#include <stdio.h>
#include <stdlib.h>
#include <assert.h>

/* ===================================================================== context */
typedef struct { int64_t a, b; int kind; } oeq_t;
typedef struct { int64_t cell; int kind; } olk_t;        /* kind 0 = internal, 1 = semantic (chip assert) */
typedef struct { int64_t cell; ofr_t c; } oceq_t;
typedef struct { int parent; int name; uint64_t cells; int first_child, next_sibling; } onode_t;

struct octx {
    ofr_t *advice; size_t n, cap;
    int lookup_bits, witness_gen_only, track;
    uint8_t *selector; size_t selcap;
    oeq_t *eq; size_t neq, capeq;
    oceq_t *ceq; size_t nceq, capceq;
    olk_t *lookup; size_t nlookup, caplookup;
    int64_t zero_cell;
    /* scope tree (util/context_tree.rs) */
    onode_t *nodes; int nnodes, capnodes; int cur;
    uint64_t *enter; int depth, capdepth;
    char **names; int nnames, capnames;
    char err[256]; int failed;
    int semantic; /* >0: constraints registered now are protocol checks on the (possibly invalid) proof */
    int streaming; uint64_t digest[4]; /* orc_ctx_new_streaming: only the last ORC_RING cells are kept, the stream is summed into h2w_advice_digest's checksum */
    uint64_t watch; int has_watch; char watch_path[2048]; uint64_t watch_enter; /* orc_watch_cell: the #[count] call stack that pushes one given cell */
};

enum { QC_EXISTING = 0, QC_WITNESS = 1, QC_CONSTANT = 2 };
typedef struct { int kind; ofr_t v; int64_t cell; } qc_t;

static qc_t Q_EX(oav_t a) { qc_t q; q.kind = QC_EXISTING; q.v = a.v; q.cell = a.cell; return q; }
static qc_t Q_W(ofr_t v) { qc_t q; q.kind = QC_WITNESS; q.v = v; q.cell = -1; return q; }
static qc_t Q_C(ofr_t v) { qc_t q; q.kind = QC_CONSTANT; q.v = v; q.cell = -1; return q; }
static qc_t Q_CU(uint64_t v) { return Q_C(fr_from_u64(v)); }

octx_t *orc_ctx_new(int lookup_bits, int witness_gen_only, int track_scopes) {
    fr_init();
    octx_t *c = (octx_t *)calloc(1, sizeof(octx_t));
    c->lookup_bits = lookup_bits; c->witness_gen_only = witness_gen_only; c->track = track_scopes;
    c->cap = 1 << 16; c->advice = (ofr_t *)malloc(c->cap * sizeof(ofr_t));
    c->zero_cell = -1;
    if (track_scopes) {
        c->capnodes = 1024; c->nodes = (onode_t *)calloc(c->capnodes, sizeof(onode_t));
        c->nodes[0].parent = -1; c->nodes[0].name = -1; c->nodes[0].first_child = -1; c->nodes[0].next_sibling = -1;
        c->nnodes = 1; c->cur = 0;
        c->capdepth = 256; c->enter = (uint64_t *)calloc(c->capdepth, sizeof(uint64_t));
        c->capnames = 256; c->names = (char **)calloc(c->capnames, sizeof(char *));
    }
    return c;
}
/* A context for streams too long for the host (cfg 3 / cfg 5 with Goldilocks-Poseidon caps: 14 / 43 GB): the cells go into a ring of the last
 * ORC_RING (every read-back of the templates is a few dozen cells deep: ctx_get) and into the position-dependent checksum that
 * include/h2w.h's h2w_advice_digest computes on the device: digest[j] = sum_i limb_j(cell i) * ((((i + 1) * 0x9E3779B97F4A7C15) | 1) + 2 j) mod 2^64. */
#define ORC_RING ((size_t)1 << 16)
octx_t *orc_ctx_new_streaming(int lookup_bits) {
    octx_t *c = orc_ctx_new(lookup_bits, 1, 0);
    c->streaming = 1; c->cap = ORC_RING; c->advice = (ofr_t *)realloc(c->advice, c->cap * sizeof(ofr_t));
    return c;
}
void orc_digest(const octx_t *c, uint64_t out[4]) { for (int j = 0; j < 4; j++) out[j] = c->digest[j]; }
void orc_ctx_free(octx_t *c) {
    if (!c) return;
    free(c->advice); free(c->selector); free(c->eq); free(c->ceq); free(c->lookup);
    free(c->nodes); free(c->enter);
    for (int i = 0; i < c->nnames; i++) free(c->names[i]);
    free(c->names); free(c);
}
void orc_ctx_reserve(octx_t *c, uint64_t ncells) { if (!c->streaming && ncells > c->cap) { c->cap = ncells; c->advice = (ofr_t *)realloc(c->advice, c->cap * sizeof(ofr_t)); if (!c->advice) { fprintf(stderr, "oracle: OOM\n"); abort(); } } }
uint64_t orc_num_cells(const octx_t *c) { return c->n; } /* util/context_wrapper.rs:24-26 */
const ofr_t *orc_advice(const octx_t *c) { return c->advice; }
const char *orc_error(const octx_t *c) { return c->failed ? c->err : ""; }
static void ctx_fail(octx_t *c, const char *msg) { if (!c->failed) { c->failed = 1; snprintf(c->err, sizeof(c->err), "%s", msg); } }

/* #[count] (macro/src/lib.rs:9-61): push_context(fn_name) ... pop_context() */
static void sc_push(octx_t *c, const char *name) {
    int id = -1;
    for (int i = 0; i < c->nnames; i++) if (strcmp(c->names[i], name) == 0) { id = i; break; }
    if (id < 0) {
        if (c->nnames == c->capnames) { c->capnames *= 2; c->names = (char **)realloc(c->names, c->capnames * sizeof(char *)); }
        id = c->nnames; c->names[c->nnames++] = strdup(name);
    }
    int ch = c->nodes[c->cur].first_child;
    while (ch >= 0 && c->nodes[ch].name != id) ch = c->nodes[ch].next_sibling;
    if (ch < 0) {
        if (c->nnodes == c->capnodes) { c->capnodes *= 2; c->nodes = (onode_t *)realloc(c->nodes, c->capnodes * sizeof(onode_t)); }
        ch = c->nnodes++;
        c->nodes[ch].parent = c->cur; c->nodes[ch].name = id; c->nodes[ch].cells = 0; c->nodes[ch].first_child = -1;
        c->nodes[ch].next_sibling = c->nodes[c->cur].first_child; c->nodes[c->cur].first_child = ch;
    }
    if (c->depth == c->capdepth) { c->capdepth *= 2; c->enter = (uint64_t *)realloc(c->enter, c->capdepth * sizeof(uint64_t)); }
    c->enter[c->depth++] = c->n; c->cur = ch;
}
static void sc_pop(octx_t *c) {
    c->nodes[c->cur].cells += c->n - c->enter[--c->depth];
    c->cur = c->nodes[c->cur].parent;
}
#define SC(name) do { if (c->track) sc_push(c, name); } while (0)
#define EC() do { if (c->track) sc_pop(c); } while (0)

static size_t scope_path(const octx_t *c, int node, char *buf, size_t cap) {
    if (node == 0) { return (size_t)snprintf(buf, cap, "all"); }
    size_t k = scope_path(c, c->nodes[node].parent, buf, cap);
    return k + (size_t)snprintf(buf + (k < cap ? k : cap), k < cap ? cap - k : 0, ";%s", c->names[c->nodes[node].name]);
}
/* Before a run on a context with track_scopes: remember which #[count] call stack appends cell `cell` (tools/compare_advice.py: names the
 * chip function - and so the halo2-base template - in which two advice streams first differ).  orc_watch_path: that stack ("all;verify_proof;...")
 * and the offset of the cell inside the innermost call's block. */
void orc_watch_cell(octx_t *c, uint64_t cell) { c->watch = cell; c->has_watch = 1; c->watch_path[0] = 0; }
const char *orc_watch_path(const octx_t *c, uint64_t *offset_in_call) { if (offset_in_call) *offset_in_call = c->watch - c->watch_enter; return c->watch_path; }
size_t orc_scope_dump(const octx_t *c, char *buf, size_t cap) {
    size_t used = 0; char tmp[2048];
    if (!c->track) return 0;
    for (int i = 0; i < c->nnodes; i++) {
        scope_path(c, i, tmp, sizeof(tmp));
        uint64_t cells = i == 0 ? c->n : c->nodes[i].cells;
        int k = snprintf(buf ? buf + (used < cap ? used : cap) : NULL, buf && used < cap ? cap - used : 0, "%s %llu\n", tmp, (unsigned long long)cells);
        used += (size_t)k;
    }
    return used;
}

static size_t scope_path(const octx_t *c, int node, char *buf, size_t cap);
static inline void adv_push(octx_t *c, const ofr_t *v) {
    if (c->has_watch && c->n == c->watch && c->track) { scope_path(c, c->cur, c->watch_path, sizeof(c->watch_path)); c->watch_enter = c->depth ? c->enter[c->depth - 1] : 0; }
    if (c->streaming) {
        const uint64_t m = (((uint64_t)c->n + 1) * 0x9E3779B97F4A7C15ULL) | 1ULL;
        for (int j = 0; j < 4; j++) c->digest[j] += v->l[j] * (m + 2 * (uint64_t)j);
        c->advice[c->n++ & (ORC_RING - 1)] = *v;
        return;
    }
    if (c->n == c->cap) { c->cap *= 2; c->advice = (ofr_t *)realloc(c->advice, c->cap * sizeof(ofr_t)); if (!c->advice) { fprintf(stderr, "oracle: OOM\n"); abort(); } }
    c->advice[c->n++] = *v;
}
static void add_eq(octx_t *c, int64_t a, int64_t b, int kind) {
    if (c->witness_gen_only) return;
    if (c->neq == c->capeq) { c->capeq = c->capeq ? c->capeq * 2 : 1024; c->eq = (oeq_t *)realloc(c->eq, c->capeq * sizeof(oeq_t)); }
    c->eq[c->neq].a = a; c->eq[c->neq].b = b; c->eq[c->neq].kind = c->semantic ? 1 : kind; c->neq++;
}
static void add_ceq(octx_t *c, int64_t cell, const ofr_t *v) {
    if (c->witness_gen_only) return;
    if (c->nceq == c->capceq) { c->capceq = c->capceq ? c->capceq * 2 : 1024; c->ceq = (oceq_t *)realloc(c->ceq, c->capceq * sizeof(oceq_t)); }
    c->ceq[c->nceq].cell = cell; c->ceq[c->nceq].c = *v; c->nceq++;
}
static void add_lookup(octx_t *c, int64_t cell) {
    if (c->witness_gen_only) return;
    if (c->nlookup == c->caplookup) { c->caplookup = c->caplookup ? c->caplookup * 2 : 1024; c->lookup = (olk_t *)realloc(c->lookup, c->caplookup * sizeof(olk_t)); }
    c->lookup[c->nlookup].cell = cell; c->lookup[c->nlookup].kind = c->semantic ? 1 : 0; c->nlookup++;
}
static void set_selector(octx_t *c, size_t row) {
    if (c->witness_gen_only) return;
    if (row >= c->selcap) { size_t nc = c->selcap ? c->selcap : 1024; while (nc <= row) nc *= 2; c->selector = (uint8_t *)realloc(c->selector, nc); memset(c->selector + c->selcap, 0, nc - c->selcap); c->selcap = nc; }
    c->selector[row] = 1;
}
/* Context::assign_region: push cells, record copy/constant equalities, enable gates */
static void assign_region(octx_t *c, const qc_t *cells, int n, const int *gates, int ng) {
    size_t row0 = c->n;
    for (int i = 0; i < n; i++) {
        adv_push(c, &cells[i].v);
        if (cells[i].kind == QC_EXISTING && cells[i].cell >= 0) add_eq(c, (int64_t)(row0 + i), cells[i].cell, 0);
        else if (cells[i].kind == QC_CONSTANT) add_ceq(c, (int64_t)(row0 + i), &cells[i].v);
    }
    for (int g = 0; g < ng; g++) set_selector(c, row0 + (size_t)gates[g]);
}
static inline oav_t ctx_get(const octx_t *c, int64_t off) { /* Context::get: negative = from end */
    oav_t a; a.cell = off < 0 ? (int64_t)c->n + off : off;
    if (c->streaming) { if ((size_t)a.cell >= c->n || c->n - (size_t)a.cell > ORC_RING) { fprintf(stderr, "oracle: streaming context read back beyond its ring\n"); abort(); } a.v = c->advice[(size_t)a.cell & (ORC_RING - 1)]; return a; }
    a.v = c->advice[a.cell]; return a;
}
static inline oav_t ctx_last(const octx_t *c) { return ctx_get(c, -1); }

int orc_mock_prover(const octx_t *c, uint64_t *gates, uint64_t *equalities, uint64_t *lookups, uint64_t *semantic_failed) {
    uint64_t ng = 0, ne = 0, nl = 0, sf = 0; int bad = 0;
    if (c->witness_gen_only) return -1;
    for (size_t i = 0; i < c->n && i < c->selcap; i++) if (c->selector[i]) {
        ng++;
        if (i + 3 >= c->n) { bad++; continue; }
        ofr_t bc = fr_mul(&c->advice[i + 1], &c->advice[i + 2]);
        ofr_t s = fr_add(&c->advice[i], &bc);
        if (!fr_eq(&s, &c->advice[i + 3])) bad++;
    }
    for (size_t i = 0; i < c->neq; i++) {
        int ok = fr_eq(&c->advice[c->eq[i].a], &c->advice[c->eq[i].b]);
        if (c->eq[i].kind == 1) { if (!ok) sf++; } else { ne++; if (!ok) bad++; }
    }
    for (size_t i = 0; i < c->nceq; i++) { ne++; if (!fr_eq(&c->advice[c->ceq[i].cell], &c->ceq[i].c)) bad++; }
    for (size_t i = 0; i < c->nlookup; i++) {
        nl++; const ofr_t *v = &c->advice[c->lookup[i].cell];
        if (v->l[1] | v->l[2] | v->l[3] || (v->l[0] >> c->lookup_bits)) { if (c->lookup[i].kind == 1) sf++; else bad++; }
    }
    if (gates) *gates = ng; if (equalities) *equalities = ne; if (lookups) *lookups = nl; if (semantic_failed) *semantic_failed = sf;
    return bad;
}

/* ===================================================================== halo2-base Context / GateChip / RangeChip */
static oav_t h2_load_witness(octx_t *c, const ofr_t *v) { qc_t q = Q_W(*v); assign_region(c, &q, 1, NULL, 0); return ctx_last(c); }
static oav_t h2_load_constant(octx_t *c, const ofr_t *v) { qc_t q = Q_C(*v); assign_region(c, &q, 1, NULL, 0); return ctx_last(c); }
static oav_t h2_load_zero(octx_t *c) { /* cached after first use [C] */
    if (c->zero_cell >= 0) { oav_t a; a.cell = c->zero_cell; a.v = fr_from_u64(0); return a; }      /* (the cached cell holds 0: no read-back, the streaming context may have dropped it) */
    ofr_t z = fr_from_u64(0); oav_t a = h2_load_constant(c, &z); c->zero_cell = a.cell; return a;
}
static void h2_constrain_equal(octx_t *c, oav_t a, oav_t b, int kind) { add_eq(c, a.cell, b.cell, kind); }

static oav_t gate_add_q(octx_t *c, qc_t a, qc_t b) { /* [a, b, 1, a+b] g@0 */
    int g = 0; qc_t cells[4] = {a, b, Q_CU(1), Q_W(fr_add(&a.v, &b.v))};
    assign_region(c, cells, 4, &g, 1); return ctx_last(c);
}
static oav_t gate_sub_q(octx_t *c, qc_t a, qc_t b) { /* [a-b, b, 1, a] g@0 -> cell -4 */
    int g = 0; qc_t cells[4] = {Q_W(fr_sub(&a.v, &b.v)), b, Q_CU(1), a};
    assign_region(c, cells, 4, &g, 1); return ctx_get(c, -4);
}
static oav_t gate_mul_q(octx_t *c, qc_t a, qc_t b) { /* [0, a, b, a*b] g@0 */
    int g = 0; qc_t cells[4] = {Q_CU(0), a, b, Q_W(fr_mul(&a.v, &b.v))};
    assign_region(c, cells, 4, &g, 1); return ctx_last(c);
}
static oav_t gate_mul_add_q(octx_t *c, qc_t a, qc_t b, qc_t cc) { /* [c, a, b, a*b+c] g@0 */
    int g = 0; ofr_t ab = fr_mul(&a.v, &b.v); qc_t cells[4] = {cc, a, b, Q_W(fr_add(&ab, &cc.v))};
    assign_region(c, cells, 4, &g, 1); return ctx_last(c);
}
static oav_t gate_select_q(octx_t *c, qc_t a, qc_t b, qc_t sel) {
    /* [a-b, 1, b, a, b, sel, a-b, out] g@0,@4; eq (0,6),(2,4); out = (a-b)*sel + b */
    ofr_t diff = fr_sub(&a.v, &b.v); ofr_t ds = fr_mul(&diff, &sel.v); ofr_t out = fr_add(&ds, &b.v);
    qc_t cells[8] = {Q_W(diff), Q_CU(1), b, a, b, sel, Q_W(diff), Q_W(out)};
    int g[2] = {0, 4}; size_t r0 = c->n;
    assign_region(c, cells, 8, g, 2);
    add_eq(c, (int64_t)r0, (int64_t)r0 + 6, 0); add_eq(c, (int64_t)r0 + 2, (int64_t)r0 + 4, 0);
    return ctx_last(c);
}
/* Rational(1, x) evaluates to x^{-1}; small |x| come from a table (halo2 batch-inverts at the end) */
#define INV_TAB 96
static ofr_t INV_POS[INV_TAB], INV_NEG[INV_TAB]; static int INV_READY = 0;
static ofr_t fr_inv_cached(const ofr_t *x) {
    if (!INV_READY) {
        for (uint64_t k = 1; k < INV_TAB; k++) { ofr_t v = fr_from_u64(k); INV_POS[k] = fr_inv(&v); INV_NEG[k] = fr_neg(&INV_POS[k]); }
        INV_READY = 1;
    }
    if (fr_fits_u128(x) && x->l[1] == 0 && x->l[0] < INV_TAB) return INV_POS[x->l[0]];
    ofr_t nx = fr_neg(x);
    if (fr_fits_u128(&nx) && nx.l[1] == 0 && nx.l[0] < INV_TAB) return INV_NEG[nx.l[0]];
    return fr_inv(x);
}
static oav_t gate_is_zero_q(octx_t *c, qc_t a, int idx_variant) {
    /* [z, a, inv, 1, 0, a, z, 0] g@0,@4; eq (0,6) (+(1,5) in idx_to_indicator's unrolled copy) -> cell -2 */
    int zero = fr_is_zero(&a.v);
    ofr_t z = fr_from_u64(zero ? 1 : 0), inv = zero ? fr_from_u64(1) : fr_inv_cached(&a.v);
    qc_t cells[8] = {Q_W(z), a, Q_W(inv), Q_CU(1), Q_CU(0), a, Q_W(z), Q_CU(0)};
    int g[2] = {0, 4}; size_t r0 = c->n;
    assign_region(c, cells, 8, g, 2);
    add_eq(c, (int64_t)r0, (int64_t)r0 + 6, 0);
    if (idx_variant) add_eq(c, (int64_t)r0 + 1, (int64_t)r0 + 5, 0);
    return ctx_get(c, -2);
}
static void gate_idx_to_indicator(octx_t *c, oav_t idx_in, int len, oav_t *out) {
    qc_t idx = Q_EX(idx_in);
    for (int i = 0; i < len; i++) {
        if (i == 0) { out[0] = gate_is_zero_q(c, idx, 1); idx = Q_EX(ctx_get(c, -3)); }
        else { oav_t d = gate_sub_q(c, idx, Q_CU((uint64_t)i)); out[i] = gate_is_zero_q(c, Q_EX(d), 0); }
    }
}
static oav_t gate_select_by_indicator(octx_t *c, const oav_t *a, int stride, const oav_t *ind, int len) {
    /* [0, a0, ind0, s0, a1, ind1, s1, ...] g@3i */
    qc_t *cells = (qc_t *)malloc((size_t)(1 + 3 * len) * sizeof(qc_t)); int *g = (int *)malloc((size_t)len * sizeof(int));
    ofr_t sum = fr_from_u64(0); cells[0] = Q_CU(0);
    for (int i = 0; i < len; i++) {
        ofr_t t = fr_mul(&a[i * stride].v, &ind[i].v); sum = fr_add(&sum, &t);
        cells[1 + 3 * i] = Q_EX(a[i * stride]); cells[2 + 3 * i] = Q_EX(ind[i]); cells[3 + 3 * i] = Q_W(sum); g[i] = 3 * i;
    }
    assign_region(c, cells, 1 + 3 * len, g, len);
    free(cells); free(g); return ctx_last(c);
}
/* inner_product(a, b) where b are constants; b[0]==1 selects the short form */
static oav_t gate_inner_product_const(octx_t *c, const qc_t *a, const ofr_t *b, int n) {
    qc_t cells_s[1 + 3 * 16]; int g_s[17];
    qc_t *cells = n <= 16 ? cells_s : (qc_t *)malloc((size_t)(1 + 3 * n) * sizeof(qc_t)); int *g = n <= 16 ? g_s : (int *)malloc((size_t)(n + 1) * sizeof(int));
    int k = 0, ng = 0, start = 0; ofr_t sum;
    ofr_t one = fr_from_u64(1);
    if (n > 0 && fr_eq(&b[0], &one)) { cells[k++] = a[0]; sum = a[0].v; start = 1; }
    else { cells[k++] = Q_CU(0); sum = fr_from_u64(0); }
    for (int i = start; i < n; i++) {
        ofr_t t = fr_mul(&a[i].v, &b[i]); sum = fr_add(&sum, &t);
        g[ng++] = k - 1; cells[k++] = a[i]; cells[k++] = Q_C(b[i]); cells[k++] = Q_W(sum);
    }
    assign_region(c, cells, k, g, ng);
    if (n > 16) { free(cells); free(g); }
    return ctx_last(c);
}
static ofr_t fr_pow2(int k) { ofr_t r = {{0, 0, 0, 0}}; r.l[k >> 6] = 1ULL << (k & 63); return r; }
static uint64_t fr_bits(const ofr_t *v, int lo, int width) { /* extract `width` (<=64) bits at bit offset lo */
    if (lo >= 256) return 0;
    int w = lo >> 6, sh = lo & 63;
    uint64_t out = v->l[w] >> sh;
    if (sh && w + 1 < 4) out |= v->l[w + 1] << (64 - sh);
    return width >= 64 ? out : out & ((1ULL << width) - 1);
}
static void gate_assert_bit(octx_t *c, oav_t x) { int g = 0; qc_t cells[4] = {Q_CU(0), Q_EX(x), Q_EX(x), Q_EX(x)}; assign_region(c, cells, 4, &g, 1); }
static oav_t gate_bits_to_num(octx_t *c, const oav_t *bits, int n) {
    qc_t *a = (qc_t *)malloc((size_t)n * sizeof(qc_t)); ofr_t *b = (ofr_t *)malloc((size_t)n * sizeof(ofr_t));
    for (int i = 0; i < n; i++) { a[i] = Q_EX(bits[i]); b[i] = fr_pow2(i); }
    oav_t r = gate_inner_product_const(c, a, b, n); free(a); free(b); return r;
}
static void gate_num_to_bits(octx_t *c, oav_t a, int nbits, oav_t *out) {
    qc_t *q = (qc_t *)malloc((size_t)nbits * sizeof(qc_t)); ofr_t *b = (ofr_t *)malloc((size_t)nbits * sizeof(ofr_t));
    for (int i = 0; i < nbits; i++) { q[i] = Q_W(fr_from_u64(fr_bits(&a.v, i, 1))); b[i] = fr_pow2(i); }
    int64_t row = (int64_t)c->n;
    oav_t acc = gate_inner_product_const(c, q, b, nbits);
    h2_constrain_equal(c, a, acc, 0);
    out[0] = ctx_get(c, row);
    for (int i = 1; i < nbits; i++) out[i] = ctx_get(c, row + 1 + 3 * (i - 1));
    for (int i = 0; i < nbits; i++) gate_assert_bit(c, out[i]);
    free(q); free(b);
}
static void range_check(octx_t *c, oav_t a, int range_bits) {
    int L = c->lookup_bits;
    if (range_bits == 0) { ofr_t z = fr_from_u64(0); add_ceq(c, a.cell, &z); return; }
    int num_limbs = (range_bits + L - 1) / L, rem = range_bits % L;
    oav_t last;
    if (num_limbs == 1) { add_lookup(c, a.cell); last = a; }
    else {
        qc_t q_s[16]; ofr_t b_s[16];
        qc_t *q = num_limbs <= 16 ? q_s : (qc_t *)malloc((size_t)num_limbs * sizeof(qc_t)); ofr_t *b = num_limbs <= 16 ? b_s : (ofr_t *)malloc((size_t)num_limbs * sizeof(ofr_t));
        for (int i = 0; i < num_limbs; i++) { q[i] = Q_W(fr_from_u64(fr_bits(&a.v, i * L, L))); b[i] = fr_pow2(i * L); }
        int64_t row = (int64_t)c->n;
        oav_t acc = gate_inner_product_const(c, q, b, num_limbs);
        h2_constrain_equal(c, a, acc, 0);
        add_lookup(c, row);
        for (int i = 0; i < num_limbs - 1; i++) add_lookup(c, row + 1 + 3 * i);
        last = ctx_get(c, row + 1 + 3 * (num_limbs - 2));
        if (num_limbs > 16) { free(q); free(b); }
    }
    if (rem == 1) gate_assert_bit(c, last);
    else if (rem > 1) { oav_t chk = gate_mul_q(c, Q_EX(last), Q_C(fr_pow2(L - rem))); add_lookup(c, chk.cell); }
}
static void check_less_than(octx_t *c, qc_t a, qc_t b, int num_bits) {
    /* [a+2^n-b, b, 1, a+2^n, -2^n, 1, a] g@0,@3 ; then range_check(cell -7, n) */
    ofr_t p2 = fr_pow2(num_bits); ofr_t sh = fr_add(&p2, &a.v); ofr_t d = fr_sub(&sh, &b.v);
    qc_t cells[7] = {Q_W(d), b, Q_CU(1), Q_W(sh), Q_C(fr_neg(&p2)), Q_CU(1), a};
    int g[2] = {0, 3};
    assign_region(c, cells, 7, g, 2);
    range_check(c, ctx_get(c, -7), num_bits);
}
static int bit_length_u64(uint64_t b) { int n = 0; while (b) { n++; b >>= 1; } return n; }
static void check_less_than_safe(octx_t *c, oav_t a, uint64_t b) {
    int L = c->lookup_bits; int rb = (bit_length_u64(b) + L - 1) / L * L;
    range_check(c, a, rb);
    check_less_than(c, Q_EX(a), Q_CU(b), rb);
}
static void range_decompose_le(octx_t *c, oav_t a, int limb_bits, int n, oav_t *out) {
    qc_t *q = (qc_t *)malloc((size_t)n * sizeof(qc_t)); ofr_t *b = (ofr_t *)malloc((size_t)n * sizeof(ofr_t));
    for (int i = 0; i < n; i++) { q[i] = Q_W(fr_from_u64(fr_bits(&a.v, i * limb_bits, limb_bits))); b[i] = fr_pow2(i * limb_bits); }
    int64_t row = (int64_t)c->n;
    oav_t acc = gate_inner_product_const(c, q, b, n);
    h2_constrain_equal(c, a, acc, 0);
    out[0] = ctx_get(c, row);
    for (int i = 0; i < n - 1; i++) out[i + 1] = ctx_get(c, row + 1 + 3 * i);
    for (int i = 0; i < n; i++) range_check(c, out[i], limb_bits);
    free(q); free(b);
}
static oav_t range_limbs_to_num(octx_t *c, const oav_t *limbs, int n, int limb_bits) {
    qc_t *q = (qc_t *)malloc((size_t)n * sizeof(qc_t)); ofr_t *b = (ofr_t *)malloc((size_t)n * sizeof(ofr_t));
    for (int i = 0; i < n; i++) { q[i] = Q_EX(limbs[i]); b[i] = fr_pow2(i * limb_bits); }
    oav_t r = gate_inner_product_const(c, q, b, n); free(q); free(b); return r;
}

/* ===================================================================== NativeChip (field/native.rs:28-193) */
oav_t orc_load_constant(octx_t *c, const ofr_t *v) { SC("load_constant"); oav_t r = h2_load_constant(c, v); EC(); return r; } /* :28-31 */
oav_t orc_load_zero(octx_t *c) { SC("load_zero"); oav_t r = h2_load_zero(c); EC(); return r; }                              /* :33-36 */
static void nat_load_constants(octx_t *c, const ofr_t *v, int n, oav_t *out) { SC("load_constants"); for (int i = 0; i < n; i++) out[i] = h2_load_constant(c, &v[i]); EC(); } /* :38-41 */
oav_t orc_load_witness(octx_t *c, const ofr_t *v) { SC("load_witness"); oav_t r = h2_load_witness(c, v); EC(); return r; }    /* :43-46 */
oav_t orc_add(octx_t *c, oav_t a, oav_t b) { SC("add"); oav_t r = gate_add_q(c, Q_EX(a), Q_EX(b)); EC(); return r; }          /* :48-57 */
oav_t orc_mul(octx_t *c, oav_t a, oav_t b) { SC("mul"); oav_t r = gate_mul_q(c, Q_EX(a), Q_EX(b)); EC(); return r; }          /* :59-68 */
oav_t orc_mul_add(octx_t *c, oav_t a, oav_t b, oav_t cc) { SC("mul_add"); oav_t r = gate_mul_add_q(c, Q_EX(a), Q_EX(b), Q_EX(cc)); EC(); return r; } /* :70-80 */
oav_t orc_select(octx_t *c, oav_t a, oav_t b, oav_t sel) { SC("select"); oav_t r = gate_select_q(c, Q_EX(a), Q_EX(b), Q_EX(sel)); EC(); return r; } /* :82-92 */
oav_t orc_select_from_idx(octx_t *c, const oav_t *arr, int n, oav_t idx) { /* :95-104 */
    SC("select_from_idx");
    oav_t *ind = (oav_t *)malloc((size_t)n * sizeof(oav_t));
    gate_idx_to_indicator(c, idx, n, ind);
    oav_t r = gate_select_by_indicator(c, arr, 1, ind, n);
    free(ind); EC(); return r;
}
void orc_select_array_by_indicator(octx_t *c, const oav_t *arr2d, int len, int w, const oav_t *ind, oav_t *out) { /* :106-115; arr2d[i*w+j] */
    SC("select_array_by_indicator");
    for (int j = 0; j < w; j++) out[j] = gate_select_by_indicator(c, arr2d + j, w, ind, len);
    EC();
}
void orc_idx_to_indicator(octx_t *c, oav_t idx, int len, oav_t *out) { SC("idx_to_indicator"); gate_idx_to_indicator(c, idx, len, out); EC(); } /* :117-126 */
void orc_num_to_bits(octx_t *c, oav_t a, int bits, oav_t *out) { SC("num_to_bits"); gate_num_to_bits(c, a, bits, out); EC(); }       /* :128-137 */
oav_t orc_bits_to_num(octx_t *c, const oav_t *bits, int n) { SC("bits_to_num"); oav_t r = gate_bits_to_num(c, bits, n); EC(); return r; } /* :139-148 */
void orc_decompose_le(octx_t *c, oav_t a, int limb_bits, int n, oav_t *out) { SC("decompose_le"); range_decompose_le(c, a, limb_bits, n, out); EC(); } /* :150-160 */
oav_t orc_limbs_to_num(octx_t *c, const oav_t *limbs, int n, int limb_bits) { SC("limbs_to_num"); oav_t r = range_limbs_to_num(c, limbs, n, limb_bits); EC(); return r; } /* :162-171 */
void orc_check_less_than_safe(octx_t *c, oav_t a, uint64_t b) { SC("check_less_than_safe"); check_less_than_safe(c, a, b); EC(); } /* :173-177 */
void orc_range_check(octx_t *c, oav_t a, int bits) { SC("range_check"); range_check(c, a, bits); EC(); }                         /* :179-183 */
static void nat_assert_equal(octx_t *c, oav_t a, oav_t b, int kind) { SC("assert_equal"); h2_constrain_equal(c, a, b, kind); EC(); } /* :185-193 */
void orc_constrain_equal(octx_t *c, oav_t a, oav_t b) { nat_assert_equal(c, a, b, 0); }

/* ===================================================================== GoldilocksChip (field/goldilocks/base.rs) */
typedef oav_t glw_t; /* GoldilocksWire(AssignedValue) base.rs:14-36 */
static inline uint64_t glw_value(octx_t *c, glw_t w) { /* :31-35 */
    if (w.v.l[1] | w.v.l[2] | w.v.l[3] || w.v.l[0] >= GL_P) ctx_fail(c, "GoldilocksWire::value: not canonical");
    return w.v.l[0];
}
static void gl_range_check(octx_t *c, glw_t a);
glw_t orc_gl_load_constant(octx_t *c, uint64_t a) { SC("load_constant"); ofr_t v = fr_from_u64(a); glw_t r = orc_load_constant(c, &v); EC(); return r; } /* :61-70 */
static glw_t gl_load_zero(octx_t *c) { SC("load_zero"); glw_t r = orc_gl_load_constant(c, 0); EC(); return r; }       /* :72-75 */
static glw_t gl_load_one(octx_t *c) { SC("load_one"); glw_t r = orc_gl_load_constant(c, 1); EC(); return r; }         /* :77-80 */
static glw_t gl_load_neg_one(octx_t *c) { SC("load_neg_one"); glw_t r = orc_gl_load_constant(c, GL_NEG_ONE); EC(); return r; } /* :82-85 */
static void gl_load_constant_array(octx_t *c, const uint64_t *a, int n, glw_t *out) { SC("load_constant_array"); for (int i = 0; i < n; i++) out[i] = orc_gl_load_constant(c, a[i]); EC(); } /* :87-94 */
glw_t orc_gl_load_witness(octx_t *c, uint64_t a) { /* :107-119 */
    SC("load_witness"); ofr_t v = fr_from_u64(a); glw_t w = orc_load_witness(c, &v); gl_range_check(c, w); EC(); return w;
}
static glw_t gl_select(octx_t *c, glw_t a, glw_t b, oav_t sel) { SC("select"); glw_t r = orc_select(c, a, b, sel); EC(); return r; } /* :138-148 */
static void gl_select_array(octx_t *c, const glw_t *a, const glw_t *b, int n, oav_t sel, glw_t *out) { /* :150-165 */
    SC("select_array"); for (int i = 0; i < n; i++) out[i] = orc_select(c, a[i], b[i], sel); EC();
}
static glw_t gl_select_from_idx(octx_t *c, const glw_t *arr, int n, glw_t idx) { return orc_select_from_idx(c, arr, n, idx); } /* :168-180 (no #[count]) */
static void gl_select_array_from_idx(octx_t *c, const glw_t *arr, int len, int w, glw_t idx, glw_t *out) { /* :182-207 */
    SC("select_array_from_idx");
    oav_t *ind = (oav_t *)malloc((size_t)len * sizeof(oav_t));
    orc_idx_to_indicator(c, idx, len, ind);
    orc_select_array_by_indicator(c, arr, len, w, ind, out);
    free(ind); EC();
}
static void gl_num_to_bits(octx_t *c, glw_t a, int bits, oav_t *out) { SC("num_to_bits"); orc_num_to_bits(c, a, bits, out); EC(); } /* :209-220 */
static glw_t gl_bits_to_num(octx_t *c, const oav_t *bits, int n) { SC("bits_to_num"); glw_t r = orc_bits_to_num(c, bits, n); EC(); return r; } /* :222-232 */
static glw_t gl_add_no_reduce(octx_t *c, glw_t a, glw_t b) { SC("add_no_reduce"); glw_t r = orc_add(c, a, b); EC(); return r; } /* :240-249 */
glw_t orc_gl_add(octx_t *c, glw_t a, glw_t b) { SC("add"); glw_t s = gl_add_no_reduce(c, a, b); glw_t r = orc_gl_reduce(c, s); EC(); return r; } /* :251-260 */
static glw_t gl_sub_no_reduce(octx_t *c, glw_t a, glw_t b) { /* :262-272: a + b*(p-1) */
    SC("sub_no_reduce"); glw_t m1 = gl_load_neg_one(c); glw_t r = orc_mul_add(c, b, m1, a); EC(); return r;
}
glw_t orc_gl_sub(octx_t *c, glw_t a, glw_t b) { SC("sub"); glw_t d = gl_sub_no_reduce(c, a, b); glw_t r = orc_gl_reduce(c, d); EC(); return r; } /* :274-283 */
static glw_t gl_mul_no_reduce(octx_t *c, glw_t a, glw_t b) { SC("mul_no_reduce"); glw_t r = orc_mul(c, a, b); EC(); return r; } /* :285-294 */
glw_t orc_gl_mul(octx_t *c, glw_t a, glw_t b) { SC("mul"); glw_t p = gl_mul_no_reduce(c, a, b); glw_t r = orc_gl_reduce(c, p); EC(); return r; } /* :296-305 */
static glw_t gl_mul_add_no_reduce(octx_t *c, glw_t a, glw_t b, glw_t cc) { SC("mul_add_no_reduce"); glw_t r = orc_mul_add(c, a, b, cc); EC(); return r; } /* :307-317 */
glw_t orc_gl_mul_add(octx_t *c, glw_t a, glw_t b, glw_t cc) { SC("mul_add"); glw_t p = gl_mul_add_no_reduce(c, a, b, cc); glw_t r = orc_gl_reduce(c, p); EC(); return r; } /* :319-329 */
glw_t orc_gl_mul_sub(octx_t *c, glw_t a, glw_t b, glw_t cc) { /* :332-343 */
    SC("mul_sub"); glw_t p = gl_mul_no_reduce(c, a, b); glw_t d = gl_sub_no_reduce(c, p, cc); glw_t r = orc_gl_reduce(c, d); EC(); return r;
}
glw_t orc_gl_reduce(octx_t *c, glw_t a) { /* :346-368 */
    SC("reduce");
    /* 1. hint: quotient = from_noncanonical_biguint(val / ORDER), remainder = from_noncanonical_biguint(val) */
    uint64_t q, r;
    if (fr_fits_u128(&a.v)) {   /* same values as BigUint val / ORDER and val % ORDER, without a 128-bit divider */
        u128 v = fr_lo128(&a.v);
        r = glf_reduce128(v);
        u128 d = v - r; uint64_t dl = (uint64_t)d; uint64_t qlo = dl + (dl << 32);       /* exact division: p^-1 = 1 + 2^32 mod 2^64 */
        uint64_t qhi = ((u128)qlo * GL_P != d) ? 1 : 0;
        q = glf_reduce128(((u128)qhi << 64) | qlo);
    }
    else { /* generic 256-bit long division by p (never reached on the FRI path) */
        u128 rem = 0; uint64_t ql[4] = {0, 0, 0, 0};
        for (int i = 255; i >= 0; i--) {
            rem = (rem << 1) | ((a.v.l[i >> 6] >> (i & 63)) & 1);
            if (rem >= GL_P) { rem -= GL_P; ql[i >> 6] |= 1ULL << (i & 63); }
        }
        u128 qm = 0; for (int i = 3; i >= 0; i--) { qm = ((qm << 64) | ql[i]) % GL_P; }
        q = (uint64_t)qm; r = (uint64_t)rem;
    }
    glw_t qw = orc_gl_load_witness(c, q);
    glw_t rw = orc_gl_load_witness(c, r);
    ofr_t pv = fr_from_u64(GL_P);
    oav_t p = orc_load_constant(c, &pv);
    oav_t rhs = orc_mul_add(c, qw, p, rw);
    nat_assert_equal(c, a, rhs, 0);
    EC(); return rw;
}
static void gl_assert_equal(octx_t *c, glw_t a, glw_t b, int kind) { SC("assert_equal"); nat_assert_equal(c, a, b, kind); EC(); } /* :456-465 */
glw_t orc_gl_div(octx_t *c, glw_t a, glw_t b) { /* :371-393 */
    SC("div");
    uint64_t bv = glw_value(c, b), av = glw_value(c, a);
    if (bv == 0) { ctx_fail(c, "GoldilocksChip::div: division by zero (base.rs:379)"); bv = 1; }
    glw_t res = orc_gl_load_witness(c, glf_mul(av, glf_inv(bv)));
    glw_t prod = orc_gl_mul(c, b, res);
    gl_assert_equal(c, a, prod, 0);
    EC(); return res;
}
glw_t orc_gl_inv(octx_t *c, glw_t a) { SC("inv"); glw_t one = gl_load_one(c); glw_t r = orc_gl_div(c, one, a); EC(); return r; } /* :395-399 */
static glw_t gl_square(octx_t *c, glw_t a) { SC("square"); glw_t r = orc_gl_mul(c, a, a); EC(); return r; } /* :401-404 */
glw_t orc_gl_exp_from_bits_const_base(octx_t *c, uint64_t base, const oav_t *bits, int n) { /* :407-430 */
    SC("exp_from_bits_const_base");
    glw_t product = gl_load_one(c);
    for (int i = 0; i < n; i++) {
        uint64_t pw = 1ULL << i;
        glw_t bpm1 = orc_gl_load_constant(c, glf_sub(glf_exp(base, pw), 1));
        glw_t a = orc_gl_mul(c, bpm1, product);
        product = orc_gl_mul_add(c, a, bits[i], product);
    }
    EC(); return product;
}
glw_t orc_gl_exp_power_of_2(octx_t *c, glw_t base, int power_log) { /* :433-445 */
    SC("exp_power_of_2"); glw_t p = base; for (int i = 0; i < power_log; i++) p = gl_square(c, p); EC(); return p;
}
static void gl_range_check(octx_t *c, glw_t a) { SC("range_check"); orc_check_less_than_safe(c, a, GL_P); EC(); } /* :447-454 */

/* ===================================================================== GoldilocksQuadExtChip (field/goldilocks/extension.rs) */
typedef struct { glw_t e[2]; } exw_t;
static gle_t exw_value(octx_t *c, exw_t a) { gle_t r; r.c[0] = glw_value(c, a.e[0]); r.c[1] = glw_value(c, a.e[1]); return r; } /* :19-25 */
static exw_t ex_load_constant(octx_t *c, gle_t a) { SC("load_constant"); exw_t r; r.e[0] = orc_gl_load_constant(c, a.c[0]); r.e[1] = orc_gl_load_constant(c, a.c[1]); EC(); return r; } /* :52-64 */
static exw_t ex_load_zero(octx_t *c) { SC("load_zero"); gle_t z = {{0, 0}}; exw_t r = ex_load_constant(c, z); EC(); return r; } /* :66-69 */
static exw_t ex_load_one(octx_t *c) { SC("load_one"); gle_t o = {{1, 0}}; exw_t r = ex_load_constant(c, o); EC(); return r; }  /* :71-74 */
static exw_t ex_load_witness(octx_t *c, gle_t a) { SC("load_witness"); exw_t r; r.e[0] = orc_gl_load_witness(c, a.c[0]); r.e[1] = orc_gl_load_witness(c, a.c[1]); EC(); return r; } /* :85-97 */
static exw_t ex_select_from_idx(octx_t *c, const exw_t *arr, int n, glw_t idx) { /* :99-118 */
    SC("select_from_idx");
    glw_t *a0 = (glw_t *)malloc((size_t)n * sizeof(glw_t)), *a1 = (glw_t *)malloc((size_t)n * sizeof(glw_t));
    for (int i = 0; i < n; i++) { a0[i] = arr[i].e[0]; a1[i] = arr[i].e[1]; }
    exw_t r; r.e[0] = gl_select_from_idx(c, a0, n, idx); r.e[1] = gl_select_from_idx(c, a1, n, idx);
    free(a0); free(a1); EC(); return r;
}
static exw_t ex_load_base(octx_t *c, glw_t a) { SC("load_base"); exw_t r; r.e[1] = gl_load_zero(c); r.e[0] = a; EC(); return r; } /* :120-128 */
static exw_t ex_reduce(octx_t *c, exw_t a) { SC("reduce"); exw_t r; r.e[0] = orc_gl_reduce(c, a.e[0]); r.e[1] = orc_gl_reduce(c, a.e[1]); EC(); return r; } /* :368-380 */
static exw_t ex_add_no_reduce(octx_t *c, exw_t a, exw_t b) { SC("add_no_reduce"); exw_t r; r.e[0] = gl_add_no_reduce(c, a.e[0], b.e[0]); r.e[1] = gl_add_no_reduce(c, a.e[1], b.e[1]); EC(); return r; } /* :130-144 */
static exw_t ex_add(octx_t *c, exw_t a, exw_t b) { SC("add"); exw_t s = ex_add_no_reduce(c, a, b); exw_t r = ex_reduce(c, s); EC(); return r; } /* :146-155 */
static exw_t ex_sub_no_reduce(octx_t *c, exw_t a, exw_t b) { SC("sub_no_reduce"); exw_t r; r.e[0] = gl_sub_no_reduce(c, a.e[0], b.e[0]); r.e[1] = gl_sub_no_reduce(c, a.e[1], b.e[1]); EC(); return r; } /* :157-171 */
static exw_t ex_sub(octx_t *c, exw_t a, exw_t b) { SC("sub"); exw_t d = ex_sub_no_reduce(c, a, b); exw_t r = ex_reduce(c, d); EC(); return r; } /* :173-182 */
static exw_t ex_mul(octx_t *c, exw_t a, exw_t b) { /* :211-234 */
    SC("mul");
    glw_t w = orc_gl_load_constant(c, 7);
    glw_t a0b0 = orc_gl_mul(c, a.e[0], b.e[0]);
    glw_t a1b1 = orc_gl_mul(c, a.e[1], b.e[1]);
    glw_t wa1b1 = orc_gl_mul(c, w, a1b1);
    exw_t r; r.e[0] = orc_gl_add(c, a0b0, wa1b1);
    glw_t a0b1 = orc_gl_mul(c, a.e[0], b.e[1]);
    glw_t a1b0 = orc_gl_mul(c, a.e[1], b.e[0]);
    r.e[1] = orc_gl_add(c, a0b1, a1b0);
    EC(); return r;
}
static exw_t ex_square(octx_t *c, exw_t a) { /* :248-268 */
    SC("square");
    glw_t w = orc_gl_load_constant(c, 7);
    glw_t a0a0 = gl_square(c, a.e[0]);
    glw_t a1a1 = gl_square(c, a.e[1]);
    glw_t wa1a1 = orc_gl_mul(c, w, a1a1);
    exw_t r; r.e[0] = orc_gl_add(c, a0a0, wa1a1);
    glw_t a0a1 = orc_gl_mul(c, a.e[0], a.e[1]);
    r.e[1] = orc_gl_add(c, a0a1, a0a1);
    EC(); return r;
}
static exw_t ex_mul_add(octx_t *c, exw_t a, exw_t b, exw_t cc) { SC("mul_add"); exw_t ab = ex_mul(c, a, b); exw_t r = ex_add(c, ab, cc); EC(); return r; } /* :284-294 */
static void ex_assert_equal(octx_t *c, exw_t a, exw_t b, int kind) { SC("assert_equal"); gl_assert_equal(c, a.e[0], b.e[0], kind); gl_assert_equal(c, a.e[1], b.e[1], kind); EC(); } /* :447-459 */
static exw_t ex_inv(octx_t *c, exw_t a) { /* :320-340 */
    SC("inv");
    gle_t av = exw_value(c, a);
    if (av.c[0] == 0 && av.c[1] == 0) { ctx_fail(c, "GoldilocksQuadExtChip::inv: zero"); av.c[0] = 1; }
    exw_t inv = ex_load_witness(c, gle_inv(av));
    exw_t prod = ex_mul(c, a, inv);
    exw_t one = ex_load_one(c);
    ex_assert_equal(c, prod, one, 0);
    EC(); return inv;
}
static exw_t ex_div(octx_t *c, exw_t a, exw_t b) { SC("div"); exw_t bi = ex_inv(c, b); exw_t r = ex_mul(c, a, bi); EC(); return r; } /* :237-246 */
static exw_t ex_scalar_mul(octx_t *c, exw_t a, glw_t b) { SC("scalar_mul"); exw_t r; r.e[0] = orc_gl_mul(c, a.e[0], b); r.e[1] = orc_gl_mul(c, a.e[1], b); EC(); return r; } /* :342-353 */
static exw_t ex_scalar_div(octx_t *c, exw_t a, glw_t b) { SC("scalar_div"); exw_t r; r.e[0] = orc_gl_div(c, a.e[0], b); r.e[1] = orc_gl_div(c, a.e[1], b); EC(); return r; } /* :355-366 */
static int bits_u64(uint64_t n) { return bit_length_u64(n); } /* plonky2::util::bits_u64 */
static exw_t ex_exp_u64(octx_t *c, exw_t base, uint64_t e) { /* :382-407 */
    SC("exp_u64");
    exw_t r;
    if (e == 0) { r = ex_load_one(c); EC(); return r; }
    if (e == 1) { EC(); return base; }
    if (e == 2) { r = ex_mul(c, base, base); EC(); return r; }
    exw_t cur = base, prod = ex_load_one(c);
    for (int j = 0; j < bits_u64(e); j++) {
        if (j != 0) cur = ex_square(c, cur);
        if ((e >> j) & 1) prod = ex_mul(c, prod, cur);
    }
    EC(); return prod;
}
static exw_t ex_reduce_with_powers(octx_t *c, const exw_t *terms, int n, exw_t scalar) { /* :424-437 */
    SC("reduce_with_powers");
    exw_t sum = ex_load_zero(c);
    for (int i = n - 1; i >= 0; i--) { sum = ex_mul(c, sum, scalar); sum = ex_add(c, sum, terms[i]); }
    EC(); return sum;
}
void orc_ext_mul(octx_t *c, const oav_t a[2], const oav_t b[2], oav_t out[2]) { exw_t x = {{a[0], a[1]}}, y = {{b[0], b[1]}}; exw_t r = ex_mul(c, x, y); out[0] = r.e[0]; out[1] = r.e[1]; }
void orc_ext_inv(octx_t *c, const oav_t a[2], oav_t out[2]) { exw_t x = {{a[0], a[1]}}; exw_t r = ex_inv(c, x); out[0] = r.e[0]; out[1] = r.e[1]; }
void orc_ext_div(octx_t *c, const oav_t a[2], const oav_t b[2], oav_t out[2]) { exw_t x = {{a[0], a[1]}}, y = {{b[0], b[1]}}; exw_t r = ex_div(c, x, y); out[0] = r.e[0]; out[1] = r.e[1]; }

/* ===================================================================== Goldilocks Poseidon (hash/poseidon/permutation.rs) */
#define SPONGE_WIDTH 12
#define SPONGE_RATE 8
#define HALF_N_FULL_ROUNDS 4
#define N_PARTIAL_ROUNDS 22
static glw_t pg_mds_row_shf(octx_t *c, const oconsts_t *k, int r, const glw_t *v) { /* :43-71 */
    SC("mds_row_shf");
    glw_t res = orc_gl_load_constant(c, 0);
    for (int i = 0; i < SPONGE_WIDTH; i++) { glw_t cc = orc_gl_load_constant(c, k->mds_circ[i]); res = orc_gl_mul_add(c, cc, v[(i + r) % SPONGE_WIDTH], res); }
    { glw_t cc = orc_gl_load_constant(c, k->mds_diag[r]); res = orc_gl_mul_add(c, cc, v[r], res); }
    EC(); return res;
}
static void pg_mds_layer(octx_t *c, const oconsts_t *k, glw_t *st) { /* :73-87 */
    SC("mds_layer");
    uint64_t z[SPONGE_WIDTH] = {0}; glw_t res[SPONGE_WIDTH];
    gl_load_constant_array(c, z, SPONGE_WIDTH, res);
    for (int r = 0; r < SPONGE_WIDTH; r++) res[r] = pg_mds_row_shf(c, k, r, st);
    memcpy(st, res, sizeof(res)); EC();
}
static void pg_partial_first_constant_layer(octx_t *c, const oconsts_t *k, glw_t *st) { /* :89-106 */
    SC("partial_first_constant_layer");
    for (int i = 0; i < SPONGE_WIDTH; i++) { glw_t cc = orc_gl_load_constant(c, k->fast_partial_first_round_constant[i]); st[i] = orc_gl_add(c, st[i], cc); }
    EC();
}
static void pg_mds_partial_layer_init(octx_t *c, const oconsts_t *k, glw_t *st) { /* :108-132 */
    SC("mds_partial_layer_init");
    uint64_t z[SPONGE_WIDTH] = {0}; glw_t res[SPONGE_WIDTH];
    gl_load_constant_array(c, z, SPONGE_WIDTH, res);
    res[0] = st[0];
    for (int r = 1; r < SPONGE_WIDTH; r++) for (int cc = 1; cc < SPONGE_WIDTH; cc++) {
        glw_t t = orc_gl_load_constant(c, k->fast_partial_round_initial_matrix[r - 1][cc - 1]);
        res[cc] = orc_gl_mul_add(c, t, st[r], res[cc]);
    }
    memcpy(st, res, sizeof(res)); EC();
}
static void pg_mds_partial_layer_fast(octx_t *c, const oconsts_t *k, glw_t *st, int r) { /* :134-173 */
    SC("mds_partial_layer_fast");
    glw_t s0 = st[0];
    glw_t m00 = orc_gl_load_constant(c, k->mds_circ[0] + k->mds_diag[0]);
    glw_t d = orc_gl_mul(c, m00, s0);
    for (int i = 1; i < SPONGE_WIDTH; i++) { glw_t t = orc_gl_load_constant(c, k->fast_partial_round_w_hats[r][i - 1]); d = orc_gl_mul_add(c, t, st[i], d); }
    uint64_t z[SPONGE_WIDTH] = {0}; glw_t res[SPONGE_WIDTH];
    gl_load_constant_array(c, z, SPONGE_WIDTH, res);
    res[0] = d;
    for (int i = 1; i < SPONGE_WIDTH; i++) { glw_t t = orc_gl_load_constant(c, k->fast_partial_round_vs[r][i - 1]); res[i] = orc_gl_mul_add(c, t, st[0], st[i]); }
    memcpy(st, res, sizeof(res)); EC();
}
static void pg_constant_layer(octx_t *c, const oconsts_t *k, glw_t *st, int round_ctr) { /* :175-193 */
    SC("constant_layer");
    for (int i = 0; i < SPONGE_WIDTH; i++) { glw_t rc = orc_gl_load_constant(c, k->all_round_constants[i + SPONGE_WIDTH * round_ctr]); st[i] = orc_gl_add(c, st[i], rc); }
    EC();
}
static glw_t pg_sbox_monomial(octx_t *c, glw_t x) { /* :195-207 */
    SC("sbox_monomial");
    glw_t x2 = orc_gl_mul(c, x, x), x4 = orc_gl_mul(c, x2, x2), x6 = orc_gl_mul(c, x4, x2), x7 = orc_gl_mul(c, x6, x);
    EC(); return x7;
}
static void pg_sbox_layer(octx_t *c, glw_t *st) { SC("sbox_layer"); for (int i = 0; i < SPONGE_WIDTH; i++) st[i] = pg_sbox_monomial(c, st[i]); EC(); } /* :209-214 */
static void pg_partial_rounds(octx_t *c, const oconsts_t *k, glw_t *st, int *round_ctr) { /* :216-239 */
    SC("partial_rounds");
    pg_partial_first_constant_layer(c, k, st);
    pg_mds_partial_layer_init(c, k, st);
    for (int r = 0; r < N_PARTIAL_ROUNDS; r++) {
        st[0] = pg_sbox_monomial(c, st[0]);
        glw_t cc = orc_gl_load_constant(c, k->fast_partial_round_constants[r]);
        st[0] = orc_gl_add(c, st[0], cc);
        pg_mds_partial_layer_fast(c, k, st, r);
    }
    *round_ctr += N_PARTIAL_ROUNDS; EC();
}
static void pg_full_rounds(octx_t *c, const oconsts_t *k, glw_t *st, int *round_ctr) { /* :241-254 */
    SC("full_rounds");
    for (int i = 0; i < HALF_N_FULL_ROUNDS; i++) { pg_constant_layer(c, k, st, *round_ctr); pg_sbox_layer(c, st); pg_mds_layer(c, k, st); *round_ctr += 1; }
    EC();
}
static void pg_permute(octx_t *c, const oconsts_t *k, glw_t *st) { /* :270-284 */
    int round_ctr = 0;
    pg_full_rounds(c, k, st, &round_ctr); pg_partial_rounds(c, k, st, &round_ctr); pg_full_rounds(c, k, st, &round_ctr);
}
void orc_gl_poseidon_permute(octx_t *c, const oconsts_t *k, const oav_t in[12], oav_t out[12]) { memcpy(out, in, 12 * sizeof(oav_t)); pg_permute(c, k, out); }
static void pg_absorb_goldilocks(octx_t *c, const oconsts_t *k, glw_t *st, const glw_t *in, int n) { /* :286-301 */
    for (int off = 0; off < n; off += SPONGE_RATE) {
        int len = n - off < SPONGE_RATE ? n - off : SPONGE_RATE;
        memcpy(st, in + off, (size_t)len * sizeof(glw_t));
        pg_permute(c, k, st);
    }
}

/* ===================================================================== BN254 Poseidon (hash/poseidon_bn254/permutation.rs) */
#define BN_WIDTH 4
#define BN_RATE 3
#define BN_FULL_ROUNDS 8
#define BN_PARTIAL_ROUNDS 56
static oav_t pb_exp5(octx_t *c, oav_t x) { SC("exp5"); oav_t x2 = orc_mul(c, x, x), x4 = orc_mul(c, x2, x2), r = orc_mul(c, x4, x); EC(); return r; } /* :48-55 */
static void pb_exp5_state(octx_t *c, oav_t *st) { SC("exp5_state"); for (int i = 0; i < BN_WIDTH; i++) st[i] = pb_exp5(c, st[i]); EC(); } /* :57-62 */
static void pb_mix(octx_t *c, oav_t *st, oav_t m[4][4]) { /* :64-81 */
    SC("mix");
    oav_t z = orc_load_zero(c); oav_t ns[BN_WIDTH] = {z, z, z, z};
    for (int i = 0; i < BN_WIDTH; i++) for (int j = 0; j < BN_WIDTH; j++) ns[i] = orc_mul_add(c, m[j][i], st[j], ns[i]);
    memcpy(st, ns, sizeof(ns)); EC();
}
static void pb_ark(octx_t *c, const oconsts_t *k, oav_t *st, int it) { /* :162-170 */
    SC("ark");
    for (int i = 0; i < BN_WIDTH; i++) { oav_t cc = orc_load_constant(c, &k->bn_c[it + i]); st[i] = orc_add(c, st[i], cc); }
    EC();
}
static void pb_partial_rounds(octx_t *c, const oconsts_t *k, oav_t *st) { /* :83-110 */
    SC("partial_rounds");
    for (int i = 0; i < BN_PARTIAL_ROUNDS; i++) {
        st[0] = pb_exp5(c, st[0]);
        oav_t cc = h2_load_constant(c, &k->bn_c[(BN_FULL_ROUNDS / 2 + 1) * BN_WIDTH + i]); /* ctx.ctx.load_constant: no #[count] frame */
        st[0] = orc_add(c, st[0], cc);
        oav_t ns0 = orc_load_zero(c);
        for (int j = 0; j < BN_WIDTH; j++) { oav_t s = h2_load_constant(c, &k->bn_s[(BN_WIDTH * 2 - 1) * i + j]); ns0 = orc_mul_add(c, s, st[j], ns0); }
        for (int kk = 1; kk < BN_WIDTH; kk++) { oav_t s = h2_load_constant(c, &k->bn_s[(BN_WIDTH * 2 - 1) * i + BN_WIDTH + kk - 1]); st[kk] = orc_mul_add(c, s, st[0], st[kk]); }
        st[0] = ns0;
    }
    EC();
}
static void pb_full_rounds(octx_t *c, const oconsts_t *k, oav_t *st, int is_first) { /* :112-160 */
    SC("full_rounds");
    oav_t m[4][4], p[4][4];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) m[i][j] = orc_load_constant(c, &k->bn_m[i][j]);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) p[i][j] = orc_load_constant(c, &k->bn_p[i][j]);
    for (int i = 0; i < BN_FULL_ROUNDS / 2 - 1; i++) {
        pb_exp5_state(c, st);
        if (is_first) pb_ark(c, k, st, (i + 1) * BN_WIDTH);
        else pb_ark(c, k, st, (BN_FULL_ROUNDS / 2 + 1) * BN_WIDTH + BN_PARTIAL_ROUNDS + i * BN_WIDTH);
        pb_mix(c, st, m);
    }
    pb_exp5_state(c, st);
    if (is_first) { pb_ark(c, k, st, (BN_FULL_ROUNDS / 2) * BN_WIDTH); pb_mix(c, st, p); }
    else pb_mix(c, st, m);
    EC();
}
static void pb_permute(octx_t *c, const oconsts_t *k, oav_t *st) { /* :190-203 */
    pb_ark(c, k, st, 0); pb_full_rounds(c, k, st, 1); pb_partial_rounds(c, k, st); pb_full_rounds(c, k, st, 0);
}
void orc_bn_poseidon_permute(octx_t *c, const oconsts_t *k, const oav_t in[4], oav_t out[4]) { memcpy(out, in, 4 * sizeof(oav_t)); pb_permute(c, k, out); }
static void pb_absorb_goldilocks(octx_t *c, const oconsts_t *k, oav_t *st, const glw_t *in, int n) { /* :205-228 */
    for (int off = 0; off < n; off += BN_RATE * 3) {
        int len = n - off < BN_RATE * 3 ? n - off : BN_RATE * 3;
        for (int j = 0, o = 0; o < len; j++, o += 3) {
            int l3 = len - o < 3 ? len - o : 3;
            st[j + 1] = orc_limbs_to_num(c, in + off + o, l3, 64);
        }
        pb_permute(c, k, st);
    }
}

/* ===================================================================== HasherChip (hash/mod.rs:52-127, hash/poseidon/hash.rs, hash/poseidon_bn254/hash.rs) */
typedef struct { oav_t e[4]; } hw_t; /* PoseidonHashWire: 4 GL wires; PoseidonBN254HashWire: e[0] = value */
static hw_t hs_load_goldilocks_slice(octx_t *c, int mode, const glw_t *in, int n) {
    SC("load_goldilocks_slice"); hw_t h;
    if (mode == 0) { uint64_t z[4] = {0, 0, 0, 0}; gl_load_constant_array(c, z, 4, h.e); for (int i = 0; i < n; i++) h.e[i] = in[i]; } /* poseidon/hash.rs:98-112 */
    else { h.e[0] = orc_limbs_to_num(c, in, n, 64); h.e[1] = h.e[2] = h.e[3] = h.e[0]; }                                          /* poseidon_bn254/hash.rs:100-114 */
    EC(); return h;
}
static hw_t hs_hash_no_pad(octx_t *c, const oconsts_t *k, int mode, const glw_t *in, int n) {
    SC("hash_no_pad"); hw_t h;
    if (mode == 0) { /* poseidon/hash.rs:161-184 */
        uint64_t z[SPONGE_WIDTH] = {0}; glw_t st[SPONGE_WIDTH];
        gl_load_constant_array(c, z, SPONGE_WIDTH, st);
        pg_absorb_goldilocks(c, k, st, in, n);
        for (int i = 0; i < 4; i++) h.e[i] = st[i];
    } else { /* poseidon_bn254/hash.rs:156-179 */
        ofr_t z[4]; memset(z, 0, sizeof(z)); oav_t st[4];
        nat_load_constants(c, z, 4, st);
        pb_absorb_goldilocks(c, k, st, in, n);
        h.e[0] = st[0]; h.e[1] = h.e[2] = h.e[3] = st[0];
    }
    EC(); return h;
}
static hw_t hs_hash_or_noop(octx_t *c, const oconsts_t *k, int mode, const glw_t *in, int n) { /* hash/mod.rs:109-119 */
    int maxg = mode == 0 ? 4 : 3;
    return n <= maxg ? hs_load_goldilocks_slice(c, mode, in, n) : hs_hash_no_pad(c, k, mode, in, n);
}
static hw_t hs_two_to_one(octx_t *c, const oconsts_t *k, int mode, hw_t l, hw_t r) {
    SC("two_to_one"); hw_t h;
    if (mode == 0) { /* poseidon/hash.rs:187-214 */
        uint64_t z[SPONGE_WIDTH] = {0}; glw_t st[SPONGE_WIDTH];
        gl_load_constant_array(c, z, SPONGE_WIDTH, st);
        for (int i = 0; i < 4; i++) { st[i] = l.e[i]; st[4 + i] = r.e[i]; }
        pg_permute(c, k, st);
        for (int i = 0; i < 4; i++) h.e[i] = st[i];
    } else { /* poseidon_bn254/hash.rs:182-209 */
        ofr_t z[4]; memset(z, 0, sizeof(z)); oav_t st[4];
        nat_load_constants(c, z, 4, st);
        st[2] = l.e[0]; st[3] = r.e[0];
        pb_permute(c, k, st);
        h.e[0] = st[0]; h.e[1] = h.e[2] = h.e[3] = st[0];
    }
    EC(); return h;
}
static hw_t hs_select(octx_t *c, int mode, hw_t a, hw_t b, oav_t sel) {
    SC("select"); hw_t h;
    if (mode == 0) gl_select_array(c, a.e, b.e, 4, sel, h.e);                         /* poseidon/hash.rs:114-126 */
    else { h.e[0] = orc_select(c, a.e[0], b.e[0], sel); h.e[1] = h.e[2] = h.e[3] = h.e[0]; } /* poseidon_bn254/hash.rs:116-127 */
    EC(); return h;
}
static hw_t hs_select_from_idx(octx_t *c, int mode, const hw_t *arr, int n, glw_t idx) {
    SC("select_from_idx"); hw_t h;
    if (mode == 0) { /* poseidon/hash.rs:128-146 */
        glw_t *flat = (glw_t *)malloc((size_t)n * 4 * sizeof(glw_t));
        for (int i = 0; i < n; i++) for (int j = 0; j < 4; j++) flat[i * 4 + j] = arr[i].e[j];
        gl_select_array_from_idx(c, flat, n, 4, idx, h.e); free(flat);
    } else { /* poseidon_bn254/hash.rs:129-143 */
        oav_t *flat = (oav_t *)malloc((size_t)n * sizeof(oav_t));
        for (int i = 0; i < n; i++) flat[i] = arr[i].e[0];
        h.e[0] = orc_select_from_idx(c, flat, n, idx); h.e[1] = h.e[2] = h.e[3] = h.e[0]; free(flat);
    }
    EC(); return h;
}
static void hs_assert_equal(octx_t *c, int mode, hw_t a, hw_t b, int kind) {
    SC("assert_equal");
    if (mode == 0) for (int i = 0; i < 4; i++) gl_assert_equal(c, a.e[i], b.e[i], kind); /* poseidon/hash.rs:148-159 */
    else nat_assert_equal(c, a.e[0], b.e[0], kind);                                      /* poseidon_bn254/hash.rs:145-154 */
    EC();
}
/* HashWire::to_goldilocks_vec */
static int hw_to_goldilocks_vec(octx_t *c, int mode, hw_t h, glw_t *out) {
    if (mode == 0) { for (int i = 0; i < 4; i++) out[i] = h.e[i]; return 4; } /* poseidon/hash.rs:22-30 */
    SC("to_goldilocks_vec"); orc_decompose_le(c, h.e[0], 56, 5, out); EC(); return 5; /* poseidon_bn254/hash.rs:29-44 */
}
void orc_hash_no_pad(octx_t *c, const oconsts_t *k, int mode, const oav_t *in, int n, oav_t out[4]) { hw_t h = hs_hash_no_pad(c, k, mode, in, n); memcpy(out, h.e, sizeof(h.e)); }
void orc_two_to_one(octx_t *c, const oconsts_t *k, int mode, const oav_t l[4], const oav_t r[4], oav_t out[4]) {
    hw_t a, b; memcpy(a.e, l, sizeof(a.e)); memcpy(b.e, r, sizeof(b.e)); hw_t h = hs_two_to_one(c, k, mode, a, b); memcpy(out, h.e, sizeof(h.e));
}

/* ===================================================================== MerkleTreeChip (merkle/mod.rs:57-78) */
static void mk_verify_proof_to_cap_with_cap_index(octx_t *c, const oconsts_t *k, int mode, const glw_t *leaf, int n_leaf,
                                                  const oav_t *bits, int n_bits, glw_t cap_index, const hw_t *cap, int n_cap,
                                                  const hw_t *sib, int n_sib) {
    SC("verify_proof_to_cap_with_cap_index");
    hw_t node = hs_hash_or_noop(c, k, mode, leaf, n_leaf);
    int n = n_sib < n_bits ? n_sib : n_bits; /* zip */
    for (int i = 0; i < n; i++) {
        hw_t left = hs_select(c, mode, sib[i], node, bits[i]);
        hw_t right = hs_select(c, mode, node, sib[i], bits[i]);
        node = hs_two_to_one(c, k, mode, left, right);
    }
    hw_t root = hs_select_from_idx(c, mode, cap, n_cap, cap_index);
    hs_assert_equal(c, mode, root, node, 1);
    EC();
}
void orc_merkle_verify(octx_t *c, const oconsts_t *k, int mode, const oav_t *leaf, int n_leaf, const oav_t *bits, int n_bits, oav_t cap_index,
                       const oav_t *cap, int n_cap, const oav_t *siblings, int n_sib) {
    int hw = mode == 0 ? 4 : 1;
    hw_t *capw = (hw_t *)malloc((size_t)n_cap * sizeof(hw_t)), *sibw = (hw_t *)malloc((size_t)(n_sib + 1) * sizeof(hw_t));
    for (int i = 0; i < n_cap; i++) for (int j = 0; j < 4; j++) capw[i].e[j] = cap[i * hw + (j < hw ? j : 0)];
    for (int i = 0; i < n_sib; i++) for (int j = 0; j < 4; j++) sibw[i].e[j] = siblings[i * hw + (j < hw ? j : 0)];
    mk_verify_proof_to_cap_with_cap_index(c, k, mode, leaf, n_leaf, bits, n_bits, cap_index, capw, n_cap, sibw, n_sib);
    free(capw); free(sibw);
}

/* ===================================================================== shape helpers */
#define MAX_STEPS 16
typedef struct {
    int lde_bits, n_steps, arity[MAX_STEPS], final_poly_len, cap_size;
    int n_oracles, oracle_polys[3];
} oderived_t;
static void derive(const oshape_t *s, oderived_t *d) {
    d->lde_bits = s->degree_bits + s->rate_bits;
    int db = s->degree_bits; d->n_steps = 0;
    /* plonky2 FriReductionStrategy::ConstantArityBits */
    while (db > s->final_poly_bits && db + s->rate_bits - s->arity_bits >= s->cap_height) { d->arity[d->n_steps++] = s->arity_bits; db -= s->arity_bits; }
    d->final_poly_len = 1 << db; d->cap_size = 1 << s->cap_height;
    d->n_oracles = 0; d->oracle_polys[d->n_oracles++] = s->n_cols;
    if (s->n_perm_z > 0) d->oracle_polys[d->n_oracles++] = s->n_perm_z;
    d->oracle_polys[d->n_oracles++] = s->n_quotient;
}
/* flat proof layout = WitnessChip load order (witness/mod.rs:236-294). hashes are 4 words in both modes. */
size_t orc_proof_words(const oshape_t *s) {
    oderived_t d; derive(s, &d); size_t w = 0;
    w += (size_t)d.cap_size * 4 * 2;                               /* trace_cap, quotient_polys_cap */
    w += 2 * (size_t)(2 * s->n_cols + 2 * s->n_perm_z + s->n_quotient); /* openings (ext) */
    if (s->n_perm_z > 0) w += (size_t)d.cap_size * 4;              /* permutation_zs_cap */
    w += 1;                                                        /* pow_witness */
    w += 2 * (size_t)d.final_poly_len;
    w += (size_t)d.n_steps * d.cap_size * 4;
    size_t per_q = 0; int bits = d.lde_bits;
    for (int o = 0; o < d.n_oracles; o++) per_q += (size_t)d.oracle_polys[o] + (size_t)(d.lde_bits - s->cap_height) * 4;
    for (int i = 0; i < d.n_steps; i++) { bits -= d.arity[i]; per_q += 2 * ((size_t)1 << d.arity[i]) + (size_t)(bits - s->cap_height) * 4; }
    w += per_q * (size_t)s->num_queries;
    w += (size_t)s->n_pis;
    return w;
}

/* ===================================================================== synthetic inputs */
static uint64_t splitmix64(uint64_t *s) { uint64_t z = (*s += 0x9E3779B97F4A7C15ULL); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
static uint64_t rand_gl(uint64_t *s) { uint64_t x; do { x = splitmix64(s); } while (x >= GL_P); return x; }
static ofr_t rand_fr(uint64_t *s) { ofr_t x; do { for (int i = 0; i < 4; i++) x.l[i] = splitmix64(s); x.l[3] &= 0x3FFFFFFFFFFFFFFFULL; } while (fr_geq(&x, &FR_MOD)); return x; }
void orc_synth_consts(oconsts_t *k, uint64_t seed) {
    fr_init(); uint64_t s = seed;
    for (int i = 0; i < 360; i++) k->all_round_constants[i] = rand_gl(&s);
    for (int i = 0; i < 12; i++) { k->mds_circ[i] = 1 + splitmix64(&s) % 63; k->mds_diag[i] = i == 0 ? 8 : 0; } /* small, like plonky2's */
    for (int i = 0; i < 12; i++) k->fast_partial_first_round_constant[i] = rand_gl(&s);
    for (int i = 0; i < 22; i++) k->fast_partial_round_constants[i] = rand_gl(&s);
    for (int i = 0; i < 11; i++) for (int j = 0; j < 11; j++) k->fast_partial_round_initial_matrix[i][j] = rand_gl(&s);
    for (int i = 0; i < 22; i++) for (int j = 0; j < 11; j++) { k->fast_partial_round_w_hats[i][j] = rand_gl(&s); k->fast_partial_round_vs[i][j] = rand_gl(&s); }
    for (int i = 0; i < 88; i++) k->bn_c[i] = rand_fr(&s);
    for (int i = 0; i < 392; i++) k->bn_s[i] = rand_fr(&s);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { k->bn_m[i][j] = rand_fr(&s); k->bn_p[i][j] = rand_fr(&s); }
}
/* SURVEY 8(d) variant (B): uniform random GL elements / Fr hashes of the right shape */
void orc_synth_proof(const oshape_t *sh, uint64_t seed, uint64_t *w) {
    fr_init(); oderived_t d; derive(sh, &d); uint64_t s = seed; size_t k = 0;
#define PUT_HASH() do { if (sh->hash_mode == 0) { for (int _i = 0; _i < 4; _i++) w[k++] = rand_gl(&s); } else { ofr_t _h = rand_fr(&s); for (int _i = 0; _i < 4; _i++) w[k++] = _h.l[_i]; } } while (0)
#define PUT_GL(n) do { for (int _j = 0; _j < (int)(n); _j++) w[k++] = rand_gl(&s); } while (0)
    for (int i = 0; i < d.cap_size; i++) PUT_HASH();
    for (int i = 0; i < d.cap_size; i++) PUT_HASH();
    PUT_GL(2 * (2 * sh->n_cols + 2 * sh->n_perm_z + sh->n_quotient));
    if (sh->n_perm_z > 0) for (int i = 0; i < d.cap_size; i++) PUT_HASH();
    PUT_GL(1); PUT_GL(2 * d.final_poly_len);
    for (int i = 0; i < d.n_steps * d.cap_size; i++) PUT_HASH();
    for (int q = 0; q < sh->num_queries; q++) {
        for (int o = 0; o < d.n_oracles; o++) { PUT_GL(d.oracle_polys[o]); for (int j = 0; j < d.lde_bits - sh->cap_height; j++) PUT_HASH(); }
        int bits = d.lde_bits;
        for (int i = 0; i < d.n_steps; i++) { bits -= d.arity[i]; PUT_GL(2 << d.arity[i]); for (int j = 0; j < bits - sh->cap_height; j++) PUT_HASH(); }
    }
    PUT_GL(sh->n_pis);
#undef PUT_HASH
#undef PUT_GL
    if (k != orc_proof_words(sh)) { fprintf(stderr, "oracle: proof layout mismatch\n"); abort(); }
}

/* ===================================================================== WitnessChip (witness/mod.rs) + wire structs */
typedef struct { glw_t *evals; int n_evals; hw_t *sib; int n_sib; } omerkle_in_t;
typedef struct { omerkle_in_t init[3]; struct { exw_t *evals; int n; hw_t *sib; int n_sib; } step[MAX_STEPS]; } oquery_t;
typedef struct {
    hw_t *trace_cap, *quotient_cap, *perm_cap;
    exw_t *local_values, *next_values, *perm_zs, *perm_zs_next, *quotient_polys;
    glw_t pow_witness; exw_t *final_poly; hw_t *commit_caps[MAX_STEPS];
    oquery_t *queries; glw_t *pis;
} oproofw_t;

static glw_t wt_load(octx_t *c, const oshape_t *s, uint64_t v) { /* :48-51 */
    SC("load"); glw_t r;
    if (s->witness_load_range_check) r = orc_gl_load_witness(c, v);
    else { SC("load_witness"); ofr_t f = fr_from_u64(v); r = orc_load_witness(c, &f); EC(); } /* SVG-era loader */
    EC(); return r;
}
static hw_t wt_load_hash(octx_t *c, const oshape_t *s, const uint64_t *w) { /* :53-60 */
    SC("load_hash"); hw_t h; SC("load_witness");
    if (s->hash_mode == 0) gl_load_constant_array(c, w, 4, h.e); /* poseidon/hash.rs:86-96 loads as constants */
    else { ofr_t v = {{w[0], w[1], w[2], w[3]}}; h.e[0] = orc_load_witness(c, &v); h.e[1] = h.e[2] = h.e[3] = h.e[0]; } /* poseidon_bn254/hash.rs:89-98 */
    EC(); EC(); return h;
}
static hw_t *wt_load_cap(octx_t *c, const oshape_t *s, const uint64_t **w, int n) { /* :62-75 */
    SC("load_cap"); hw_t *r = (hw_t *)malloc((size_t)n * sizeof(hw_t));
    for (int i = 0; i < n; i++) { r[i] = wt_load_hash(c, s, *w); *w += 4; }
    EC(); return r;
}
static exw_t wt_load_extension(octx_t *c, const oshape_t *s, const uint64_t **w) { /* :89-97 (load_array inside) */
    SC("load_extension"); SC("load_array"); exw_t r; r.e[0] = wt_load(c, s, (*w)[0]); r.e[1] = wt_load(c, s, (*w)[1]); *w += 2; EC(); EC(); return r;
}
static exw_t *wt_load_extensions(octx_t *c, const oshape_t *s, const uint64_t **w, int n) { /* :99-109 */
    SC("load_extensions"); exw_t *r = (exw_t *)malloc((size_t)(n ? n : 1) * sizeof(exw_t));
    for (int i = 0; i < n; i++) r[i] = wt_load_extension(c, s, w);
    EC(); return r;
}
static void wt_load_proof_with_pis(octx_t *c, const oshape_t *s, const oderived_t *d, const uint64_t *w, oproofw_t *p) { /* :267-294 */
    SC("load_proof_with_pis"); SC("load_proof"); /* :235-265 */
    p->trace_cap = wt_load_cap(c, s, &w, d->cap_size);
    p->quotient_cap = wt_load_cap(c, s, &w, d->cap_size);
    SC("load_openings_set"); /* :129-147 */
    p->local_values = wt_load_extensions(c, s, &w, s->n_cols);
    p->next_values = wt_load_extensions(c, s, &w, s->n_cols);
    p->perm_zs = NULL; p->perm_zs_next = NULL;
    if (s->n_perm_z > 0) { p->perm_zs = wt_load_extensions(c, s, &w, s->n_perm_z); p->perm_zs_next = wt_load_extensions(c, s, &w, s->n_perm_z); }
    p->quotient_polys = wt_load_extensions(c, s, &w, s->n_quotient);
    EC();
    p->perm_cap = s->n_perm_z > 0 ? wt_load_cap(c, s, &w, d->cap_size) : NULL;
    SC("load_fri_proof"); /* :149-233 */
    p->pow_witness = wt_load(c, s, *w++);
    p->final_poly = (exw_t *)malloc((size_t)d->final_poly_len * sizeof(exw_t));
    for (int i = 0; i < d->final_poly_len; i++) p->final_poly[i] = wt_load_extension(c, s, &w);
    for (int i = 0; i < d->n_steps; i++) p->commit_caps[i] = wt_load_cap(c, s, &w, d->cap_size);
    p->queries = (oquery_t *)calloc((size_t)s->num_queries, sizeof(oquery_t));
    for (int q = 0; q < s->num_queries; q++) {
        oquery_t *Q = &p->queries[q];
        for (int o = 0; o < d->n_oracles; o++) {
            omerkle_in_t *m = &Q->init[o]; m->n_evals = d->oracle_polys[o]; m->n_sib = d->lde_bits - s->cap_height;
            m->evals = (glw_t *)malloc((size_t)m->n_evals * sizeof(glw_t)); m->sib = (hw_t *)malloc((size_t)(m->n_sib + 1) * sizeof(hw_t));
            for (int i = 0; i < m->n_evals; i++) m->evals[i] = wt_load(c, s, *w++);
            for (int i = 0; i < m->n_sib; i++) { m->sib[i] = wt_load_hash(c, s, w); w += 4; }
        }
        int bits = d->lde_bits;
        for (int st = 0; st < d->n_steps; st++) {
            bits -= d->arity[st];
            Q->step[st].n = 1 << d->arity[st]; Q->step[st].n_sib = bits - s->cap_height;
            Q->step[st].evals = (exw_t *)malloc((size_t)Q->step[st].n * sizeof(exw_t));
            Q->step[st].sib = (hw_t *)malloc((size_t)(Q->step[st].n_sib + 1) * sizeof(hw_t));
            for (int i = 0; i < Q->step[st].n; i++) Q->step[st].evals[i] = wt_load_extension(c, s, &w);
            for (int i = 0; i < Q->step[st].n_sib; i++) { Q->step[st].sib[i] = wt_load_hash(c, s, w); w += 4; }
        }
    }
    EC(); EC(); /* load_fri_proof, load_proof */
    p->pis = (glw_t *)malloc((size_t)(s->n_pis ? s->n_pis : 1) * sizeof(glw_t));
    for (int i = 0; i < s->n_pis; i++) p->pis[i] = wt_load(c, s, *w++);
    EC();
}
static void proofw_free(const oshape_t *s, const oderived_t *d, oproofw_t *p) {
    free(p->trace_cap); free(p->quotient_cap); free(p->perm_cap); free(p->local_values); free(p->next_values);
    free(p->perm_zs); free(p->perm_zs_next); free(p->quotient_polys); free(p->final_poly); free(p->pis);
    for (int i = 0; i < d->n_steps; i++) free(p->commit_caps[i]);
    for (int q = 0; q < s->num_queries; q++) {
        for (int o = 0; o < d->n_oracles; o++) { free(p->queries[q].init[o].evals); free(p->queries[q].init[o].sib); }
        for (int st = 0; st < d->n_steps; st++) { free(p->queries[q].step[st].evals); free(p->queries[q].step[st].sib); }
    }
    free(p->queries);
}

/* ===================================================================== ChallengerChip (challenger/mod.rs) */
typedef struct {
    const oconsts_t *k; glw_t state[SPONGE_WIDTH];
    glw_t *in; int n_in, cap_in; glw_t out[SPONGE_RATE]; int n_out;
} ochal_t;
static void ch_observe_element(ochal_t *ch, glw_t t) { /* :45-50 */
    ch->n_out = 0;
    if (ch->n_in == ch->cap_in) { ch->cap_in = ch->cap_in ? ch->cap_in * 2 : 64; ch->in = (glw_t *)realloc(ch->in, (size_t)ch->cap_in * sizeof(glw_t)); }
    ch->in[ch->n_in++] = t;
}
static void ch_observe_hash(octx_t *c, ochal_t *ch, int mode, hw_t h) { /* :59-63 */
    SC("observe_hash"); glw_t v[5]; int n = hw_to_goldilocks_vec(c, mode, h, v);
    for (int i = 0; i < n; i++) ch_observe_element(ch, v[i]);
    EC();
}
static void ch_observe_cap(octx_t *c, ochal_t *ch, int mode, const hw_t *cap, int n) { SC("observe_cap"); for (int i = 0; i < n; i++) ch_observe_hash(c, ch, mode, cap[i]); EC(); } /* :65-74 */
static void ch_observe_extension_elements(ochal_t *ch, const exw_t *e, int n) { for (int i = 0; i < n; i++) { ch_observe_element(ch, e[i].e[0]); ch_observe_element(ch, e[i].e[1]); } } /* :76-84 */
static void ch_absorb_buffered_inputs(octx_t *c, ochal_t *ch) { /* :260-277 */
    SC("absorb_buffered_inputs");
    if (ch->n_in == 0) { EC(); return; }
    pg_absorb_goldilocks(c, ch->k, ch->state, ch->in, ch->n_in);
    memcpy(ch->out, ch->state, SPONGE_RATE * sizeof(glw_t)); ch->n_out = SPONGE_RATE; /* squeeze_goldilocks */
    ch->n_in = 0; EC();
}
static glw_t ch_get_challenge(octx_t *c, ochal_t *ch) { /* :92-108 */
    SC("get_challenge");
    ch_absorb_buffered_inputs(c, ch);
    if (ch->n_out == 0) { pg_permute(c, ch->k, ch->state); memcpy(ch->out, ch->state, SPONGE_RATE * sizeof(glw_t)); ch->n_out = SPONGE_RATE; }
    glw_t r = ch->out[--ch->n_out];
    EC(); return r;
}
static void ch_get_n_challenges(octx_t *c, ochal_t *ch, int n, glw_t *out) { SC("get_n_challenges"); for (int i = 0; i < n; i++) out[i] = ch_get_challenge(c, ch); EC(); } /* :110-117 */
static exw_t ch_get_extension_challenge(octx_t *c, ochal_t *ch) { SC("get_extension_challenge"); exw_t r; ch_get_n_challenges(c, ch, 2, r.e); EC(); return r; } /* :119-126 */

typedef struct { exw_t fri_alpha; exw_t fri_betas[MAX_STEPS]; glw_t fri_pow_response; glw_t *fri_query_indices; exw_t stark_zeta; } ochallenges_t;

/* ===================================================================== FriChip (fri/mod.rs) */
typedef struct { exw_t point; int n_polys; int oracle_index[16]; int poly_index[16]; } obatch_t;

static exw_t fri_combine_initial(octx_t *c, const oshape_t *s, const obatch_t *batches, const oquery_t *Q, exw_t alpha, glw_t subgroup_x, const exw_t *reduced_openings) { /* :169-220 */
    SC("combine_initial"); (void)s;
    exw_t sx = ex_load_base(c, subgroup_x);
    exw_t sum = ex_load_zero(c);
    for (int b = 0; b < 2; b++) {
        const obatch_t *B = &batches[b]; exw_t evals[16];
        for (int i = 0; i < B->n_polys; i++) evals[i] = ex_load_base(c, Q->init[B->oracle_index[i]].evals[B->poly_index[i]]);
        exw_t reduced_evals = ex_reduce_with_powers(c, evals, B->n_polys, alpha);
        exw_t numerator = ex_sub(c, reduced_evals, reduced_openings[b]);
        exw_t denominator = ex_sub(c, sx, B->point);
        exw_t denominator_inv = ex_inv(c, denominator);
        exw_t alpha_shift = ex_exp_u64(c, alpha, (uint64_t)B->n_polys);
        sum = ex_mul(c, alpha_shift, sum);
        sum = ex_mul_add(c, numerator, denominator_inv, sum);
    }
    EC(); return sum;
}
static exw_t fri_interpolate_coset(octx_t *c, glw_t coset_shift, const exw_t *values, int n, exw_t evaluation_point) { /* :222-283 */
    SC("interpolate_coset");
    int arity_bits = 0; while ((1 << arity_bits) < n) arity_bits++;
    exw_t shifted = ex_scalar_div(c, evaluation_point, coset_shift);
    uint64_t g = glf_primitive_root_of_unity(arity_bits), dom[64]; exw_t domain[64]; glw_t bw[64]; exw_t wv[64];
    dom[0] = 1; for (int i = 1; i < n; i++) dom[i] = glf_mul(dom[i - 1], g); /* two_adic_subgroup */
    for (int i = 0; i < n; i++) { gle_t e = {{dom[i], 0}}; domain[i] = ex_load_constant(c, e); }
    for (int i = 0; i < n; i++) { /* barycentric_weights: 1/prod_{j!=i}(x_i-x_j) */
        uint64_t pr = 1; for (int j = 0; j < n; j++) if (j != i) pr = glf_mul(pr, glf_sub(dom[i], dom[j]));
        bw[i] = orc_gl_load_constant(c, glf_inv(pr));
    }
    for (int i = 0; i < n; i++) wv[i] = ex_scalar_mul(c, values[i], bw[i]);
    exw_t eval = ex_load_zero(c), tpp = ex_load_one(c);
    for (int i = 0; i < n; i++) {
        exw_t term = ex_sub(c, shifted, domain[i]);
        exw_t next_tpp = ex_mul(c, tpp, term);
        exw_t tmp1 = ex_mul(c, eval, term);
        exw_t tmp2 = ex_mul(c, wv[i], tpp);
        eval = ex_add(c, tmp1, tmp2); tpp = next_tpp;
    }
    EC(); return eval;
}
static exw_t fri_compute_evaluation(octx_t *c, glw_t x, const oav_t *within_bits, int arity_bits, const exw_t *evals_in, exw_t beta) { /* :285-322 */
    SC("compute_evaluation");
    int arity = 1 << arity_bits;
    uint64_t g = glf_primitive_root_of_unity(arity_bits), g_inv = glf_exp(g, (uint64_t)arity - 1);
    exw_t evals[64];
    for (int i = 0; i < arity; i++) { int r = 0; for (int b = 0; b < arity_bits; b++) if (i & (1 << b)) r |= 1 << (arity_bits - 1 - b); evals[r] = evals_in[i]; } /* reverse_index_bits_in_place */
    oav_t rev[8]; for (int i = 0; i < arity_bits; i++) rev[i] = within_bits[arity_bits - 1 - i];
    glw_t start = orc_gl_exp_from_bits_const_base(c, g_inv, rev, arity_bits);
    glw_t coset_start = orc_gl_mul(c, start, x);
    exw_t r = fri_interpolate_coset(c, coset_start, evals, arity, beta);
    EC(); return r;
}
static exw_t fri_eval_scalar(octx_t *c, const exw_t *poly, int n, glw_t point) { /* :324-335 */
    SC("eval_scalar"); exw_t p = ex_load_base(c, point); exw_t r = ex_reduce_with_powers(c, poly, n, p); EC(); return r;
}
static void fri_verify_query_round(octx_t *c, const oshape_t *s, const oderived_t *d, const oconsts_t *k, const obatch_t *batches, const ochallenges_t *chal,
                                   const exw_t *reduced_openings, hw_t **initial_caps, const oproofw_t *p, glw_t x_index, const oquery_t *Q) { /* :337-444 */
    SC("verify_query_round");
    int n_log = d->lde_bits, mode = s->hash_mode;
    oav_t bits64[64];
    gl_num_to_bits(c, x_index, 64, bits64);
    oav_t *x_index_bits = bits64; int nb = n_log; /* truncate */
    glw_t cap_index = gl_bits_to_num(c, x_index_bits + nb - s->cap_height, s->cap_height);
    SC("verify_initial_proof"); /* :147-167 */
    for (int o = 0; o < d->n_oracles; o++)
        mk_verify_proof_to_cap_with_cap_index(c, k, mode, Q->init[o].evals, Q->init[o].n_evals, x_index_bits, nb, cap_index, initial_caps[o], d->cap_size, Q->init[o].sib, Q->init[o].n_sib);
    EC();
    glw_t subgroup_x;
    {
        glw_t g = orc_gl_load_constant(c, 7); /* coset_shift */
        uint64_t phi = glf_primitive_root_of_unity(n_log);
        oav_t rev[64]; for (int i = 0; i < nb; i++) rev[i] = x_index_bits[nb - 1 - i];
        glw_t phiw = orc_gl_exp_from_bits_const_base(c, phi, rev, nb);
        subgroup_x = orc_gl_mul(c, g, phiw);
    }
    exw_t old_eval = fri_combine_initial(c, s, batches, Q, chal->fri_alpha, subgroup_x, reduced_openings);
    for (int i = 0; i < d->n_steps; i++) {
        int ab = d->arity[i];
        const exw_t *evals = Q->step[i].evals;
        oav_t *coset_index_bits = x_index_bits + ab; int ncb = nb - ab;
        glw_t within = gl_bits_to_num(c, x_index_bits, ab);
        exw_t new_eval = ex_select_from_idx(c, evals, 1 << ab, within);
        ex_assert_equal(c, new_eval, old_eval, 1);
        old_eval = fri_compute_evaluation(c, subgroup_x, x_index_bits, ab, evals, chal->fri_betas[i]);
        glw_t leaf[128]; for (int j = 0; j < (1 << ab); j++) { leaf[2 * j] = evals[j].e[0]; leaf[2 * j + 1] = evals[j].e[1]; }
        mk_verify_proof_to_cap_with_cap_index(c, k, mode, leaf, 2 << ab, coset_index_bits, ncb, cap_index, p->commit_caps[i], d->cap_size, Q->step[i].sib, Q->step[i].n_sib);
        subgroup_x = orc_gl_exp_power_of_2(c, subgroup_x, ab);
        x_index_bits = coset_index_bits; nb = ncb;
    }
    exw_t eval = fri_eval_scalar(c, p->final_poly, d->final_poly_len, subgroup_x);
    ex_assert_equal(c, eval, old_eval, 1);
    EC();
}

/* ===================================================================== StarkChip (stark/mod.rs) + driver */
int orc_verify_stark(octx_t *c, const oshape_t *s, const oconsts_t *k, const uint64_t *w) {
    oderived_t d; derive(s, &d); oproofw_t p; ochal_t ch; ochallenges_t chal;
    memset(&ch, 0, sizeof(ch)); ch.k = k; memset(&chal, 0, sizeof(chal));
    int mode = s->hash_mode;
    /* stark/mod.rs:497-499: state = permutation_chip.load_zero(ctx) (hash/poseidon/permutation.rs:264-268) */
    { uint64_t z[SPONGE_WIDTH] = {0}; gl_load_constant_array(c, z, SPONGE_WIDTH, ch.state); }
    wt_load_proof_with_pis(c, s, &d, w, &p);                                   /* stark/mod.rs:506 */
    SC("verify_proof");                                                        /* stark/mod.rs:346-374 */
    SC("get_stark_challenges");                                                /* challenger/mod.rs:167-222 */
    ch_observe_cap(c, &ch, mode, p.trace_cap, d.cap_size);
    if (s->n_perm_z > 0) {
        SC("get_n_permutation_challenge_sets");                                /* :246-256 (num_challenges, batch_size) */
        for (int set = 0; set < s->perm_batch_size; set++) {
            SC("get_permutation_challenge_set");
            for (int i = 0; i < s->num_challenges; i++) { SC("get_permutation_challenge"); ch_get_challenge(c, &ch); ch_get_challenge(c, &ch); EC(); }
            EC();
        }
        EC();
        ch_observe_cap(c, &ch, mode, p.perm_cap, d.cap_size);
    }
    { glw_t alphas[16]; ch_get_n_challenges(c, &ch, s->num_challenges, alphas); }
    ch_observe_cap(c, &ch, mode, p.quotient_cap, d.cap_size);
    chal.stark_zeta = ch_get_extension_challenge(c, &ch);
    /* observe_openings(to_fri_openings()) stark/mod.rs:48-69 */
    int nz = s->n_cols + s->n_perm_z + s->n_quotient, nzn = s->n_cols + s->n_perm_z;
    exw_t *zeta_vals = (exw_t *)malloc((size_t)nz * sizeof(exw_t)), *zeta_next_vals = (exw_t *)malloc((size_t)nzn * sizeof(exw_t));
    { int t = 0; for (int i = 0; i < s->n_cols; i++) zeta_vals[t++] = p.local_values[i]; for (int i = 0; i < s->n_perm_z; i++) zeta_vals[t++] = p.perm_zs[i]; for (int i = 0; i < s->n_quotient; i++) zeta_vals[t++] = p.quotient_polys[i];
      t = 0; for (int i = 0; i < s->n_cols; i++) zeta_next_vals[t++] = p.next_values[i]; for (int i = 0; i < s->n_perm_z; i++) zeta_next_vals[t++] = p.perm_zs_next[i]; }
    ch_observe_extension_elements(&ch, zeta_vals, nz); ch_observe_extension_elements(&ch, zeta_next_vals, nzn);
    SC("get_fri_challenges");                                                  /* challenger/mod.rs:128-165 */
    chal.fri_alpha = ch_get_extension_challenge(c, &ch);
    for (int i = 0; i < d.n_steps; i++) { ch_observe_cap(c, &ch, mode, p.commit_caps[i], d.cap_size); chal.fri_betas[i] = ch_get_extension_challenge(c, &ch); }
    ch_observe_extension_elements(&ch, p.final_poly, d.final_poly_len);
    ch_observe_element(&ch, p.pow_witness);
    chal.fri_pow_response = ch_get_challenge(c, &ch);
    chal.fri_query_indices = (glw_t *)malloc((size_t)s->num_queries * sizeof(glw_t));
    for (int i = 0; i < s->num_queries; i++) chal.fri_query_indices[i] = ch_get_challenge(c, &ch);
    EC(); EC();
    SC("verify_proof_with_challenges");                                        /* stark/mod.rs:230-344 */
    hw_t *merkle_caps[3]; int nc = 0; merkle_caps[nc++] = p.trace_cap; if (s->n_perm_z > 0) merkle_caps[nc++] = p.perm_cap; merkle_caps[nc++] = p.quotient_cap;
    /* fri_instance_info stark/mod.rs:144-200 (no #[count]) */
    obatch_t batches[2]; memset(batches, 0, sizeof(batches));
    { int t = 0, o = 0;
      for (int i = 0; i < s->n_cols; i++) { batches[0].oracle_index[t] = o; batches[0].poly_index[t++] = i; }
      if (s->n_perm_z > 0) { o++; for (int i = 0; i < s->n_perm_z; i++) { batches[0].oracle_index[t] = o; batches[0].poly_index[t++] = i; } }
      o++; for (int i = 0; i < s->n_quotient; i++) { batches[0].oracle_index[t] = o; batches[0].poly_index[t++] = i; }
      batches[0].n_polys = t; t = 0;
      for (int i = 0; i < s->n_cols; i++) { batches[1].oracle_index[t] = 0; batches[1].poly_index[t++] = i; }
      for (int i = 0; i < s->n_perm_z; i++) { batches[1].oracle_index[t] = 1; batches[1].poly_index[t++] = i; }
      batches[1].n_polys = t; }
    batches[0].point = chal.stark_zeta;
    { gle_t gv = {{glf_primitive_root_of_unity(s->degree_bits), 0}}; exw_t g = ex_load_constant(c, gv); batches[1].point = ex_mul(c, g, chal.stark_zeta); }
    SC("verify_fri_proof");                                                    /* fri/mod.rs:446-502 */
    SC("verify_proof_of_work"); c->semantic++; orc_range_check(c, chal.fri_pow_response, 64 - s->pow_bits); c->semantic--; EC(); /* :130-145 */
    exw_t reduced_openings[2];
    SC("from_os_and_alpha");                                                   /* :45-62 */
    reduced_openings[0] = ex_reduce_with_powers(c, zeta_vals, nz, chal.fri_alpha);
    reduced_openings[1] = ex_reduce_with_powers(c, zeta_next_vals, nzn, chal.fri_alpha);
    EC();
    for (int q = 0; q < s->num_queries; q++)
        fri_verify_query_round(c, s, &d, k, batches, &chal, reduced_openings, merkle_caps, &p, chal.fri_query_indices[q], &p.queries[q]);
    EC(); EC(); EC();
    free(zeta_vals); free(zeta_next_vals); free(chal.fri_query_indices); free(ch.in);
    proofw_free(s, &d, &p);
    return c->failed ? -1 : 0;
}

/* value-domain exports for tests */
uint64_t orc_glf_mul(uint64_t a, uint64_t b) { return glf_mul(a, b); }
uint64_t orc_glf_inv(uint64_t a) { return glf_inv(a); }
uint64_t orc_glf_exp(uint64_t a, uint64_t e) { return glf_exp(a, e); }
uint64_t orc_glf_primitive_root_of_unity(int bits) { return glf_primitive_root_of_unity(bits); }
void orc_fr_mul(const ofr_t *a, const ofr_t *b, ofr_t *out) { fr_init(); *out = fr_mul(a, b); }
void orc_fr_inv(const ofr_t *a, ofr_t *out) { fr_init(); *out = fr_inv(a); }
void orc_fr_modulus(ofr_t *out) { *out = FR_MOD; }

/* ===================================================================== keygen metadata + FlexGate column layout
 * halo2-lib `community-edition` (not in /root/reference) semantics [R]:
 *  - gates/flex_gate/threads/single_phase.rs assign_with_constraints::<F, ROTATIONS = 4>: walk ctx.advice down column `gate_index`;
 *    `if (q && row_offset + ROTATIONS > max_rows) || row_offset >= max_rows - 1 { break_points.push(row_offset); row_offset = 0;
 *    gate_index += 1; assign the same value again at (gate_index, 0) + copy constraint }`, then enable q at the current row.
 *  - assign_witnesses: the same walk driven by the pinned break points.
 *  - lookup advice: cells_to_lookup copied in order down lookup columns of max_rows rows.
 *  max_rows = 2^k - unusable_rows (base_test: 9). */
uint64_t orc_num_gates(const octx_t *c) { uint64_t n = 0; for (size_t i = 0; i < c->n && i < c->selcap; i++) n += c->selector[i]; return n; }
void orc_selector_bitmap(const octx_t *c, uint8_t *out) {
    memset(out, 0, (c->n + 7) / 8);
    for (size_t i = 0; i < c->n && i < c->selcap; i++) if (c->selector[i]) out[i / 8] |= (uint8_t)(1u << (i & 7));
}
uint64_t orc_num_equalities(const octx_t *c) { return c->neq; }
void orc_equalities(const octx_t *c, uint64_t *pairs) { for (size_t i = 0; i < c->neq; i++) { pairs[2 * i] = (uint64_t)c->eq[i].a; pairs[2 * i + 1] = (uint64_t)c->eq[i].b; } }
uint64_t orc_num_const_equalities(const octx_t *c) { return c->nceq; }
void orc_const_equalities(const octx_t *c, uint64_t *cells, ofr_t *values) { for (size_t i = 0; i < c->nceq; i++) { cells[i] = (uint64_t)c->ceq[i].cell; values[i] = c->ceq[i].c; } }
uint64_t orc_num_lookups(const octx_t *c) { return c->nlookup; }
void orc_lookup_cells(const octx_t *c, uint64_t *out) { for (size_t i = 0; i < c->nlookup; i++) out[i] = (uint64_t)c->lookup[i].cell; }
uint64_t orc_break_points(const octx_t *c, int k, int unusable_rows, uint64_t *out, uint64_t cap) {
    const uint64_t max_rows = ((uint64_t)1 << k) - (uint64_t)unusable_rows; uint64_t row = 0, n = 0;
    for (size_t i = 0; i < c->n; i++) {
        int q = i < c->selcap ? c->selector[i] : 0;
        if ((q && row + 4 > max_rows) || row >= max_rows - 1) { if (out && n < cap) out[n] = row; n++; row = 0; }
        row++;
    }
    return n;
}
/* columns[(n_bp + 1)][2^k] from the pinned break points (assign_witnesses); unassigned rows are zero */
void orc_layout_columns(const octx_t *c, const uint64_t *bp, uint64_t n_bp, int k, ofr_t *out) {
    const uint64_t rows = (uint64_t)1 << k; uint64_t col = 0, row = 0, nb = 0;
    memset(out, 0, (size_t)((n_bp + 1) * rows) * sizeof(ofr_t));
    for (size_t i = 0; i < c->n; i++) {
        out[col * rows + row] = c->advice[i];
        if (nb < n_bp && bp[nb] == row) { nb++; row = 0; col++; out[col * rows + row] = c->advice[i]; }
        row++;
    }
}
uint64_t orc_layout_lookup_columns(const octx_t *c, int k, int unusable_rows, ofr_t *out) {
    const uint64_t rows = (uint64_t)1 << k, max_rows = rows - (uint64_t)unusable_rows, ncols = (c->nlookup + max_rows - 1) / max_rows;
    if (!out) return ncols;
    memset(out, 0, (size_t)(ncols * rows) * sizeof(ofr_t));
    uint64_t col = 0, row = 0;
    for (size_t i = 0; i < c->nlookup; i++) { if (row >= max_rows) { row = 0; col++; } out[col * rows + row] = c->advice[c->lookup[i].cell]; row++; }
    return ncols;
}

/* ===================================================================== native FRI prover (valid synthetic proofs) */
#include "prover.inc"
