// replay.hip — record and replay (include/h2w.h 2d): h2w_plan_from_trace lowers the tape an eager context recorded in trace mode (trace.h: one
// entry per level-1 / level-2 call of ONE run of the reference's gadget through the operator API, field/native.rs:28-193 and
// field/goldilocks/base.rs:61-399) to a device program; h2w_fri_witness_batch on such a plan replays it on a batch of other proofs.
//
//   segments   the instances of the scopes the caller names as parallel (fri/mod.rs:488-501 "verify_query_round", merkle/mod.rs:57-78
//              "verify_proof_to_cap_with_cap_index") + the root; the claim is VERIFIED on the tape: an op may read values of its own segment and of
//              the segments enclosing it (produced before it started), nothing else - which also says that nothing outside reads what a parallel
//              segment computes.  Isomorphic instances (word-for-word equal lowered tapes) share one TEMPLATE.
//   device     one lane per (proof, instance), one kernel launch per template, depth by depth (a segment needs its ancestors' values).  The lanes of a
//              wavefront run the same template in lockstep: the tape is read with scalar loads, branches are uniform.  An op computes its value
//              with the arithmetic of the value backend (valbackend.h: the same templates the batched kernels use) and appends its cells - block
//              records for the Goldilocks-level templates (expanded by expand_fast afterwards), direct cells for the rest.  Values live in a
//              per-template store [slot][lane] (coalesced across the lanes), static constants in the tape.
// Which op becomes which template is decided on STATIC widths (a value is provably below 2^64 when a Goldilocks-level op, a bit decomposition, a
// one-word input ... produced it), never on the traced values: the layout of the records must not depend on the proof.
#include <hip/hip_runtime.h>
#include <unordered_map>
#include <map>
#include <vector>
#include <string>
#include <cstring>
#include "plan.h"
#include "trace.h"

struct h2w_ctx;
namespace h2w {
Trace *ctx_trace(h2w_ctx *); int ctx_lookup_bits(const h2w_ctx *); uint64_t ctx_num_cells(const h2w_ctx *);      // eager.cpp

enum { W64 = 0, W128 = 1, WFR = 2 };
enum { RK_LOCAL = 0, RK_IMPORT = 1, RK_LIT64 = 2, RK_INPUT = 3, RK_LITFR = 4 };
HD uint32_t mkref(int kind, int width, uint32_t idx) { return ((uint32_t)kind << 29) | ((uint32_t)width << 27) | idx; }
HD int ref_kind(uint32_t r) { return (int)(r >> 29); }
HD int ref_width(uint32_t r) { return (int)((r >> 27) & 3); }
HD uint32_t ref_idx(uint32_t r) { return r & ((1u << 27) - 1); }
constexpr int SLOTS_OF[3] = {1, 2, 4};
// device ops; word 0 = op | n << 8 | aux << 16, then operand refs, then (ops with results) the first output slot
enum { DOP_END = 0, DOP_SKIP, DOP_CONST1, DOP_FRCELL, DOP_LOADW, DOP_LOADW_DIV, DOP_LOADW_EXTINV, DOP_GLOP, DOP_GATE, DOP_REDUCE, DOP_CLT,
       DOP_FR_ADD, DOP_FR_MUL, DOP_FR_MULADD, DOP_SELECT, DOP_FR_SELECT, DOP_IDX2IND, DOP_SELIND, DOP_FR_SELIND, DOP_NUM2BITS, DOP_BITS2NUM,
       DOP_DECOMP565, DOP_LIMBS2NUM, DOP_RANGE };
constexpr int MAX_TMPL = 48;
constexpr uint32_t NO_SLOT = 0xffffffffu;

struct InstD { uint64_t cell0, rec0; uint32_t imp0, in0; };
struct ImpD { uint32_t tmpl, inst, slot; };
struct TmplD { uint32_t tape0, nslots, ninst, inst0, depth; };

struct ReplayArgs {
    const uint32_t *tape; const InstD *insts; const ImpD *imps; const uint32_t *inputs; const uint64_t *pool64; const fr_t *poolfr;
    const uint64_t *proofs; uint64_t proof_words; rec_t *recs; uint64_t rec_stride; fr_t *out; uint64_t cell_stride; uint64_t *vals; uint32_t *status;
    const uint16_t *ncells; const fr_t *inv_pos, *inv_neg; FrParams P; int L; uint32_t nproofs;
    uint32_t tape0, nslots, ninst, inst0, tmpl;
    uint64_t tbase[MAX_TMPL]; uint32_t tninst[MAX_TMPL];
};

struct TracedPlan {
    std::vector<TmplD> tmpls; uint64_t total_slot_lanes = 0;      // sum over templates of nslots * ninst: u64 elements of the value store per proof
    uint32_t *d_tape = nullptr; InstD *d_insts = nullptr; ImpD *d_imps = nullptr; uint32_t *d_inputs = nullptr; uint64_t *d_pool64 = nullptr; fr_t *d_poolfr = nullptr;
    uint64_t n_ops = 0, n_segments = 0;
};

// ------------------------------------------------------------------------------------------------------------------- device
typedef ValBackend<DevSink> RB;
struct Lane {
    const ReplayArgs &R; uint32_t p, inst, g; const InstD &I; uint64_t *vals; uint64_t L_;      // vals: this template's store, L_: its lanes per slot
    const uint64_t *proof;
    __device__ __forceinline__ uint64_t ld(uint32_t slot) const { return H2W_GLOAD64(vals + (uint64_t)slot * L_ + g); }
    __device__ __forceinline__ void st(uint32_t slot, uint64_t v) const { H2W_GSTORE64(vals + (uint64_t)slot * L_ + g, v); }
    __device__ uint64_t imp(uint32_t k, int j) const {
        const ImpD d = R.imps[I.imp0 + k];
        const uint64_t Lt = (uint64_t)R.nproofs * R.tninst[d.tmpl];
        return H2W_GLOAD64(R.vals + R.tbase[d.tmpl] + (uint64_t)(d.slot + j) * Lt + (uint64_t)p * R.tninst[d.tmpl] + d.inst);
    }
    __device__ uint64_t get64(uint32_t r, int j = 0) const {
        const uint32_t i = ref_idx(r);
        switch (ref_kind(r)) {
            case RK_LOCAL: return ld(i + j);
            case RK_IMPORT: return imp(i, j);
            case RK_LIT64: return H2W_CLOAD64(R.pool64 + i + j);
            case RK_INPUT: return g_load_u64(proof + R.inputs[I.in0 + i] + j);
            default: return H2W_CLOAD64(reinterpret_cast<const uint64_t *>(R.poolfr + i) + j);
        }
    }
    __device__ fr_t getfr(uint32_t r) const {
        fr_t v = fr_zero(); const int w = ref_kind(r) == RK_LITFR ? 4 : (ref_kind(r) == RK_INPUT ? (ref_width(r) == WFR ? 4 : 1) : SLOTS_OF[ref_width(r)]);
        for (int j = 0; j < w; j++) v.l[j] = get64(r, j);
        return v;
    }
    __device__ __forceinline__ void put64(uint32_t slot, uint64_t v) const { if (slot != NO_SLOT) st(slot, v); }
    __device__ void putfr(uint32_t slot, const fr_t &v) const { if (slot != NO_SLOT) for (int j = 0; j < 4; j++) st(slot + j, v.l[j]); }
};
__device__ __forceinline__ uint32_t tw(const uint32_t *tape, uint32_t i) { return *(const __attribute__((address_space(4))) uint32_t *)(tape + i); }

__global__ __launch_bounds__(64) void k_replay(ReplayArgs R) {
    const uint32_t g = blockIdx.x * 64 + threadIdx.x;
    if (g >= R.nproofs * R.ninst) return;
    const uint32_t p = g / R.ninst, inst = g % R.ninst;
    const InstD &I = R.insts[R.inst0 + inst];
    DevSink sink; sink.recs = R.recs + (uint64_t)p * R.rec_stride; sink.out = R.out + (uint64_t)p * R.cell_stride; sink.ncells = R.ncells; sink.cc.init(ColMap{nullptr, 0, 0});
    sink.nrec = I.rec0; sink.cell_off = I.cell0;
    ValCfg cfg; cfg.proof = R.proofs + (uint64_t)p * R.proof_words; cfg.mode = 1; cfg.L = R.L; cfg.P = R.P; cfg.inv_pos = R.inv_pos; cfg.inv_neg = R.inv_neg; cfg.st = nullptr;
    cfg.split = false; cfg.split_bn = false; cfg.load_items = nullptr; cfg.n_load_items = 0; cfg.load_nrec = cfg.load_ncell = 0; cfg.n_cap_items = 0; cfg.fri = nullptr;
    RB be(sink, cfg, true);
    const Lane ln{R, p, inst, g, I, R.vals + R.tbase[R.tmpl], (uint64_t)R.nproofs * R.ninst, cfg.proof};
    const uint32_t *tape = R.tape;
    uint64_t ta[64], tb[64];
    uint32_t pc = R.tape0;
    for (;;) {
        const uint32_t h = tw(tape, pc); const uint32_t op = h & 0xff, n = (h >> 8) & 0xff, aux = h >> 16;
        if (op == DOP_END) break;
        switch (op) {
            case DOP_SKIP: { const uint64_t nr = ((uint64_t)tw(tape, pc + 2) << 32) | tw(tape, pc + 1), nc = ((uint64_t)tw(tape, pc + 4) << 32) | tw(tape, pc + 3); sink.skip(nr, nc); pc += 5; break; }
            case DOP_CONST1: { const uint64_t v = ln.get64(tw(tape, pc + 1)); sink.rec(T_CONST1, v, 0, 0, 0); ln.put64(tw(tape, pc + 2), v); pc += 3; break; }
            case DOP_FRCELL: { const fr_t v = ln.getfr(tw(tape, pc + 1)); be.cell(v); ln.putfr(tw(tape, pc + 2), v); pc += 3; break; }
            case DOP_LOADW: { const uint64_t v = ln.get64(tw(tape, pc + 1)); sink.rec(T_LOADW, v, 0, 0, 0); ln.put64(tw(tape, pc + 2), v); pc += 3; break; }
            case DOP_LOADW_DIV: {      // the hint of GoldilocksChip::div (base.rs:371-393): a / b; b == 0: status 1, the cells of the op on 1
                const uint64_t a = ln.get64(tw(tape, pc + 1)); uint64_t b = ln.get64(tw(tape, pc + 2));
                if (b == 0) { be.fail(1); b = 1; }
                const uint64_t v = gl_mul(a, gl_inv(b)); sink.rec(T_LOADW, v, 0, 0, 0); ln.put64(tw(tape, pc + 3), v); pc += 4; break;
            }
            case DOP_LOADW_EXTINV: {   // extension.rs:320-340
                gle_t a; a.c[0] = ln.get64(tw(tape, pc + 1)); a.c[1] = ln.get64(tw(tape, pc + 2));
                if (a.c[0] == 0 && a.c[1] == 0) { be.fail(2); a.c[0] = 1; }
                const uint64_t v = gle_inv(a).c[aux & 1]; sink.rec(T_LOADW, v, 0, 0, 0); ln.put64(tw(tape, pc + 3), v); pc += 4; break;
            }
            case DOP_GLOP: { const uint64_t A = ln.get64(tw(tape, pc + 1)), B = ln.get64(tw(tape, pc + 2)), C = ln.get64(tw(tape, pc + 3)); sink.rec((int)aux, A, B, C, 0); ln.put64(tw(tape, pc + 4), gl_reduce128((u128)A * B + C)); pc += 5; break; }
            case DOP_GATE: { const uint64_t A = ln.get64(tw(tape, pc + 1)), B = ln.get64(tw(tape, pc + 2)), C = ln.get64(tw(tape, pc + 3)); sink.rec((int)aux, A, B, C, 0);
                             const u128 v = (u128)A * B + C; const uint32_t o = tw(tape, pc + 4); if (o != NO_SLOT) { ln.st(o, (uint64_t)v); ln.st(o + 1, (uint64_t)(v >> 64)); } pc += 5; break; }
            case DOP_REDUCE: { const uint32_t r = tw(tape, pc + 1); const uint64_t lo = ln.get64(r, 0), hi = ln.get64(r, 1); sink.rec(T_REDUCE, lo, hi, 0, 0); ln.put64(tw(tape, pc + 2), gl_reduce128(((u128)hi << 64) | lo)); pc += 3; break; }
            case DOP_CLT: { sink.rec(T_CLT_SAFE, ln.get64(tw(tape, pc + 1)), 0, 0, 0); pc += 2; break; }
            case DOP_FR_ADD: { const fr_t v = be.fr_add(ln.getfr(tw(tape, pc + 1)), ln.getfr(tw(tape, pc + 2))); ln.putfr(tw(tape, pc + 3), v); pc += 4; break; }
            case DOP_FR_MUL: { const fr_t v = be.fr_mul(ln.getfr(tw(tape, pc + 1)), ln.getfr(tw(tape, pc + 2))); ln.putfr(tw(tape, pc + 3), v); pc += 4; break; }
            case DOP_FR_MULADD: { const fr_t v = be.fr_mul_add(ln.getfr(tw(tape, pc + 1)), ln.getfr(tw(tape, pc + 2)), ln.getfr(tw(tape, pc + 3))); ln.putfr(tw(tape, pc + 4), v); pc += 5; break; }
            case DOP_SELECT: { const uint64_t v = be.select(ln.get64(tw(tape, pc + 1)), ln.get64(tw(tape, pc + 2)), ln.get64(tw(tape, pc + 3))); ln.put64(tw(tape, pc + 4), v); pc += 5; break; }
            case DOP_FR_SELECT: { const fr_t v = be.fr_select(ln.getfr(tw(tape, pc + 1)), ln.getfr(tw(tape, pc + 2)), ln.get64(tw(tape, pc + 3))); ln.putfr(tw(tape, pc + 4), v); pc += 5; break; }
            case DOP_IDX2IND: { be.idx_to_indicator(ln.get64(tw(tape, pc + 1)), (int)n, ta); const uint32_t o = tw(tape, pc + 2); for (uint32_t i = 0; i < n; i++) ln.st(o + i, ta[i]); pc += 3; break; }
            case DOP_SELIND: { for (uint32_t i = 0; i < n; i++) { ta[i] = ln.get64(tw(tape, pc + 1 + i)); tb[i] = ln.get64(tw(tape, pc + 1 + n + i)); }
                               ln.put64(tw(tape, pc + 1 + 2 * n), be.select_by_indicator(ta, 1, tb, (int)n)); pc += 2 + 2 * n; break; }
            case DOP_FR_SELIND: {      // GateChip::select_by_indicator on native values: [0, a0, ind0, s0, a1, ind1, s1, ...] (gates at 3 i)
                fr_t sum = fr_zero(); if (n > 0) be.G(); be.cell64(0);
                for (uint32_t i = 0; i < n; i++) { const fr_t a = ln.getfr(tw(tape, pc + 1 + i)); const uint64_t ind = ln.get64(tw(tape, pc + 1 + n + i)); if (ind) sum = h2w::fr_add(sum, a); be.cell(a); be.cell64(ind); if (i + 1 < n) be.G(); be.cell(sum); }
                ln.putfr(tw(tape, pc + 1 + 2 * n), sum); pc += 2 + 2 * n; break;
            }
            case DOP_NUM2BITS: { be.num_to_bits(ln.get64(tw(tape, pc + 1)), (int)n, ta); const uint32_t o = tw(tape, pc + 2); for (uint32_t i = 0; i < n; i++) ln.st(o + i, ta[i]); pc += 3; break; }
            case DOP_BITS2NUM: { for (uint32_t i = 0; i < n; i++) ta[i] = ln.get64(tw(tape, pc + 1 + i)); ln.put64(tw(tape, pc + 1 + n), be.bits_to_num(ta, (int)n)); pc += 2 + n; break; }
            case DOP_DECOMP565: { be.decompose_le_56_5(ln.getfr(tw(tape, pc + 1)), ta); const uint32_t o = tw(tape, pc + 2); for (int i = 0; i < 5; i++) ln.st(o + i, ta[i]); pc += 3; break; }
            case DOP_LIMBS2NUM: { for (uint32_t i = 0; i < n; i++) ta[i] = ln.get64(tw(tape, pc + 1 + i)); ln.putfr(tw(tape, pc + 1 + n), be.limbs_to_num(ta, (int)n)); pc += 2 + n; break; }
            case DOP_RANGE: { be.range_check(ln.get64(tw(tape, pc + 1)), (int)aux); pc += 2; break; }
            default: pc = R.tape0; be.fail(99); goto done;      // (unreachable: the lowering emits nothing else)
        }
    }
done:
    if (be.status) atomicCAS(&R.status[p], 0u, be.status);
}

// ------------------------------------------------------------------------------------------------------------------- host: lowering
struct ValInfo { uint32_t seg, slot; uint8_t width, is_static; uint32_t lit; };
struct SegInfo {
    int parent = -1, depth = 0; uint32_t name = 0;
    std::vector<uint32_t> tape; uint32_t nslots = 0; std::vector<ImpD> imps /* tmpl field holds the producer SEGMENT until templates exist */; std::vector<uint32_t> inputs;
    std::map<std::pair<uint32_t, uint32_t>, uint32_t> imp_of;
    uint64_t cell0 = 0, rec0 = 0, ncells = 0, nrecs = 0; bool started = false;
    int tmpl = -1; uint32_t inst = 0;
};
static uint64_t rc_cells(int L, uint64_t bits) { if (bits == 0) return 0; const uint64_t n = (bits + L - 1) / L, rem = bits % L; return (n > 1 ? 1 + 3 * (n - 1) : 0) + (rem ? 4 : 0); }

}  // namespace h2w

using namespace h2w;

namespace h2w {
uint64_t traced_workspace_bytes(const h2w_plan *p, uint64_t n);
int traced_run(h2w_plan *p, const uint64_t *proofs_dev, uint64_t n_proofs, void *advice_dev, void *workspace_dev, void *stream_);
void traced_free(h2w_plan *p);
struct TracedWs { size_t recs, status, lflag, ctr, vals, total; };
static size_t al(size_t x) { return (x + 255) / 256 * 256; }
static TracedWs traced_ws(const h2w_plan *p, uint64_t n) {
    TracedWs w; size_t o = 0;
    w.recs = o; o += al((size_t)n * p->nrec * sizeof(rec_t));
    w.status = o; o += al((size_t)n * 4); w.lflag = o; o += al((size_t)n * 4); w.ctr = o; o += al((size_t)n * 4);
    w.vals = o; o += al((size_t)n * p->traced->total_slot_lanes * 8);
    w.total = o; return w;
}
uint64_t traced_workspace_bytes(const h2w_plan *p, uint64_t n) { return traced_ws(p, n).total; }
uint64_t traced_status_offset(const h2w_plan *p, uint64_t n, bool flags) { const TracedWs w = traced_ws(p, n); return flags ? w.lflag : w.status; }
void traced_free(h2w_plan *p) {
    TracedPlan *t = p->traced; if (!t) return;
    if (t->d_tape) (void)hipFree(t->d_tape); if (t->d_insts) (void)hipFree(t->d_insts); if (t->d_imps) (void)hipFree(t->d_imps);
    if (t->d_inputs) (void)hipFree(t->d_inputs); if (t->d_pool64) (void)hipFree(t->d_pool64); if (t->d_poolfr) (void)hipFree(t->d_poolfr);
    delete t; p->traced = nullptr;
}
int traced_run(h2w_plan *p, const uint64_t *proofs_dev, uint64_t n_proofs, void *advice_dev, void *workspace_dev, void *stream_) {
    TracedPlan *t = p->traced;
    if (n_proofs > 65535) { set_error("h2w_fri_witness_batch: more than 65535 proofs per call"); return -1; }
    DeviceGuard dg(p->device);
    hipStream_t stream = (hipStream_t)stream_;
    const TracedWs wl = traced_ws(p, n_proofs); char *ws = (char *)workspace_dev;
    ReplayArgs R; memset(&R, 0, sizeof(R));
    R.tape = t->d_tape; R.insts = t->d_insts; R.imps = t->d_imps; R.inputs = t->d_inputs; R.pool64 = t->d_pool64; R.poolfr = t->d_poolfr;
    R.proofs = proofs_dev; R.proof_words = p->pl.total; R.recs = (rec_t *)(ws + wl.recs); R.rec_stride = p->nrec; R.out = (fr_t *)advice_dev; R.cell_stride = p->ncells;
    R.vals = (uint64_t *)(ws + wl.vals); R.status = (uint32_t *)(ws + wl.status); R.ncells = p->d_ncells; R.inv_pos = p->d_inv; R.inv_neg = p->d_inv + INV_TAB; R.P = p->P; R.L = p->shape.lookup_bits;
    R.nproofs = (uint32_t)n_proofs;
    uint64_t base = 0;
    for (size_t i = 0; i < t->tmpls.size(); i++) { R.tbase[i] = base; R.tninst[i] = t->tmpls[i].ninst; base += (uint64_t)t->tmpls[i].nslots * n_proofs * t->tmpls[i].ninst; }
    H2W_HIP(hipMemsetAsync(ws + wl.status, 0, n_proofs * 4, stream));
    H2W_HIP(hipMemsetAsync(ws + wl.lflag, 0, n_proofs * 4, stream));
    uint32_t maxd = 0; for (const TmplD &T : t->tmpls) if (T.depth > maxd) maxd = T.depth;
    for (uint32_t d = 0; d <= maxd; d++)
        for (size_t i = 0; i < t->tmpls.size(); i++) {
            const TmplD &T = t->tmpls[i]; if (T.depth != d) continue;
            R.tape0 = T.tape0; R.nslots = T.nslots; R.ninst = T.ninst; R.inst0 = T.inst0; R.tmpl = (uint32_t)i;
            const uint64_t lanes = n_proofs * T.ninst;
            hipLaunchKernelGGL(k_replay, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, stream, R);
        }
    // expansion of the block records
    ExpandArgs E;
    E.meta = p->d_meta; E.recs = R.recs; E.nrec = p->nrec; E.rec_stride = p->nrec; E.out = R.out; E.cell_stride = p->ncells; E.pool = nullptr; E.cm = ColMap{nullptr, 0, 0};
    expand_unsharded(E); p->dt.fill(E);
    E.tile_ctr = (uint32_t *)(ws + wl.ctr); E.roam_per_cu = 2;
    H2W_HIP(hipMemsetAsync(E.tile_ctr, 0, n_proofs * 4, stream));
    int gx = (int)(2048 / (n_proofs < 2048 ? n_proofs : 2048)); if (gx < 8) gx = 8;
    if (launch_expand(E, n_proofs, gx, stream) != 0) return -1;
    H2W_HIP(hipGetLastError());
    return 0;
}
}  // namespace h2w

extern "C" h2w_plan *h2w_plan_from_trace(h2w_ctx *ctx, uint64_t proof_words, const char *const *parallel_scopes, size_t n_scopes, int device_id) {
    Trace *tr = ctx_trace(ctx);
    if (!tr) { set_error("h2w_plan_from_trace: the context is not in trace mode (h2w_ctx_trace_begin)"); return nullptr; }
    if (!tr->err.empty()) { set_error("h2w_plan_from_trace: " + tr->err); return nullptr; }
    if (tr->pending.kind) { set_error("h2w_plan_from_trace: a h2w_trace_input tag was never consumed"); return nullptr; }
    const int L = ctx_lookup_bits(ctx);
    std::string err;
    auto bad = [&](const std::string &m) { if (err.empty()) err = m; };
    TemplateTable tt(L);
    // ---- pass 1: segments
    std::vector<SegInfo> segs(1);
    std::vector<int> op_seg(tr->ops.size(), 0);
    {
        std::vector<int> stack;      // per open scope: the segment it opened, or -1 (an ordinary scope)
        int cur = 0;
        for (size_t i = 0; i < tr->ops.size(); i++) {
            const TraceOp &o = tr->ops[i];
            if (o.code == TR_SCOPE_PUSH) {
                bool par = false; for (size_t k = 0; k < n_scopes; k++) if (tr->names[(size_t)o.imm] == parallel_scopes[k]) par = true;
                if (par) { SegInfo s; s.parent = cur; s.depth = segs[(size_t)cur].depth + 1; s.name = (uint32_t)o.imm; segs.push_back(s); cur = (int)segs.size() - 1; stack.push_back(cur); }
                else stack.push_back(-1);
            } else if (o.code == TR_SCOPE_POP) {
                if (stack.empty()) { bad("unbalanced scopes in the trace"); break; }
                if (stack.back() >= 0) cur = segs[(size_t)stack.back()].parent;
                stack.pop_back();
            }
            op_seg[i] = cur;
        }
        if (!stack.empty()) bad("a scope is still open at the end of the trace");
    }
    // scope names compare by string: give every parallel name one id
    { std::map<std::string, uint32_t> ids; for (SegInfo &s : segs) { if (s.parent < 0) continue; auto it = ids.find(tr->names[s.name]); if (it == ids.end()) it = ids.emplace(tr->names[s.name], (uint32_t)ids.size() + 1).first; s.name = it->second; } }
    // ---- pass 2: lowering, in tape order
    std::unordered_map<uint64_t, uint32_t> val_of;      // cell offset of a handle -> value
    std::vector<ValInfo> vals;
    std::vector<uint64_t> pool64; std::map<uint64_t, uint32_t> pool64_of; std::vector<fr_t> poolfr;
    auto lit64 = [&](uint64_t v) { auto it = pool64_of.find(v); if (it != pool64_of.end()) return it->second; pool64.push_back(v); pool64_of[v] = (uint32_t)pool64.size() - 1; return (uint32_t)pool64.size() - 1; };
    auto litfr = [&](const fr_t &v) { for (size_t i = 0; i < poolfr.size(); i++) if (fr_eq(poolfr[i], v)) return (uint32_t)i; poolfr.push_back(v); return (uint32_t)poolfr.size() - 1; };
    std::vector<uint64_t> meta; uint64_t nrec = 0;
    auto is_ancestor = [&](int a, int s) { for (int x = s; x >= 0; x = segs[(size_t)x].parent) if (x == a) return true; return false; };
    // operand of op in segment s: the handle's cell, or a literal
    auto ref_of = [&](int s, const TraceIn &in, int *width_out) -> uint32_t {
        if (in.lit) { if (width_out) *width_out = W64; return mkref(RK_LIT64, W64, lit64(in.v)); }
        auto it = val_of.find(in.v);
        if (it == val_of.end()) { bad("an operand is not the result of a traced call (cell " + std::to_string(in.v) + ")"); if (width_out) *width_out = W64; return mkref(RK_LIT64, W64, lit64(0)); }
        const ValInfo &v = vals[it->second]; if (width_out) *width_out = v.width;
        if (v.is_static) return v.width == WFR ? mkref(RK_LITFR, WFR, v.lit) : mkref(RK_LIT64, W64, v.lit);
        if ((int)v.seg == s) return mkref(RK_LOCAL, v.width, v.slot);
        if (!is_ancestor((int)v.seg, s)) { bad("a parallel scope reads a value computed in a scope that does not enclose it: its instances are not independent (cell " + std::to_string(in.v) + ")"); return mkref(RK_LIT64, W64, lit64(0)); }
        SegInfo &S = segs[(size_t)s]; const auto key = std::make_pair(v.seg, v.slot);
        auto im = S.imp_of.find(key);
        if (im == S.imp_of.end()) { S.imps.push_back(ImpD{v.seg, 0, v.slot}); im = S.imp_of.emplace(key, (uint32_t)S.imps.size() - 1).first; }
        return mkref(RK_IMPORT, v.width, im->second);
    };
    auto new_val = [&](int s, uint64_t cell, int width, bool stat, uint32_t lit) -> uint32_t {
        SegInfo &S = segs[(size_t)s]; ValInfo v; v.seg = (uint32_t)s; v.width = (uint8_t)width; v.is_static = stat ? 1 : 0; v.lit = lit; v.slot = NO_SLOT;
        if (!stat) { v.slot = S.nslots; S.nslots += (uint32_t)SLOTS_OF[width]; }
        vals.push_back(v); val_of[cell] = (uint32_t)vals.size() - 1; return v.slot;
    };
    auto add_rec = [&](int s, int tmpl, uint64_t cell) { meta.push_back(meta_pack((uint32_t)tmpl, cell)); nrec++; segs[(size_t)s].nrecs++; };
    std::vector<int> open_child(segs.size(), -1);
    int prev_seg = 0;
    for (size_t i = 0; i < tr->ops.size() && err.empty(); i++) {
        const TraceOp &o = tr->ops[i]; const int s = op_seg[i]; SegInfo &S = segs[(size_t)s];
        if (o.code == TR_SCOPE_PUSH || o.code == TR_SCOPE_POP) {
            // entering a parallel child: the parent steps over its records and cells (filled in when the child ends)
            if (o.code == TR_SCOPE_PUSH && s != prev_seg && segs[(size_t)s].parent == prev_seg) {
                SegInfo &Pn = segs[(size_t)prev_seg]; Pn.tape.push_back(DOP_SKIP); open_child[(size_t)s] = (int)Pn.tape.size(); for (int k = 0; k < 4; k++) Pn.tape.push_back(0);
                S.cell0 = o.cell0; S.rec0 = nrec; S.started = true;
            }
            if (o.code == TR_SCOPE_POP && s != prev_seg && segs[(size_t)prev_seg].parent == s) {
                SegInfo &C = segs[(size_t)prev_seg]; C.ncells = o.cell0 - C.cell0; C.tape.push_back(DOP_END);
                // the child's totals include its own children's (they are nested in its cell and record ranges)
                const uint64_t nr = nrec - C.rec0; uint32_t *w = S.tape.data() + open_child[(size_t)prev_seg];
                w[0] = (uint32_t)nr; w[1] = (uint32_t)(nr >> 32); w[2] = (uint32_t)C.ncells; w[3] = (uint32_t)(C.ncells >> 32);
            }
            prev_seg = s; continue;
        }
        prev_seg = s;
        const TraceIn *in = tr->ins.data() + o.first_in; const uint64_t *out = tr->outs.data() + o.first_out;
        std::vector<uint32_t> &T = S.tape; uint64_t want_cells = 0; int w0 = 0, w1 = 0, w2 = 0;
        auto head = [&](uint32_t op, uint32_t n = 0, uint32_t aux = 0) { T.push_back(op | (n << 8) | (aux << 16)); };
        switch (o.code) {
            case TR_LOAD_CONSTANT: {
                const fr_t c = tr->consts[(size_t)o.imm]; const bool small = (c.l[1] | c.l[2] | c.l[3]) == 0;
                if (o.tag.kind == 1) {
                    if (o.tag.n != 1) { bad("a constant that is a proof value must be one Goldilocks word"); break; }
                    S.inputs.push_back((uint32_t)o.tag.word); head(DOP_CONST1); T.push_back(mkref(RK_INPUT, W64, (uint32_t)S.inputs.size() - 1)); T.push_back(new_val(s, out[0], W64, false, 0));
                    add_rec(s, T_CONST1, o.cell0); want_cells = 1;
                } else if (o.tag.kind != 0) bad("a hint tag on a constant");
                else if (small) { const uint32_t li = lit64(c.l[0]); head(DOP_CONST1); T.push_back(mkref(RK_LIT64, W64, li)); T.push_back(NO_SLOT); new_val(s, out[0], W64, true, li); add_rec(s, T_CONST1, o.cell0); want_cells = 1; }
                else { const uint32_t li = litfr(c); head(DOP_FRCELL); T.push_back(mkref(RK_LITFR, WFR, li)); T.push_back(NO_SLOT); new_val(s, out[0], WFR, true, li); want_cells = 1; }
                break;
            }
            case TR_LOAD_WITNESS: {
                if (o.tag.kind != 1) { bad("h2w_load_witness without h2w_trace_input"); break; }
                S.inputs.push_back((uint32_t)o.tag.word);
                if (o.tag.n == 4) { head(DOP_FRCELL); T.push_back(mkref(RK_INPUT, WFR, (uint32_t)S.inputs.size() - 1)); T.push_back(new_val(s, out[0], WFR, false, 0)); }
                else { head(DOP_CONST1); T.push_back(mkref(RK_INPUT, W64, (uint32_t)S.inputs.size() - 1)); T.push_back(new_val(s, out[0], W64, false, 0)); add_rec(s, T_CONST1, o.cell0); }
                want_cells = 1; break;
            }
            case TR_ADD: case TR_MUL: case TR_MUL_ADD: {
                const uint32_t a = ref_of(s, in[0], &w0), b = ref_of(s, in[1], &w1), c3 = o.code == TR_MUL_ADD ? ref_of(s, in[2], &w2) : 0;
                const bool narrow = w0 == W64 && w1 == W64 && (o.code != TR_MUL_ADD || w2 == W64);
                if (narrow) {      // [C, A, B, A B + C] on values below 2^64: one gate record (GoldilocksChip::*_no_reduce, base.rs:240-294)
                    head(DOP_GATE, 0, T_GATE);
                    if (o.code == TR_ADD) { T.push_back(b); T.push_back(mkref(RK_LIT64, W64, lit64(1))); T.push_back(a); }
                    else if (o.code == TR_MUL) { T.push_back(a); T.push_back(b); T.push_back(mkref(RK_LIT64, W64, lit64(0))); }
                    else { T.push_back(a); T.push_back(b); T.push_back(c3); }
                    T.push_back(new_val(s, out[0], W128, false, 0)); add_rec(s, T_GATE, o.cell0);
                } else {
                    head(o.code == TR_ADD ? DOP_FR_ADD : o.code == TR_MUL ? DOP_FR_MUL : DOP_FR_MULADD); T.push_back(a); T.push_back(b); if (o.code == TR_MUL_ADD) T.push_back(c3);
                    T.push_back(new_val(s, out[0], WFR, false, 0));
                }
                want_cells = 4; break;
            }
            case TR_SELECT: {
                const uint32_t a = ref_of(s, in[0], &w0), b = ref_of(s, in[1], &w1), sl = ref_of(s, in[2], &w2);
                if (w2 != W64) { bad("select: the selector is not a bit"); break; }
                const bool narrow = w0 == W64 && w1 == W64;
                head(narrow ? DOP_SELECT : DOP_FR_SELECT); T.push_back(a); T.push_back(b); T.push_back(sl); T.push_back(new_val(s, out[0], narrow ? W64 : WFR, false, 0));
                want_cells = 8; break;
            }
            case TR_IDX_TO_INDICATOR: {
                const uint32_t n = (uint32_t)o.imm; const uint32_t a = ref_of(s, in[0], &w0);
                if (w0 != W64 || n < 1 || n > 64) { bad("idx_to_indicator: a wide index or more than 64 entries"); break; }
                head(DOP_IDX2IND, n); T.push_back(a); const uint32_t base = S.nslots;
                for (uint32_t k = 0; k < n; k++) { const uint32_t sl = new_val(s, out[k], W64, false, 0); if (sl != base + k) bad("internal: slots of an array result"); }
                T.push_back(base); want_cells = 8 + 12ull * (n - 1); break;
            }
            case TR_SELECT_BY_INDICATOR: {
                const uint32_t n = (uint32_t)o.imm; if (n < 1 || n > 64) { bad("select_by_indicator: more than 64 entries"); break; }
                std::vector<uint32_t> r(2 * n); bool narrow = true;
                for (uint32_t k = 0; k < 2 * n; k++) { int w; r[k] = ref_of(s, in[k], &w); if (k < n && w != W64) narrow = false; if (k >= n && w != W64) bad("select_by_indicator: an indicator that is not a bit"); }
                head(narrow ? DOP_SELIND : DOP_FR_SELIND, n); for (uint32_t x : r) T.push_back(x); T.push_back(new_val(s, out[0], narrow ? W64 : WFR, false, 0));
                want_cells = 1 + 3ull * n; break;
            }
            case TR_NUM_TO_BITS: {
                const uint32_t n = (uint32_t)o.imm; const uint32_t a = ref_of(s, in[0], &w0);
                if (w0 != W64 || n < 1 || n > 64) { bad("num_to_bits: a wide value or more than 64 bits"); break; }
                head(DOP_NUM2BITS, n); T.push_back(a); const uint32_t base = S.nslots;
                for (uint32_t k = 0; k < n; k++) new_val(s, out[k], W64, false, 0);
                T.push_back(base); want_cells = (1 + 3ull * (n - 1)) + 4ull * n; break;
            }
            case TR_BITS_TO_NUM: {
                const uint32_t n = (uint32_t)o.imm; if (n > 64) { bad("bits_to_num: more than 64 bits"); break; }
                head(DOP_BITS2NUM, n); for (uint32_t k = 0; k < n; k++) { int w; T.push_back(ref_of(s, in[k], &w)); if (w != W64) bad("bits_to_num: an operand that is not a bit"); }
                T.push_back(new_val(s, out[0], W64, false, 0)); want_cells = n ? 1 + 3ull * (n - 1) : 1; break;
            }
            case TR_DECOMPOSE_LE: {
                if ((o.imm >> 32) != 56 || (uint32_t)o.imm != 5) { bad("decompose_le: only (56 bits, 5 limbs) is replayable (HashWire::to_goldilocks_vec, hash/poseidon_bn254/hash.rs:31-43)"); break; }
                head(DOP_DECOMP565); T.push_back(ref_of(s, in[0], &w0)); const uint32_t base = S.nslots; for (int k = 0; k < 5; k++) new_val(s, out[k], W64, false, 0); T.push_back(base);
                want_cells = 13 + 5 * rc_cells(L, 56); break;
            }
            case TR_LIMBS_TO_NUM: {
                const uint32_t n = o.n_in; if (o.imm != 64 || n < 1 || n > 4) { bad("limbs_to_num: only up to four 64-bit limbs are replayable"); break; }
                head(DOP_LIMBS2NUM, n); for (uint32_t k = 0; k < n; k++) { int w; T.push_back(ref_of(s, in[k], &w)); if (w != W64) bad("limbs_to_num: a wide limb"); }
                T.push_back(new_val(s, out[0], WFR, false, 0)); want_cells = 1 + 3ull * (n - 1); break;
            }
            case TR_RANGE_CHECK: {
                const uint32_t a = ref_of(s, in[0], &w0); if (w0 != W64 || o.imm > 64) { bad("range_check: a wide value"); break; }
                head(DOP_RANGE, 0, (uint32_t)o.imm); T.push_back(a); want_cells = rc_cells(L, o.imm); break;
            }
            case TR_CLT_SAFE: {
                const uint32_t a = ref_of(s, in[0], &w0); if (w0 != W64 || o.imm != GL_P) { bad("check_less_than_safe: only (64-bit value, Goldilocks order) is replayable"); break; }
                head(DOP_CLT); T.push_back(a); add_rec(s, T_CLT_SAFE, o.cell0); want_cells = (uint64_t)tt.ncells(T_CLT_SAFE); break;
            }
            case TR_GL_WITNESS: {
                if (o.tag.kind == 1) { if (o.tag.n != 1) { bad("a Goldilocks witness of more than one word"); break; } S.inputs.push_back((uint32_t)o.tag.word); head(DOP_LOADW); T.push_back(mkref(RK_INPUT, W64, (uint32_t)S.inputs.size() - 1)); }
                else if (o.tag.kind == 2) { head(DOP_LOADW_DIV); T.push_back(ref_of(s, TraceIn{o.tag.a, 0}, &w0)); T.push_back(ref_of(s, TraceIn{o.tag.b, 0}, &w1)); if (w0 != W64 || w1 != W64) bad("div: wide operands"); }
                else if (o.tag.kind == 3 || o.tag.kind == 4) { head(DOP_LOADW_EXTINV, 0, (uint32_t)(o.tag.kind - 3)); T.push_back(ref_of(s, TraceIn{o.tag.a, 0}, &w0)); T.push_back(ref_of(s, TraceIn{o.tag.b, 0}, &w1)); if (w0 != W64 || w1 != W64) bad("ext inverse: wide operands"); }
                else { bad("h2w_gl_load_witness without h2w_trace_input"); break; }
                T.push_back(new_val(s, out[0], W64, false, 0)); add_rec(s, T_LOADW, o.cell0); want_cells = (uint64_t)tt.ncells(T_LOADW); break;
            }
            case TR_GL_REDUCE: {
                const uint32_t a = ref_of(s, in[0], &w0); if (w0 == W64) { bad("gl_reduce of a value that is not a gate output"); break; }
                if (ref_kind(a) != RK_LOCAL && ref_kind(a) != RK_IMPORT) { bad("gl_reduce of a constant"); break; }
                head(DOP_REDUCE); T.push_back(a); T.push_back(new_val(s, out[0], W64, false, 0)); add_rec(s, T_REDUCE, o.cell0); want_cells = (uint64_t)tt.ncells(T_REDUCE); break;
            }
            case TR_GLOP: {
                const uint32_t a = ref_of(s, in[0], &w0), b = ref_of(s, in[1], &w1), c3 = ref_of(s, in[2], &w2);
                if (w0 != W64 || w1 != W64 || w2 != W64) { bad("a Goldilocks op on a wide value"); break; }
                head(DOP_GLOP, 0, o.sub); T.push_back(a); T.push_back(b); T.push_back(c3); T.push_back(new_val(s, out[0], W64, false, 0)); add_rec(s, o.sub, o.cell0); want_cells = (uint64_t)tt.ncells(o.sub); break;
            }
            default: bad("unknown op in the trace");
        }
        if (err.empty() && want_cells != o.ncells) bad("internal: op " + std::to_string(o.code) + " appended " + std::to_string(o.ncells) + " cells on the host, the device template has " + std::to_string(want_cells));
    }
    if (!err.empty()) { set_error("h2w_plan_from_trace: " + err); return nullptr; }
    segs[0].tape.push_back(DOP_END); segs[0].cell0 = 0; segs[0].rec0 = 0;
    // ---- templates: isomorphic instances (equal tapes, slot counts, table sizes) share one
    TracedPlan *tp = new TracedPlan();
    std::vector<uint32_t> tape_all; std::vector<InstD> insts; std::vector<ImpD> imps; std::vector<uint32_t> inputs;
    std::vector<std::vector<int>> members;
    {
        uint32_t maxd = 0; for (const SegInfo &S : segs) if ((uint32_t)S.depth > maxd) maxd = (uint32_t)S.depth;
        for (uint32_t d = 0; d <= maxd; d++)
            for (size_t si = 0; si < segs.size(); si++) {
                SegInfo &S = segs[si]; if ((uint32_t)S.depth != d) continue;
                int found = -1;
                for (size_t t = 0; t < members.size() && found < 0; t++) {
                    const SegInfo &M = segs[(size_t)members[t][0]];
                    if (M.depth == S.depth && M.name == S.name && M.nslots == S.nslots && M.imps.size() == S.imps.size() && M.inputs.size() == S.inputs.size() && M.tape == S.tape) found = (int)t;
                }
                if (found < 0) { members.push_back({}); found = (int)members.size() - 1; }
                S.tmpl = found; S.inst = (uint32_t)members[(size_t)found].size(); members[(size_t)found].push_back((int)si);
            }
        if (members.size() > (size_t)MAX_TMPL) { set_error("h2w_plan_from_trace: more than " + std::to_string(MAX_TMPL) + " distinct scope shapes"); delete tp; return nullptr; }
        for (size_t t = 0; t < members.size(); t++) {
            const SegInfo &M = segs[(size_t)members[t][0]];
            TmplD T; T.tape0 = (uint32_t)tape_all.size(); T.nslots = M.nslots ? M.nslots : 1; T.ninst = (uint32_t)members[t].size(); T.inst0 = (uint32_t)insts.size(); T.depth = (uint32_t)M.depth;
            tape_all.insert(tape_all.end(), M.tape.begin(), M.tape.end());
            for (int si : members[t]) {
                const SegInfo &S = segs[(size_t)si];
                InstD I; I.cell0 = S.cell0; I.rec0 = S.rec0; I.imp0 = (uint32_t)imps.size(); I.in0 = (uint32_t)inputs.size();
                for (const ImpD &m : S.imps) { const SegInfo &Pn = segs[(size_t)m.tmpl]; imps.push_back(ImpD{(uint32_t)Pn.tmpl, Pn.inst, m.slot}); }
                inputs.insert(inputs.end(), S.inputs.begin(), S.inputs.end());
                insts.push_back(I);
            }
            tp->tmpls.push_back(T); tp->total_slot_lanes += (uint64_t)T.nslots * T.ninst;
        }
    }
    for (uint32_t w : inputs) if (w >= proof_words) { set_error("h2w_plan_from_trace: an input tag beyond proof_words"); delete tp; return nullptr; }
    tp->n_ops = tr->ops.size(); tp->n_segments = segs.size();
    // ---- the plan handle
    h2w_plan *pl = new h2w_plan(L);
    memset(&pl->shape, 0, sizeof(pl->shape)); pl->shape.lookup_bits = L; pl->shape.num_queries = 1; pl->shape.hash_mode = 1;
    pl->device = device_id; pl->P = fr_params_init(); memset(&pl->st, 0, sizeof(pl->st)); memset(&pl->pl, 0, sizeof(pl->pl)); memset(&pl->d, 0, sizeof(pl->d));
    pl->pl.total = proof_words; pl->nrec = nrec; pl->ncells = ctx_num_cells(ctx); pl->traced = tp;
    for (uint64_t m : meta) pl->rec_cells += (uint64_t)pl->tt.ncells((int)meta_tmpl(m));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { pl->device = -1; return pl; }      // layout queries only
    if (device_id < 0 || device_id >= ndev) { set_error("h2w_plan_from_trace: device_id out of range"); h2w_plan_free(pl); return nullptr; }
    DeviceGuard dg(device_id);
    auto up = [&]() -> int {
        if (pl->dt.upload(pl->tt) != 0) return -1;
        auto put = [&](void **d, const void *h, size_t bytes) -> int { H2W_HIP(hipMalloc(d, bytes ? bytes : 8)); if (bytes) H2W_HIP(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice)); return 0; };
        if (put((void **)&pl->d_meta, meta.data(), meta.size() * 8) != 0) return -1;
        if (put((void **)&tp->d_tape, tape_all.data(), tape_all.size() * 4) != 0) return -1;
        if (put((void **)&tp->d_insts, insts.data(), insts.size() * sizeof(InstD)) != 0) return -1;
        if (put((void **)&tp->d_imps, imps.data(), imps.size() * sizeof(ImpD)) != 0) return -1;
        if (put((void **)&tp->d_inputs, inputs.data(), inputs.size() * 4) != 0) return -1;
        if (put((void **)&tp->d_pool64, pool64.data(), pool64.size() * 8) != 0) return -1;
        if (put((void **)&tp->d_poolfr, poolfr.data(), poolfr.size() * sizeof(fr_t)) != 0) return -1;
        std::vector<uint16_t> nc(T_MAX, 0); for (size_t i = 0; i < pl->tt.info.size(); i++) nc[i] = pl->tt.info[i].ncells;
        if (put((void **)&pl->d_ncells, nc.data(), nc.size() * 2) != 0) return -1;
        std::vector<fr_t> inv(2 * INV_TAB, fr_zero());
        for (int k2 = 1; k2 < INV_TAB; k2++) { inv[k2] = fr_inv(fr_from_u64((uint64_t)k2), pl->P); inv[INV_TAB + k2] = fr_neg(inv[k2]); }
        if (put((void **)&pl->d_inv, inv.data(), inv.size() * sizeof(fr_t)) != 0) return -1;
        return 0;
    };
    if (up() != 0) { h2w_plan_free(pl); return nullptr; }
    return pl;
}
