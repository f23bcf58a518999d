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
//   the ring   A load waits for every store issued before it (one in-order counter on this family), and every op stores a record: with the values in
//              global memory alone an op cost 2-7 us.  The last RING_K value slots of a lane live in LDS as well (a different counter): the lowering
//              knows the distance from every operand to its producer (99.8 % of the prologue's operands, 93 % of the glue's, all of a PoseidonBN254
//              path's are within 256 slots) and addresses those through the ring; the global store stays (write-through: far operands, imports).
// Which op becomes which template is decided on STATIC widths (a value is provably below 2^64 when a Goldilocks-level op, a bit decomposition, a
// one-word input ... produced it), never on the traced values: the layout of the records must not depend on the proof.
#define H2W_FLATTEN_CHIPS 1      // the interpreter's ops are the value backend's, inlined: out of line every op is a function call, and a call waits for the stores in flight (glue.hip)
#include <hip/hip_runtime.h>
#include <unordered_map>
#include <map>
#include <vector>
#include <string>
#include <cstring>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include "plan.h"
#include "trace.h"

struct h2w_ctx;
namespace h2w {
Trace *ctx_trace(h2w_ctx *); int ctx_lookup_bits(const h2w_ctx *); uint64_t ctx_num_cells(const h2w_ctx *);      // eager.cpp

enum { W64 = 0, W128 = 1, WFR = 2 };
enum { RK_LOCAL = 0, RK_IMPORT = 1, RK_LIT64 = 2, RK_INPUT = 3, RK_LITFR = 4, RK_RING = 5 };
constexpr uint32_t RING_K = 256;      // value slots of a lane kept in LDS (64 lanes x 256 x 8 B = 128 KB per block)
// LDS of a block: the ring, then the constant pools (as far as they fit: the rest is fetched like any far operand)
constexpr uint32_t LDS_RING_BYTES = RING_K * 64 * 8, POOL64_CAP = 2048, POOLFR_CAP = 472;
constexpr uint32_t LDS_POOL64 = LDS_RING_BYTES, LDS_POOLFR = LDS_POOL64 + POOL64_CAP * 8, LDS_WORDS = (LDS_POOLFR + POOLFR_CAP * 32) / 8;
// An operand word of a compute op ("fast ref"): bit 31 set: the ring, bits 0..7 = slot mod 256 (word j of the value at slot + j); bit 31 clear: a pool
// entry in LDS, bits 0..17 = its byte offset (word j at + 8 j); bits 27..28: the width as before.  Everything else (a far slot, an enclosing segment's
// value, a proof word, a pool entry beyond the LDS part) is brought into the ring by a DOP_FETCH in front of the op, which carries the old-style ref.
HD uint32_t fastref_ring(int width, uint32_t slot) { return 0x80000000u | ((uint32_t)width << 27) | (slot & (RING_K - 1)); }
HD uint32_t fastref_pool(int width, uint32_t byte_off) { return ((uint32_t)width << 27) | byte_off; }
HD uint32_t mkref(int kind, int width, uint32_t idx) { return ((uint32_t)kind << 29) | ((uint32_t)width << 27) | idx; }
HD int ref_kind(uint32_t r) { return (int)(r >> 29); }
HD int ref_width(uint32_t r) { return (int)((r >> 27) & 3); }
HD uint32_t ref_idx(uint32_t r) { return r & ((1u << 27) - 1); }
constexpr int SLOTS_OF[3] = {1, 2, 4};
// device ops; word 0 = op | n << 8 | aux << 16 | (words of the op) << 24, then operand refs, then (ops with results) the first output slot
enum { DOP_END = 0, DOP_SKIP, DOP_CONST1, DOP_FRCELL, DOP_LOADW, DOP_LOADW_DIV, DOP_LOADW_EXTINV, DOP_GLOP, DOP_GATE, DOP_REDUCE, DOP_CLT,
       DOP_FR_ADD, DOP_FR_MUL, DOP_FR_MULADD, DOP_SELECT, DOP_FR_SELECT, DOP_IDX2IND, DOP_SELIND, DOP_FR_SELIND, DOP_NUM2BITS, DOP_BITS2NUM,
       DOP_DECOMP565, DOP_LIMBS2NUM, DOP_RANGE, DOP_FETCH,
       DOP_GLOPRUN };      // n consecutive DOP_GLOP ops as one: [hdr][cells of the run][A, B, C, out slot | template << 24] x n (2 + 4 n words: its length is NOT in the header)
// operand words of an op: [first, first + count)
HD void operand_span(uint32_t op, uint32_t n, uint32_t &first, uint32_t &count) {
    first = 1; count = 0;
    switch (op) {
        case DOP_CONST1: case DOP_FRCELL: case DOP_LOADW: case DOP_REDUCE: case DOP_CLT: case DOP_RANGE: case DOP_IDX2IND: case DOP_NUM2BITS: case DOP_DECOMP565: count = 1; break;
        case DOP_LOADW_DIV: case DOP_LOADW_EXTINV: case DOP_FR_ADD: case DOP_FR_MUL: count = 2; break;
        case DOP_GLOP: case DOP_GATE: case DOP_FR_MULADD: case DOP_SELECT: case DOP_FR_SELECT: count = 3; break;
        case DOP_SELIND: case DOP_FR_SELIND: count = 2 * n; break;
        case DOP_BITS2NUM: case DOP_LIMBS2NUM: count = n; break;
        default: break;
    }
}
constexpr int MAX_TMPL = 48;
constexpr uint32_t NO_SLOT = 0xffffffffu;

struct InstD { uint64_t cell0, rec0; uint32_t imp0, in0; };
struct ImpD { uint32_t tmpl, inst, slot; };
struct TmplD { uint32_t tape0, nslots, ninst, inst0, depth; };

struct ReplayArgs {
    const uint32_t *tape; const InstD *insts; const ImpD *imps; const uint32_t *inputs; const uint64_t *pool64; const fr_t *poolfr;
    const uint64_t *proofs; uint64_t proof_words; rec_t *recs; uint64_t rec_stride; fr_t *out; uint64_t cell_stride; ColMap cm; uint64_t *vals; uint32_t *status;
    const uint16_t *ncells; const fr_t *inv_pos, *inv_neg; FrParams P; int L; uint32_t nproofs;
    uint32_t depth, ntmpl, npool64, npoolfr;
    const TmplD *tm; const uint64_t *prefix;      // per template (device tables of the plan): its description; the u64 elements per proof of the value stores before it
    uint32_t blk0[MAX_TMPL + 1];                  // first block of each template among the blocks of this launch (templates of `depth` only)
};

struct TracedPlan {
    std::vector<TmplD> tmpls; uint64_t total_slot_lanes = 0;      // sum over templates of nslots * ninst: u64 elements of the value store per proof
    TmplD *d_tm = nullptr; uint64_t *d_prefix = nullptr; uint32_t npool64 = 0, npoolfr = 0;
    uint32_t *d_tape = nullptr; InstD *d_insts = nullptr; ImpD *d_imps = nullptr; uint32_t *d_inputs = nullptr; uint64_t *d_pool64 = nullptr; fr_t *d_poolfr = nullptr;
    uint64_t n_ops = 0, n_segments = 0;
};

// ------------------------------------------------------------------------------------------------------------------- device
// (its own sink type: the flattened value backend of this unit is instantiated nowhere else - field.h HNI)
struct ReplaySink {
    static constexpr bool kCoop = false, kSplitOnly = false, kBnUnits = false, kDevSponge = false; static constexpr int kHashMode = -1;
    HF void coop_poseidon_permute(uint64_t *, const h2w_poseidon_consts_t *) {}
    rec_t *recs; uint64_t nrec; fr_t *out; uint64_t cell_off; const uint32_t *ncells;      // ncells: an LDS table (a global load per record would wait for the record stores in flight)
    ColCursor cc;      // the FlexGate column layout of a direct cell (flat stream: the identity, never located)
    HF void rec(int t, uint64_t a, uint64_t b, uint64_t c, uint64_t d) { g_store_rec(recs + nrec, a, b, c, d); nrec++; cell_off += ncells[t]; }
    HF void cell(const fr_t &v) { g_store_fr(out + cc.map(cell_off), v); cell_off++; }
    HF void gate() {} HF void lookup() {}
    HF void skip(uint64_t nr, uint64_t nc) { nrec += nr; cell_off += nc; }
    HF void merkle_begin(int, int, bool, uint64_t) {} HF void merkle_end(int, int, bool) {} HF void query_begin(int, uint64_t) {} HF void query_end(int, uint64_t) {}
    HF void bn_perm_begin(bool) {} HF void bn_perm_end(bool) {} HF void glp_note() {} HF void note_load(uint64_t, int) {} HF void note_cap_hash(uint64_t) {}
    HF bool coop_load_proof(const ValCfg &) { return false; } HF bool bn_emit_inline(fr_t *, const ValCfg &, bool &) { return false; }
};
typedef ValBackend<ReplaySink> RB;
__shared__ uint64_t s_lds[LDS_WORDS];      // [ring | pool64 | poolfr]
__shared__ uint32_t s_nc[T_MAX];
// what the rare operand kinds need (uniform over the block), in LDS: the out-of-line path below must not take the address of the kernel arguments
struct SlowCtx { uint64_t *gvals; const TmplD *tm; const uint64_t *prefix; const ImpD *imps; const uint32_t *inputs; const uint64_t *pool64; const fr_t *poolfr; uint32_t nproofs; };
__shared__ SlowCtx s_cx;
__device__ __forceinline__ uint32_t tw(const uint32_t *tape, uint32_t i) { return *(const __attribute__((address_space(4))) uint32_t *)(tape + i); }
// an operand word that is not in LDS: a far slot of the own store, a value of an enclosing segment, a proof word, a far pool entry (DOP_FETCH only)
__device__ __noinline__ uint64_t slow_get64(uint32_t r, uint32_t j, const uint64_t *vals_g, uint64_t Lt, uint32_t p, uint32_t imp0, uint32_t in0, const uint64_t *proof) {
    const uint32_t i = ref_idx(r); const int k = ref_kind(r);
    if (k == RK_LOCAL || k == RK_RING) return H2W_GLOAD64(vals_g + (uint64_t)(i + j) * Lt);
    if (k == RK_INPUT) return g_load_u64(proof + s_cx.inputs[in0 + i] + j);
    if (k == RK_LIT64) return g_load_u64(s_cx.pool64 + i + j);
    if (k == RK_LITFR) return g_load_u64(reinterpret_cast<const uint64_t *>(s_cx.poolfr + i) + j);
    const ImpD d = s_cx.imps[imp0 + i]; const uint32_t ni = s_cx.tm[d.tmpl].ninst;
    return H2W_GLOAD64(s_cx.gvals + (uint64_t)s_cx.nproofs * s_cx.prefix[d.tmpl] + (uint64_t)(d.slot + j) * ((uint64_t)s_cx.nproofs * ni) + (uint64_t)p * ni + d.inst);
}

__global__ __launch_bounds__(64) __attribute__((flatten)) void k_replay(ReplayArgs R) {
    const uint32_t lane = threadIdx.x;
    if (lane < T_MAX) s_nc[lane] = R.ncells[lane];
    if (lane == 0) { s_cx.gvals = R.vals; s_cx.tm = R.tm; s_cx.prefix = R.prefix; s_cx.imps = R.imps; s_cx.inputs = R.inputs; s_cx.pool64 = R.pool64; s_cx.poolfr = R.poolfr; s_cx.nproofs = R.nproofs; }
    for (uint32_t i = lane; i < R.npool64 && i < POOL64_CAP; i += 64) s_lds[LDS_POOL64 / 8 + i] = g_load_u64(R.pool64 + i);
    for (uint32_t i = lane; i < 4 * R.npoolfr && i < 4 * POOLFR_CAP; i += 64) s_lds[LDS_POOLFR / 8 + i] = g_load_u64(reinterpret_cast<const uint64_t *>(R.poolfr) + i);
    __syncthreads();
    // which template this block runs (the templates of one depth share a launch: they do not depend on one another)
    uint32_t t = 0;
#pragma unroll 1
    for (uint32_t i = 0; i < R.ntmpl; i++) if (R.tm[i].depth == R.depth && blockIdx.x >= R.blk0[i]) t = i;
    // (a value loaded from global memory is "divergent" to the compiler even at a uniform address: without the readfirstlane the tape pointer is a
    //  vector register, every tape word a VECTOR load - behind the record stores in flight, 2.5 us per op - and every branch of the interpreter a lane mask)
    t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    const TmplD T = R.tm[t];
    const uint32_t ninst = (uint32_t)__builtin_amdgcn_readfirstlane((int)T.ninst), tape0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)T.tape0), inst0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)T.inst0);
    const uint32_t g = (blockIdx.x - R.blk0[t]) * 64 + lane;
    if (g >= R.nproofs * ninst) return;
    const uint32_t p = g / ninst, inst = g % ninst;
    const InstD *const I = R.insts + inst0 + inst;
    const uint32_t imp0 = I->imp0, in0 = I->in0;
    ReplaySink sink; sink.recs = R.recs + (uint64_t)p * R.rec_stride; sink.out = R.out + (uint64_t)p * R.cell_stride; sink.ncells = s_nc; sink.cc.init(R.cm);
    sink.nrec = I->rec0; sink.cell_off = I->cell0;
    ValCfg cfg; cfg.proof = R.proofs + (uint64_t)p * R.proof_words; cfg.mode = 1; cfg.L = R.L; cfg.P = R.P; cfg.inv_pos = R.inv_pos; cfg.inv_neg = R.inv_neg; cfg.st = nullptr;
    cfg.split = false; cfg.split_bn = false; cfg.load_items = nullptr; cfg.n_load_items = 0; cfg.load_nrec = cfg.load_ncell = 0; cfg.n_cap_items = 0; cfg.fri = nullptr;
    RB be(sink, cfg, true);
    const uint64_t Lt = (uint64_t)R.nproofs * ninst;
    const uint64_t pre = R.prefix[t];
    uint64_t *const vals_g = R.vals + (uint64_t)R.nproofs * (((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pre >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pre)) + g;
    const uint64_t *const proof = cfg.proof; const uint32_t *const tape = R.tape;
    const uint32_t lane8 = lane * 8;
    const char *const lds = reinterpret_cast<const char *>(s_lds);
    // word j of an operand: the ring or a pool entry, both in LDS, no branch (the ref says which: fastref_*)
    auto get64 = [&](uint32_t r, uint32_t j) -> uint64_t {
        const bool ring = (r >> 31) != 0;
        const uint32_t off = ring ? (((r + j) & (RING_K - 1)) << 9) : ((r & 0x3ffffu) + (j << 3));
        return *reinterpret_cast<const uint64_t *>(lds + off + (ring ? lane8 : 0u));
    };
    auto getfr = [&](uint32_t r) -> fr_t {
        const uint32_t w = (r >> 27) & 3u;      // W64, W128, WFR
        const uint64_t a = get64(r, 0), b = w != W64 ? get64(r, 1) : 0, c = w == WFR ? get64(r, 2) : 0, d = w == WFR ? get64(r, 3) : 0;
        fr_t v; v.l[0] = a; v.l[1] = b; v.l[2] = c; v.l[3] = d; return v;
    };
    auto ring_put = [&](uint32_t slot, uint64_t v) { s_lds[(slot & (RING_K - 1)) * 64 + lane] = v; };
    auto st1 = [&](uint32_t slot, uint64_t v) { ring_put(slot, v); H2W_GSTORE64(vals_g + (uint64_t)slot * Lt, v); };
    auto put64 = [&](uint32_t slot, uint64_t v) { if (slot != NO_SLOT) st1(slot, v); };
    auto putfr = [&](uint32_t slot, const fr_t &v) { if (slot != NO_SLOT) { st1(slot, v.l[0]); st1(slot + 1, v.l[1]); st1(slot + 2, v.l[2]); st1(slot + 3, v.l[3]); } };
    uint64_t ta[64], tb[64];
    // The op at pc sits in eight scalar registers (a window of the tape); the window of the NEXT op is requested before this one runs: a scalar load
    // waited for where it is used was ~200 cycles, five times per op.  (Ops with operand lists read the words beyond the window themselves.)
    uint32_t pc = tape0;
    uint32_t w0 = tw(tape, pc), w1 = tw(tape, pc + 1u), w2 = tw(tape, pc + 2u), w3 = tw(tape, pc + 3u), w4 = tw(tape, pc + 4u);
#pragma unroll 1
    for (;;) {
        const uint32_t h = w0; const uint32_t op = h & 0xff, n = (h >> 8) & 0xff, aux = (h >> 16) & 0xff;
        if (op == DOP_END) break;
        const uint32_t pcn = pc + (op == DOP_GLOPRUN ? 2u + 4u * n : (h >> 24));
        const uint32_t n0 = tw(tape, pcn), n1 = tw(tape, pcn + 1), n2 = tw(tape, pcn + 2), n3 = tw(tape, pcn + 3), n4 = tw(tape, pcn + 4);
        switch (op) {
            case DOP_SKIP: { const uint64_t nr = ((uint64_t)w2 << 32) | w1, nc = ((uint64_t)w4 << 32) | w3; sink.skip(nr, nc); break; }
            case DOP_CONST1: { const uint64_t v = get64(w1, 0); sink.rec(T_CONST1, v, 0, 0, 0); put64(w2, v); break; }
            case DOP_FRCELL: { const fr_t v = getfr(w1); be.cell(v); putfr(w2, v); break; }
            case DOP_LOADW: { const uint64_t v = get64(w1, 0); sink.rec(T_LOADW, v, 0, 0, 0); put64(w2, v); break; }
            case DOP_LOADW_DIV: {      // the hint of GoldilocksChip::div (base.rs:371-393): a / b; b == 0: status 1, the cells of the op on 1
                const uint64_t a = get64(w1, 0); uint64_t b = get64(w2, 0);
                if (b == 0) { be.fail(1); b = 1; }
                const uint64_t v = gl_mul(a, gl_inv(b)); sink.rec(T_LOADW, v, 0, 0, 0); put64(w3, v); break;
            }
            case DOP_LOADW_EXTINV: {   // extension.rs:320-340
                gle_t a; a.c[0] = get64(w1, 0); a.c[1] = get64(w2, 0);
                if (a.c[0] == 0 && a.c[1] == 0) { be.fail(2); a.c[0] = 1; }
                const gle_t iv = gle_inv(a); const uint64_t v = (aux & 1) ? iv.c[1] : iv.c[0]; sink.rec(T_LOADW, v, 0, 0, 0); put64(w3, v); break;
            }
            case DOP_GLOP: { const uint64_t A = get64(w1, 0), B = get64(w2, 0), C = get64(w3, 0); sink.rec((int)aux, A, B, C, 0); put64(w4, gl_reduce128((u128)A * B + C)); break; }
            case DOP_GATE: { const uint64_t A = get64(w1, 0), B = get64(w2, 0), C = get64(w3, 0); sink.rec((int)aux, A, B, C, 0);
                             const u128 v = (u128)A * B + C; const uint32_t o = w4; if (o != NO_SLOT) { st1(o, (uint64_t)v); st1(o + 1, (uint64_t)(v >> 64)); } break; }
            case DOP_REDUCE: { const uint32_t r = w1; const uint64_t lo = get64(r, 0), hi = get64(r, 1); sink.rec(T_REDUCE, lo, hi, 0, 0); put64(w2, gl_reduce128(((u128)hi << 64) | lo)); break; }
            case DOP_CLT: { sink.rec(T_CLT_SAFE, get64(w1, 0), 0, 0, 0); break; }
            case DOP_FR_ADD: { const fr_t v = be.fr_add(getfr(w1), getfr(w2)); putfr(w3, v); break; }
            case DOP_FR_MUL: { const fr_t v = be.fr_mul(getfr(w1), getfr(w2)); putfr(w3, v); break; }
            case DOP_FR_MULADD: { const fr_t v = be.fr_mul_add(getfr(w1), getfr(w2), getfr(w3)); putfr(w4, v); break; }
            case DOP_SELECT: { const uint64_t v = be.select(get64(w1, 0), get64(w2, 0), get64(w3, 0)); put64(w4, v); break; }
            case DOP_FR_SELECT: { const fr_t v = be.fr_select(getfr(w1), getfr(w2), get64(w3, 0)); putfr(w4, v); break; }
            case DOP_IDX2IND: { be.idx_to_indicator(get64(w1, 0), (int)n, ta); const uint32_t o = w2; for (uint32_t i = 0; i < n; i++) st1(o + i, ta[i]); break; }
            case DOP_SELIND: { for (uint32_t i = 0; i < n; i++) { ta[i] = get64(tw(tape, pc + 1 + i), 0); tb[i] = get64(tw(tape, pc + 1 + n + i), 0); }
                               put64(tw(tape, pc + 1 + 2 * n), be.select_by_indicator(ta, 1, tb, (int)n)); break; }
            case DOP_FR_SELIND: {      // GateChip::select_by_indicator on native values: [0, a0, ind0, s0, a1, ind1, s1, ...] (gates at 3 i)
                fr_t sum = fr_zero(); if (n > 0) be.G(); be.cell64(0);
                for (uint32_t i = 0; i < n; i++) { const fr_t a = getfr(tw(tape, pc + 1 + i)); const uint64_t ind = get64(tw(tape, pc + 1 + n + i), 0); if (ind) sum = h2w::fr_add(sum, a); be.cell(a); be.cell64(ind); if (i + 1 < n) be.G(); be.cell(sum); }
                putfr(tw(tape, pc + 1 + 2 * n), sum); break;
            }
            case DOP_NUM2BITS: { be.num_to_bits(get64(w1, 0), (int)n, ta); const uint32_t o = w2; for (uint32_t i = 0; i < n; i++) st1(o + i, ta[i]); break; }
            case DOP_BITS2NUM: { for (uint32_t i = 0; i < n; i++) ta[i] = get64(tw(tape, pc + 1 + i), 0); put64(tw(tape, pc + 1 + n), be.bits_to_num(ta, (int)n)); break; }
            case DOP_DECOMP565: { be.decompose_le_56_5(getfr(w1), ta); const uint32_t o = w2; for (int i = 0; i < 5; i++) st1(o + i, ta[i]); break; }
            case DOP_LIMBS2NUM: { for (uint32_t i = 0; i < n; i++) ta[i] = get64(tw(tape, pc + 1 + i), 0); putfr(tw(tape, pc + 1 + n), be.limbs_to_num(ta, (int)n)); break; }
            case DOP_RANGE: { be.range_check(get64(w1, 0), (int)aux); break; }
            case DOP_GLOPRUN: {      // the Goldilocks-level ops of a gadget come in runs (a Poseidon round: hash/poseidon/permutation.rs:43-239): no dispatch between them, the next op's words requested before this one is computed
                uint32_t q = pc + 2; uint32_t a = w2, b = w3, c = w4, o = tw(tape, pc + 5);
                rec_t *rp = sink.recs + sink.nrec;      // (the cell cursor is only needed by direct cells: it moves by the run's total, from the tape, once)
#pragma unroll 1
                for (uint32_t i = 0; i < n; i++) {
                    q += 4;
                    const uint32_t na = tw(tape, q), nb = tw(tape, q + 1), nc = tw(tape, q + 2), no = tw(tape, q + 3);      // (past the last op: the next op's first words, unused)
                    const uint64_t A = get64(a, 0), B = get64(b, 0), C = get64(c, 0);
                    g_store_rec(rp, A, B, C, 0); rp++; st1(o & 0xffffffu, gl_reduce128((u128)A * B + C));
                    a = na; b = nb; c = nc; o = no;
                }
                sink.nrec += n; sink.cell_off += w1;
                break;
            }
            case DOP_FETCH: { for (uint32_t j = 0; j < n; j++) ring_put(w2 + j, slow_get64(w1, j, vals_g, Lt, p, imp0, in0, proof)); break; }      // a far operand into the ring (no write-through: a copy)
            default: be.fail(99); break;      // (unreachable: the lowering emits nothing else)
        }
        pc = pcn; w0 = n0; w1 = n1; w2 = n2; w3 = n3; w4 = n4;
    }
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
    long last_const_at = -1; uint64_t last_const_cell = 0;      // the op emitted last is a static CONST1 (its tape position, its cell): a GLOP right behind it that takes it as operand A fuses with it
};
static uint64_t rc_cells(int L, uint64_t bits) { if (bits == 0) return 0; const uint64_t n = (bits + L - 1) / L, rem = bits % L; return (n > 1 ? 1 + 3 * (n - 1) : 0) + (rem ? 4 : 0); }

}  // namespace h2w

using namespace h2w;

namespace h2w {
uint64_t traced_workspace_bytes(const h2w_plan *p, uint64_t n);
int traced_run(h2w_plan *p, const uint64_t *proofs_dev, uint64_t n_proofs, void *advice_dev, void *workspace_dev, void *stream_, ColMap cm, uint64_t cell_stride);
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
    if (t->d_tm) (void)hipFree(t->d_tm); if (t->d_prefix) (void)hipFree(t->d_prefix);
    if (t->d_tape) (void)hipFree(t->d_tape); if (t->d_insts) (void)hipFree(t->d_insts); if (t->d_imps) (void)hipFree(t->d_imps);
    if (t->d_inputs) (void)hipFree(t->d_inputs); if (t->d_pool64) (void)hipFree(t->d_pool64); if (t->d_poolfr) (void)hipFree(t->d_poolfr);
    delete t; p->traced = nullptr;
}
int traced_run(h2w_plan *p, const uint64_t *proofs_dev, uint64_t n_proofs, void *advice_dev, void *workspace_dev, void *stream_, ColMap cm, uint64_t cell_stride) {
    TracedPlan *t = p->traced;
    if (n_proofs > 65535) { set_error("h2w_fri_witness_batch: more than 65535 proofs per call"); return -1; }
    DeviceGuard dg(p->device);
    hipStream_t stream = (hipStream_t)stream_;
    const TracedWs wl = traced_ws(p, n_proofs); char *ws = (char *)workspace_dev;
    ReplayArgs R; memset(&R, 0, sizeof(R));
    R.tape = t->d_tape; R.insts = t->d_insts; R.imps = t->d_imps; R.inputs = t->d_inputs; R.pool64 = t->d_pool64; R.poolfr = t->d_poolfr;
    R.proofs = proofs_dev; R.proof_words = p->pl.total; R.recs = (rec_t *)(ws + wl.recs); R.rec_stride = p->nrec; R.out = (fr_t *)advice_dev; R.cell_stride = cell_stride; R.cm = cm;      // (cm.starts: the FlexGate columns of every proof, cell_stride = ncols << k; else the flat stream)
    R.vals = (uint64_t *)(ws + wl.vals); R.status = (uint32_t *)(ws + wl.status); R.ncells = p->d_ncells; R.inv_pos = p->d_inv; R.inv_neg = p->d_inv + INV_TAB; R.P = p->P; R.L = p->shape.lookup_bits;
    R.nproofs = (uint32_t)n_proofs;
    R.ntmpl = (uint32_t)t->tmpls.size(); R.tm = t->d_tm; R.prefix = t->d_prefix; R.npool64 = t->npool64; R.npoolfr = t->npoolfr;
    H2W_HIP(hipMemsetAsync(ws + wl.status, 0, n_proofs * 4, stream));
    H2W_HIP(hipMemsetAsync(ws + wl.lflag, 0, n_proofs * 4, stream));
    uint32_t maxd = 0; for (const TmplD &T : t->tmpls) if (T.depth > maxd) maxd = T.depth;
    for (uint32_t d = 0; d <= maxd; d++) {      // a segment reads its ancestors' values: depth by depth; the templates of one depth in one launch
        uint32_t nb = 0;
        for (size_t i = 0; i < t->tmpls.size(); i++) { R.blk0[i] = nb; if (t->tmpls[i].depth == d) nb += (uint32_t)((n_proofs * t->tmpls[i].ninst + 63) / 64); }
        R.blk0[t->tmpls.size()] = nb; R.depth = d;
        if (nb) hipLaunchKernelGGL(k_replay, dim3(nb), dim3(64), 0, stream, R);
    }
    // expansion of the block records
    ExpandArgs E;
    E.meta = p->d_meta; E.recs = R.recs; E.nrec = p->nrec; E.rec_stride = p->nrec; E.out = R.out; E.cell_stride = cell_stride; E.pool = nullptr; E.cm = cm;
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
                SegInfo &Pn = segs[(size_t)prev_seg]; Pn.tape.push_back(DOP_SKIP | (5u << 24)); Pn.last_const_at = -1; open_child[(size_t)s] = (int)Pn.tape.size(); for (int k = 0; k < 4; k++) Pn.tape.push_back(0);
                S.cell0 = o.cell0; S.rec0 = nrec; S.started = true;
            }
            if (o.code == TR_SCOPE_POP && s != prev_seg && segs[(size_t)prev_seg].parent == s) {
                SegInfo &C = segs[(size_t)prev_seg]; C.ncells = o.cell0 - C.cell0; C.tape.push_back(DOP_END | (1u << 24));
                // the child's totals include its own children's (they are nested in its cell and record ranges)
                const uint64_t nr = nrec - C.rec0; uint32_t *w = S.tape.data() + open_child[(size_t)prev_seg];
                w[0] = (uint32_t)nr; w[1] = (uint32_t)(nr >> 32); w[2] = (uint32_t)C.ncells; w[3] = (uint32_t)(C.ncells >> 32);
            }
            prev_seg = s; continue;
        }
        prev_seg = s;
        const TraceIn *in = tr->ins.data() + o.first_in; const uint64_t *out = tr->outs.data() + o.first_out;
        std::vector<uint32_t> &T = S.tape; uint64_t want_cells = 0; int w0 = 0, w1 = 0, w2 = 0;
        size_t op_at = T.size();
        auto head = [&](uint32_t op, uint32_t n = 0, uint32_t aux = 0) { if (aux > 255) bad("internal: op parameter too wide"); T.push_back(op | (n << 8) | (aux << 16)); };
        switch (o.code) {
            case TR_LOAD_CONSTANT: {
                const fr_t c = tr->consts[(size_t)o.imm]; const bool small = (c.l[1] | c.l[2] | c.l[3]) == 0;
                if (o.tag.kind == 1) {
                    if (o.tag.n != 1) { bad("a constant that is a proof value must be one Goldilocks word"); break; }
                    S.inputs.push_back((uint32_t)o.tag.word); head(DOP_CONST1); T.push_back(mkref(RK_INPUT, W64, (uint32_t)S.inputs.size() - 1)); T.push_back(new_val(s, out[0], W64, false, 0));
                    add_rec(s, T_CONST1, o.cell0); want_cells = 1;
                } else if (o.tag.kind != 0) bad("a hint tag on a constant");
                else if (small) { const uint32_t li = lit64(c.l[0]); head(DOP_CONST1); T.push_back(mkref(RK_LIT64, W64, li)); T.push_back(NO_SLOT); new_val(s, out[0], W64, true, li); add_rec(s, T_CONST1, o.cell0); want_cells = 1;
                                  S.last_const_at = (long)op_at; S.last_const_cell = o.cell0; }
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
                if (ref_kind(a) != RK_LOCAL && ref_kind(a) != RK_IMPORT && ref_kind(a) != RK_RING) { bad("gl_reduce of a constant"); break; }
                head(DOP_REDUCE); T.push_back(a); T.push_back(new_val(s, out[0], W64, false, 0)); add_rec(s, T_REDUCE, o.cell0); want_cells = (uint64_t)tt.ncells(T_REDUCE); break;
            }
            case TR_GLOP: {
                const uint32_t a = ref_of(s, in[0], &w0), b = ref_of(s, in[1], &w1), c3 = ref_of(s, in[2], &w2);
                if (w0 != W64 || w1 != W64 || w2 != W64) { bad("a Goldilocks op on a wide value"); break; }
                if (o.sub == T_GLOP && !in[0].lit && S.last_const_at >= 0 && (size_t)S.last_const_at + 3 == T.size() && in[0].v == S.last_const_cell && S.last_const_cell + 1 == o.cell0 && ref_kind(a) == RK_LIT64) {
                    // load_constant(K) immediately followed by the op that takes it (GoldilocksChip's constant operands, e.g. hash/poseidon/permutation.rs:55-68): ONE record [K][C, A, B, V]...
                    op_at = (size_t)S.last_const_at; T.resize(op_at); meta.pop_back(); nrec--; S.nrecs--;
                    T.push_back(DOP_GLOP | ((uint32_t)T_KA_GLOP << 16)); T.push_back(a); T.push_back(b); T.push_back(c3); T.push_back(new_val(s, out[0], W64, false, 0)); add_rec(s, T_KA_GLOP, o.cell0 - 1);
                    S.last_const_at = -1; want_cells = (uint64_t)tt.ncells(T_GLOP); break;
                }
                head(DOP_GLOP, 0, o.sub); T.push_back(a); T.push_back(b); T.push_back(c3); T.push_back(new_val(s, out[0], W64, false, 0)); add_rec(s, o.sub, o.cell0); want_cells = (uint64_t)tt.ncells(o.sub); break;
            }
            default: bad("unknown op in the trace");
        }
        if (!(o.code == TR_LOAD_CONSTANT && S.last_const_at == (long)op_at)) S.last_const_at = -1;
        if (T.size() > op_at) {
            const size_t len = T.size() - op_at; if (len > 255) bad("internal: op too long"); T[op_at] |= (uint32_t)len << 24;
            // operands: those in the lane's ring or in the LDS part of the pools become fast refs; the others are fetched into fresh ring slots by DOP_FETCH
            // ops in front of this one.  The ring holds the last RING_K slots WRITTEN, the temporaries included: decided against the slot count after them.
            uint32_t first, count; operand_span(T[op_at] & 0xff, (T[op_at] >> 8) & 0xff, first, count);
            auto in_lds = [&](uint32_t r, uint32_t nslots_final) {
                const int k = ref_kind(r); const uint32_t i = ref_idx(r);
                if (k == RK_LOCAL) return (uint64_t)i + RING_K >= (uint64_t)nslots_final;
                if (k == RK_LIT64) return i + 1 <= POOL64_CAP;
                if (k == RK_LITFR) return i + 1 <= POOLFR_CAP;
                return false;
            };
            auto words_of = [&](uint32_t r) { const int k = ref_kind(r); return (uint32_t)(k == RK_LITFR ? 4 : k == RK_INPUT ? (ref_width(r) == WFR ? 4 : 1) : SLOTS_OF[ref_width(r)]); };
            uint32_t temps = 0;
            for (;;) { uint32_t need = 0; for (uint32_t k2 = 0; k2 < count; k2++) { const uint32_t r = T[op_at + first + k2]; if (!in_lds(r, S.nslots + temps)) need += words_of(r); } if (need == temps) break; temps = need; }
            std::vector<uint32_t> fetches; const uint32_t nf = S.nslots + temps; uint32_t tslot = S.nslots;
            for (uint32_t k2 = 0; k2 < count; k2++) {
                const uint32_t r = T[op_at + first + k2]; const int k = ref_kind(r), w = ref_width(r); uint32_t fr2;
                if (in_lds(r, nf)) fr2 = k == RK_LOCAL ? fastref_ring(w, ref_idx(r)) : k == RK_LIT64 ? fastref_pool(W64, LDS_POOL64 + ref_idx(r) * 8) : fastref_pool(WFR, LDS_POOLFR + ref_idx(r) * 32);
                else {
                    const uint32_t nw = words_of(r);
                    fetches.push_back(DOP_FETCH | (nw << 8) | (3u << 24)); fetches.push_back(r); fetches.push_back(tslot);
                    fr2 = fastref_ring(nw == 4 ? WFR : nw == 2 ? W128 : W64, tslot); tslot += nw;
                }
                T[op_at + first + k2] = fr2;
            }
            S.nslots = nf;
            if (!fetches.empty()) { T.insert(T.begin() + (long)op_at, fetches.begin(), fetches.end()); if (S.last_const_at == (long)op_at) S.last_const_at += (long)fetches.size(); }
        }
        if (err.empty() && want_cells != o.ncells) bad("internal: op " + std::to_string(o.code) + " appended " + std::to_string(o.ncells) + " cells on the host, the device template has " + std::to_string(want_cells));
    }
    if (!err.empty()) { set_error("h2w_plan_from_trace: " + err); return nullptr; }
    segs[0].tape.push_back(DOP_END | (1u << 24)); segs[0].cell0 = 0; segs[0].rec0 = 0;
    // ---- consecutive Goldilocks-level ops -> runs (DOP_GLOPRUN)
    for (SegInfo &S : segs) {
        if (S.nslots >= (1u << 24)) continue;
        std::vector<uint32_t> out; out.reserve(S.tape.size()); size_t pc = 0; const std::vector<uint32_t> &T = S.tape;
        while (pc < T.size()) {
            const uint32_t h = T[pc], op = h & 0xff, len = h >> 24;
            if (op != DOP_GLOP) { out.insert(out.end(), T.begin() + (long)pc, T.begin() + (long)(pc + len)); pc += len; continue; }
            size_t e = pc; uint32_t cnt = 0;
            while (e < T.size() && (T[e] & 0xff) == DOP_GLOP && cnt < 255) { e += 5; cnt++; }
            if (cnt < 2) { out.insert(out.end(), T.begin() + (long)pc, T.begin() + (long)(pc + 5)); pc += 5; continue; }
            out.push_back(DOP_GLOPRUN | (cnt << 8));
            { uint32_t cells = 0; for (size_t k2 = pc; k2 < e; k2 += 5) cells += (uint32_t)tt.ncells((int)((T[k2] >> 16) & 0xff)); out.push_back(cells); }
            for (size_t k2 = pc; k2 < e; k2 += 5) { out.push_back(T[k2 + 1]); out.push_back(T[k2 + 2]); out.push_back(T[k2 + 3]); out.push_back((T[k2 + 4] & 0xffffffu) | (((T[k2] >> 16) & 0xff) << 24)); }
            pc = e;
        }
        S.tape.swap(out);
    }
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
    // ---- every word the device will follow, checked here: op lengths, operand kinds and indices, result slots (a wild reference is a GPU fault)
    for (size_t t = 0; t < members.size() && err.empty(); t++) {
        const SegInfo &M = segs[(size_t)members[t][0]]; const std::vector<uint32_t> &T = M.tape; size_t pc = 0; bool ended = false;
        auto okslow = [&](uint32_t r) {      // the old-style ref a DOP_FETCH carries
            const uint32_t i = ref_idx(r); const int k = ref_kind(r), w = ref_width(r);
            if (w > WFR) return false;
            const uint32_t span = k == RK_LITFR ? 4u : (uint32_t)(k == RK_INPUT ? (w == WFR ? 4 : 1) : SLOTS_OF[w]);
            switch (k) {
                case RK_LOCAL: return (uint64_t)i + span <= M.nslots;
                case RK_IMPORT: return i < M.imps.size();
                case RK_LIT64: return (uint64_t)i + span <= pool64.size();
                case RK_INPUT: return i < M.inputs.size() && (uint64_t)M.inputs[i] + span <= proof_words;
                case RK_LITFR: return i < poolfr.size();
                default: return false;
            }
        };
        auto okref = [&](uint32_t r, int) {      // a fast ref
            if (r & 0x60000000u) return false;
            const uint32_t w = (r >> 27) & 3u; if (w > WFR) return false;
            if (r >> 31) return (r & 0x07ffff00u) == 0;
            const uint32_t off = r & 0x07ffffffu, span = w == WFR ? 32u : w == W128 ? 16u : 8u;
            return (off & 7u) == 0 && ((off >= LDS_POOL64 && off + span <= LDS_POOL64 + (uint32_t)std::min<size_t>(pool64.size(), POOL64_CAP) * 8) || (off >= LDS_POOLFR && off + span <= LDS_POOLFR + (uint32_t)std::min<size_t>(poolfr.size(), POOLFR_CAP) * 32));
        };
        auto okout = [&](uint32_t slot, uint32_t nsl) { return slot == NO_SLOT || (uint64_t)slot + nsl <= M.nslots; };
        while (pc < T.size()) {
            const uint32_t h = T[pc], op = h & 0xff, n = (h >> 8) & 0xff, len = op == DOP_GLOPRUN ? 2 + 4 * n : h >> 24; bool ok = len >= 1 && pc + len <= T.size();
            if (op == DOP_END) { ended = ok && pc + 1 == T.size(); break; }
            auto R_ = [&](size_t k2) { return T[pc + k2]; };
            if (ok) switch (op) {
                case DOP_SKIP: ok = len == 5; break;
                case DOP_GLOPRUN: ok = n >= 2; for (uint32_t k2 = 0; ok && k2 < n; k2++) ok = okref(R_(2 + 4 * k2), 1) && okref(R_(3 + 4 * k2), 1) && okref(R_(4 + 4 * k2), 1) && (uint64_t)(R_(5 + 4 * k2) & 0xffffffu) + 1 <= M.nslots && (R_(5 + 4 * k2) >> 24) < T_DYNAMIC; break;
                case DOP_FETCH: ok = len == 3 && n >= 1 && n <= 4 && okslow(R_(1)) && (uint64_t)R_(2) + n <= M.nslots; break;
                case DOP_CONST1: case DOP_LOADW: ok = len == 3 && okref(R_(1), 1) && okout(R_(2), 1); break;
                case DOP_FRCELL: ok = len == 3 && okref(R_(1), 4) && okout(R_(2), 4); break;
                case DOP_LOADW_DIV: case DOP_LOADW_EXTINV: ok = len == 4 && okref(R_(1), 1) && okref(R_(2), 1) && okout(R_(3), 1); break;
                case DOP_GLOP: ok = len == 5 && okref(R_(1), 1) && okref(R_(2), 1) && okref(R_(3), 1) && okout(R_(4), 1) && ((h >> 16) & 0xff) < T_DYNAMIC; break;
                case DOP_GATE: ok = len == 5 && okref(R_(1), 1) && okref(R_(2), 1) && okref(R_(3), 1) && okout(R_(4), 2); break;
                case DOP_REDUCE: ok = len == 3 && okref(R_(1), 2) && ((R_(1) >> 27) & 3u) != W64 && okout(R_(2), 1); break;
                case DOP_CLT: case DOP_RANGE: ok = len == 2 && okref(R_(1), 1); break;
                case DOP_FR_ADD: case DOP_FR_MUL: ok = len == 4 && okref(R_(1), 4) && okref(R_(2), 4) && okout(R_(3), 4); break;
                case DOP_FR_MULADD: ok = len == 5 && okref(R_(1), 4) && okref(R_(2), 4) && okref(R_(3), 4) && okout(R_(4), 4); break;
                case DOP_SELECT: ok = len == 5 && okref(R_(1), 1) && okref(R_(2), 1) && okref(R_(3), 1) && okout(R_(4), 1); break;
                case DOP_FR_SELECT: ok = len == 5 && okref(R_(1), 4) && okref(R_(2), 4) && okref(R_(3), 1) && okout(R_(4), 4); break;
                case DOP_IDX2IND: case DOP_NUM2BITS: ok = len == 3 && n >= 1 && n <= 64 && okref(R_(1), 1) && okout(R_(2), n); break;
                case DOP_DECOMP565: ok = len == 3 && okref(R_(1), 4) && okout(R_(2), 5); break;
                case DOP_SELIND: case DOP_FR_SELIND: ok = len == 2 + 2 * n && n >= 1 && n <= 64; for (uint32_t k2 = 0; ok && k2 < 2 * n; k2++) ok = okref(R_(1 + k2), 4); ok = ok && okout(R_(1 + 2 * n), op == DOP_SELIND ? 1 : 4); break;
                case DOP_BITS2NUM: case DOP_LIMBS2NUM: ok = len == 2 + n && n <= 64; for (uint32_t k2 = 0; ok && k2 < n; k2++) ok = okref(R_(1 + k2), 1); ok = ok && okout(R_(1 + n), op == DOP_BITS2NUM ? 1 : 4); break;
                default: ok = false;
            }
            if (!ok) { bad("internal: malformed device tape (template " + std::to_string(t) + ", word " + std::to_string(pc) + ", op " + std::to_string(op) + ")"); break; }
            pc += len;
        }
        if (err.empty() && !ended) bad("internal: a device tape does not end");
    }
    if (!err.empty()) { set_error("h2w_plan_from_trace: " + err); delete tp; return nullptr; }
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
        for (int k2 = 0; k2 < 16; k2++) tape_all.push_back(DOP_END | (1u << 24));      // the interpreter reads a window ahead
        if (put((void **)&tp->d_tape, tape_all.data(), tape_all.size() * 4) != 0) return -1;
        std::vector<uint64_t> prefix(tp->tmpls.size()); uint64_t acc = 0;
        for (size_t i = 0; i < tp->tmpls.size(); i++) { prefix[i] = acc; acc += (uint64_t)tp->tmpls[i].nslots * tp->tmpls[i].ninst; }
        if (put((void **)&tp->d_tm, tp->tmpls.data(), tp->tmpls.size() * sizeof(TmplD)) != 0) return -1;
        if (put((void **)&tp->d_prefix, prefix.data(), prefix.size() * 8) != 0) return -1;
        if (put((void **)&tp->d_insts, insts.data(), insts.size() * sizeof(InstD)) != 0) return -1;
        if (put((void **)&tp->d_imps, imps.data(), imps.size() * sizeof(ImpD)) != 0) return -1;
        if (put((void **)&tp->d_inputs, inputs.data(), inputs.size() * 4) != 0) return -1;
        tp->npool64 = (uint32_t)pool64.size(); tp->npoolfr = (uint32_t)poolfr.size();
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
