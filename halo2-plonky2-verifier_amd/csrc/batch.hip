// batch.hip — API level 3 of include/h2w.h: the batched hot path.
//
//   h2w_plan_compile : shape compiler.  Replays the gadget once on the host with a counting sink
//                      (ValBackend<PlanSink>) to fix the offset of every cell block and strand for this shape.
//   h2w_fri_witness_batch : per batch, on the caller's stream (+ one side stream of the plan)
//        k_prologue_values  one wavefront per proof : witness load, Fiat-Shamir challenger ON VALUES (serial sponge), PoW, reduced
//                           openings -> challenge block, the prologue's direct cells / non-permutation records, the permutation list
//        k_glp_emit         one wavefront per listed Goldilocks-Poseidon permutation -> its 2,604 block records
//        k_strands          one lane per (proof, query) : FRI query glue (glue.hip) -> records
//        PoseidonBN254 caps : k_merkle_bn_values  one quad per (proof, query, tree): the Merkle path ON VALUES -> unit states
//                             k_merkle_bn_emit    one quad per permutation unit of every path: its cells, straight to the advice
//        Goldilocks caps    : k_merkle_gl_values  one wavefront per (proof, query, tree) -> select records, the permutation list
//        expand_fast        block records -> cells: the HBM-write-bound materialisation (expand.hip)
// Data layout in HBM (per batch): proofs [n][proof_words] u64 ; records [n][n_records] 32 B ; challenge blocks [n] ; unit states
// [n][units][4] 32 B ; permutation list [n][perms][13] u64 ; advice [n][n_cells] 32 B canonical-LE Fr.  Record metas / templates /
// Poseidon constants are per-shape and shared.
#include <hip/hip_runtime.h>
#include <vector>
#include <string>
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include "common.h"
#include "batchargs.h"

namespace h2w {


struct PlanSink {
    static constexpr bool kCoop = false, kSplitOnly = false, kBnUnits = false, kDevSponge = false; static constexpr int kHashMode = -1;
    void coop_poseidon_permute(uint64_t *, const h2w_poseidon_consts_t *) {}
    std::vector<uint64_t> *meta; const TemplateTable *tt; StrandTable *st;
    uint64_t nrec = 0, cell_off = 0, cur_q_rec = 0, cur_q_cell = 0, mk_rec0 = 0, mk_cell0 = 0; bool mk_zc = false;
    std::vector<uint64_t> *unit_cell = nullptr; uint64_t nunit = 0, cur_q_unit = 0, mk_unit0 = 0; bool pu_zc = false;
    uint64_t nglp = 0, cur_q_glp = 0, mk_glp0 = 0; int cur_q = -1;
    std::vector<LoadItem> *items = nullptr, *cap_items = nullptr;
    void note_load(uint64_t w, int kind) { LoadItem it; it.word = (uint32_t)w; it.kind = (uint32_t)kind; it.rec = nrec; it.cell = cell_off; items->push_back(it); }
    void note_cap_hash(uint64_t w) { if (cap_items) { LoadItem it; it.word = (uint32_t)w; it.kind = 4; it.rec = nrec; it.cell = cell_off; cap_items->push_back(it); } }
    bool coop_load_proof(const ValCfg &) { return false; }
    bool bn_emit_inline(fr_t *, const ValCfg &, bool &) { return false; }
    void bn_perm_begin(bool zc) { unit_cell->push_back(cell_off); pu_zc = zc; }
    void bn_perm_end(bool zc) { if (!pu_zc && zc) st->first_zero_unit = (int64_t)nunit; nunit++; }
    void glp_note() { nglp++; }
    // keygen metadata pass (h2w_plan_metadata): one bit per cell, set from the template slot flags / the backend's G()/LK() markers
    std::vector<uint8_t> *sel_bits = nullptr, *lk_bits = nullptr; uint8_t pend = 0;
    void mark(uint64_t cell, uint8_t f) {
        if (f & CF_GATE) { if (sel_bits->size() <= cell / 8) sel_bits->resize(cell / 8 + 4096, 0); (*sel_bits)[cell / 8] |= (uint8_t)(1u << (cell & 7)); }
        if (f & CF_LOOKUP) { if (lk_bits->size() <= cell / 8) lk_bits->resize(cell / 8 + 4096, 0); (*lk_bits)[cell / 8] |= (uint8_t)(1u << (cell & 7)); }
    }
    void gate() { pend |= CF_GATE; }
    void lookup() { pend |= CF_LOOKUP; }
    void rec(int t, uint64_t, uint64_t, uint64_t, uint64_t) {
        if (meta) meta->push_back(meta_pack((uint32_t)t, cell_off));
        if (sel_bits) { const tmpl_info_t &ti = tt->info[t]; for (int i = 0; i < ti.ncells; i++) { const uint8_t f = tt->slot_flags[ti.slot_base + i]; if (f) mark(cell_off + i, f); } }
        nrec++; cell_off += (uint64_t)tt->ncells(t);
    }
    void cell(const fr_t &) { if (sel_bits && pend) mark(cell_off, pend); pend = 0; cell_off++; }
    void skip(uint64_t, uint64_t) {}
    void merkle_begin(int, int, bool zc, uint64_t) { mk_rec0 = nrec; mk_cell0 = cell_off; mk_zc = zc; mk_unit0 = nunit; mk_glp0 = nglp; }
    void merkle_end(int q, int kind, bool zc) {
        if (q > 1) return;
        st->mk_rec_rel[q][kind] = mk_rec0 - cur_q_rec; st->mk_cell_rel[q][kind] = mk_cell0 - cur_q_cell;
        st->mk_nrec[q][kind] = nrec - mk_rec0; st->mk_ncell[q][kind] = cell_off - mk_cell0; st->mk_unit_rel[q][kind] = mk_unit0 - cur_q_unit;
        st->mk_nunit[kind] = (uint32_t)(nunit - mk_unit0); st->mk_glp_rel[kind] = (uint32_t)(mk_glp0 - cur_q_glp); st->mk_nglp[kind] = (uint32_t)(nglp - mk_glp0);
        if (!mk_zc && zc) st->first_zero_kind = (q == 0) ? kind : -2;
    }
    void query_begin(int q, uint64_t) {
        cur_q_rec = nrec; cur_q_cell = cell_off; cur_q_unit = nunit; cur_q_glp = nglp;
        if (q == 0) st->pro_nglp = (uint32_t)nglp;
        if (q <= 1) { st->q_rec0[q] = nrec; st->q_cell0[q] = cell_off; st->q_unit0[q] = nunit; }
    }
    void query_end(int q, uint64_t) { if (q <= 1) { st->q_nrec[q] = nrec - cur_q_rec; st->q_ncell[q] = cell_off - cur_q_cell; st->q_nunit[q] = nunit - cur_q_unit; st->q_nglp = (uint32_t)(nglp - cur_q_glp); } }
};

// WitnessChip::load_proof_with_pis (witness/mod.rs:267-294) and the limb decompositions of the caps' BN254 hashes (challenger/mod.rs:65-74,
// hash/poseidon_bn254/hash.rs:31-43): every item is a function of a few proof words - one lane per (proof, item).  Every rank checks every
// proof's words (status 4); the proof's owner writes the records and cells.
template <bool COLS> __global__ __launch_bounds__(256) void k_prologue_load(BatchArgs A) {
    const int p = blockIdx.y; const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n_load_items + A.n_cap_items) return;
    const bool emit = own_prologue(A, p);
    rec_t *recs = A.recs + (uint64_t)p * A.rec_stride; fr_t *out = block_out(A, p, -1);
    const uint64_t *ip = reinterpret_cast<const uint64_t *>(A.load_items + i);
    const uint64_t wk = g_load_u64(ip), irec = g_load_u64(ip + 1), icell = g_load_u64(ip + 2);
    const uint32_t word = (uint32_t)wk, kind = (uint32_t)(wk >> 32);
    const uint64_t *w = A.proofs + (uint64_t)p * A.proof_words + word; const uint64_t w0 = g_load_u64(w);
    bool bad = false;
    if (kind <= 1) { if (emit) g_store_rec(recs + irec, w0, 0, 0, 0); bad = w0 >= GL_P; }
    else {
        const uint64_t w1 = g_load_u64(w + 1), w2 = g_load_u64(w + 2), w3 = g_load_u64(w + 3);
        if (kind == 2) { if (emit) g_store_rec(recs + irec, w0, w1, w2, w3); bad = w0 >= GL_P || w1 >= GL_P || w2 >= GL_P || w3 >= GL_P; }
        else {
            fr_t v; v.l[0] = w0; v.l[1] = w1; v.l[2] = w2; v.l[3] = w3;
            if (kind == 3) { ColPolicy<COLS> cc; cc.init(A.cm); if (emit) g_store_fr(out + cc.map(icell), v); bad = fr_geq_mod(v); }
            else if (emit) {      // kind 4: RangeChip::decompose_le(x, 56, 5) - 13 cells of limb sums + 5 range checks
                DevSinkT<COLS> sink; sink.recs = recs; sink.nrec = irec; sink.out = out; sink.cell_off = icell; sink.ncells = A.ncells; sink.cc.init(A.cm);
                ValBackend<DevSinkT<COLS>> be(sink, make_cfg(A, p), true);
                uint64_t limbs[5]; be.decompose_le_56_5(v, limbs);
            }
        }
    }
    if (bad) atomicOr(&A.load_flag[p], 4u);
}

// one wavefront per LISTED Goldilocks-Poseidon permutation of this rank's blocks: the prologues' (every owned proof), then - Goldilocks
// caps - the Merkle strands' (every owned (proof, query) unit)
template <bool COLS> __global__ __launch_bounds__(64) void k_glp_emit(BatchArgs A) {
    typedef CoopSinkT<COLS, false> Sink;
    stage_glp_consts(A.consts, threadIdx.x, 64);
    unsigned b = blockIdx.x; int p, slot;
    const unsigned n_pro = A.sh.n_own_proofs * A.st->pro_nglp;
    if (b < n_pro) { p = (int)(b / A.st->pro_nglp) * A.sh.world + A.sh.rank; slot = (int)(b % A.st->pro_nglp); }
    else {
        b -= n_pro; int q;
        if (A.st->q_nglp == 0 || !own_unit_at(A, b / A.st->q_nglp, p, q)) return;
        slot = (int)(A.st->pro_nglp + (unsigned)q * A.st->q_nglp + b % A.st->q_nglp);
    }
    Sink sink; coop_sink_init(sink, A, p, -1); sink.cell_off = 0; sink.emit = true;
    const uint64_t *e = A.glp_list + ((uint64_t)p * A.st->total_glp + (uint64_t)slot) * GLP_LIST_WORDS;
    const uint64_t w = threadIdx.x < GLP_LIST_WORDS ? g_load_u64(e + threadIdx.x) : 0;
    uint64_t st[SPONGE_WIDTH];
#pragma unroll
    for (int i = 0; i < SPONGE_WIDTH; i++) st[i] = readlane64(w, i + 1);
    sink.nrec = readlane64(w, 0);
    sink.coop_poseidon_permute(st, A.consts);
}

// between the two passes: the S-box values of the owned units' partial rounds, as the values pass left them (times R), to the canonical
// values the emission shows - one lane per value, 168 per permutation unit
__global__ __launch_bounds__(256) void k_sbox_canon(BatchArgs A, uint32_t units_per_query) {
    const uint64_t per_q = (uint64_t)units_per_query * (BN_PARTIAL_ROUNDS * 3);
    const uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const unsigned i = (unsigned)(idx / per_q); const uint64_t v = idx % per_q;
    int p, q; if (!own_unit_at(A, i, p, q)) return;
    if (v >= A.st->q_nunit[q == 0 ? 0 : 1] * (uint64_t)(BN_PARTIAL_ROUNDS * 3)) return;
    fr_t *const x = A.unit_sbox + (uint64_t)i * A.sh.unit_slot * (BN_PARTIAL_ROUNDS * 3) + v;
    g_store_fr(x, fr_mont_mul(g_load_fr(x), fr_from_u64(1), A.P.ninv));
}
// the same for the row-cooperative values pass (rowperm.h): it leaves every S-box value as twelve dwords of 29-bit limbs (times R, lazy, limbs a few
// units above 2^29): carry pass, one Montgomery product by 1, one conditional subtraction -> the canonical value, into unit_sbox
__global__ __launch_bounds__(256) void k_sbox_canon9(BatchArgs A, uint32_t units_per_query) {
    const uint64_t per_q = (uint64_t)units_per_query * (BN_PARTIAL_ROUNDS * 3);
    const uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const unsigned i = (unsigned)(idx / per_q); const uint64_t v = idx % per_q;
    int p, q; if (!own_unit_at(A, i, p, q)) return;
    if (v >= A.st->q_nunit[q == 0 ? 0 : 1] * (uint64_t)(BN_PARTIAL_ROUNDS * 3)) return;
    const uint64_t at = (uint64_t)i * A.sh.unit_slot * (BN_PARTIAL_ROUNDS * 3) + v;
    const uint4 *src = reinterpret_cast<const uint4 *>(A.unit_sbox9 + at * rf::SBX9_W);
    const uint4 a = src[0], b = src[1], c4 = src[2];
    const uint32_t t[9] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c4.x};
    fr9_t n; uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { const uint32_t x = t[j] + c; n.t[j] = x & rf::M29; c = x >> 29; }
    n.t[8] = t[8] + c;
    fr9_t one9;
#pragma unroll
    for (int j = 0; j < 9; j++) one9.t[j] = j == 0 ? 1u : 0u;
    fr_t s = fr9_pack(fr9_mont(n, one9, (uint32_t)A.P.ninv & rf::M29));
    if (fr_geq_mod(s)) s = fr_sub_mod_raw(s);
    g_store_fr(A.unit_sbox + at, s);
}
// one pass (H2W_OPT_CHAIN_PASSES 1): four lanes per (owned unit, kind) walk the path and emit every unit of it - the least arithmetic per
// cell (352 wavefront-level products per permutation, none twice), serial in the path's depth; blockIdx.y = kind slot
template <bool COLS> __global__ __launch_bounds__(QUAD_BLOCK) H2W_QUAD_ATTR void k_merkle_bn_fused(BatchArgs A) {
    typedef QuadSinkT<COLS, QUAD_FUSED> Sink; typedef ValBackend<Sink> QuadB;
    stage_bn_consts(A.bn_tab, threadIdx.x, QUAD_BLOCK);
    const unsigned total = A.sh.n_own_units;
    if (((blockIdx.x * QUAD_BLOCK + (threadIdx.x & ~63u)) >> 2) >= total) return;
    unsigned idx = (blockIdx.x * QUAD_BLOCK + threadIdx.x) >> 2;
    if (idx >= total) idx = total - 1;
    int p, q; own_unit_at(A, idx, p, q);
    const int n_or = A.shape.n_perm_z > 0 ? 3 : 2;
    const int slot = blockIdx.y, kind = slot < n_or ? slot : 3 + (slot - n_or);
    Sink sink;
    quad_strand<QuadB>(A, sink, idx, p, q, kind);
}
__global__ void k_digest(const ulonglong4 *cells, uint64_t n, unsigned long long *out4) {
    unsigned long long a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        ulonglong4 c = cells[i]; unsigned long long m = (i + 1) * 0x9E3779B97F4A7C15ULL | 1ULL;
        a0 += c.x * m; a1 += c.y * (m + 2); a2 += c.z * (m + 4); a3 += c.w * (m + 6);
    }
    for (int d = 32; d > 0; d >>= 1) { a0 += __shfl_down(a0, d, 64); a1 += __shfl_down(a1, d, 64); a2 += __shfl_down(a2, d, 64); a3 += __shfl_down(a3, d, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out4[0], a0); atomicAdd(&out4[1], a1); atomicAdd(&out4[2], a2); atomicAdd(&out4[3], a3); }
}

}  // namespace h2w

using namespace h2w;

#include "plan.h"

namespace h2w {
// a traced plan (replay.hip)
uint64_t traced_workspace_bytes(const h2w_plan *p, uint64_t n);
uint64_t traced_status_offset(const h2w_plan *p, uint64_t n, bool flags);
int traced_run(h2w_plan *p, const uint64_t *proofs_dev, uint64_t n_proofs, void *advice_dev, void *workspace_dev, void *stream_, ColMap cm, uint64_t cell_stride);
void traced_free(h2w_plan *p);
PlanEqualities &plan_equalities(h2w_plan *p) { return p->eqs; }
const h2w_shape_t &plan_shape(const h2w_plan *p) { return p->shape; }
const h2w_poseidon_consts_t &plan_consts(const h2w_plan *p) { return p->h_consts; }
uint64_t plan_cells(const h2w_plan *p) { return p->ncells; }
}
static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

extern "C" {

h2w_plan *h2w_plan_compile(const h2w_shape_t *shape, const h2w_poseidon_consts_t *consts, int device_id) {
    if (!shape || !consts) { set_error("h2w_plan_compile: null argument"); return nullptr; }
    const h2w_shape_t &s = *shape;
    if (const char *why = shape_check(s)) { set_error(std::string("h2w_plan_compile: unsupported shape: ") + why); return nullptr; }
    h2w_plan *pl = new h2w_plan(s.lookup_bits);
    pl->shape = s; pl->device = device_id; pl->P = fr_params_init(); pl->h_consts = *consts;
    pl->d = derive_shape(s); pl->pl = proof_layout(s, pl->d);
    memset(&pl->st, 0, sizeof(pl->st)); pl->st.first_zero_kind = -1; pl->st.first_zero_unit = -1;
    std::vector<uint64_t> unit_cell; std::vector<LoadItem> items, cap_items;
    // host inverse table
    std::vector<fr_t> inv(2 * INV_TAB, fr_zero());
    for (int k2 = 1; k2 < INV_TAB; k2++) { inv[k2] = fr_inv(fr_from_u64((uint64_t)k2), pl->P); inv[INV_TAB + k2] = fr_neg(inv[k2]); }
    // shape compile: sequential replay with the counting sink on an all-zero proof
    std::vector<uint64_t> meta; std::vector<uint64_t> zero_proof(pl->pl.total, 0);
    {
        PlanSink sink; sink.meta = &meta; sink.tt = &pl->tt; sink.st = &pl->st; sink.unit_cell = &unit_cell; sink.items = &items; sink.cap_items = &cap_items;
        ValCfg cfg; cfg.proof = zero_proof.data(); cfg.mode = s.hash_mode; cfg.L = s.lookup_bits; cfg.P = pl->P;
        cfg.inv_pos = inv.data(); cfg.inv_neg = inv.data() + INV_TAB; cfg.st = nullptr; cfg.split = false; cfg.split_bn = false; cfg.load_items = nullptr; cfg.n_load_items = 0; cfg.load_nrec = cfg.load_ncell = 0;
        ValBackend<PlanSink> be(sink, cfg, false);
        Verifier<ValBackend<PlanSink>> V(be, pl->shape, consts);
        ChallengeBlock<ValBackend<PlanSink>> *cb = new ChallengeBlock<ValBackend<PlanSink>>();
        V.run_all(*cb);
        delete cb;
        pl->n_items = (uint32_t)items.size(); pl->n_cap_items = s.hash_mode == 1 ? (uint32_t)cap_items.size() : 0;
        if (!items.empty()) {   // records / cells of the load phase: from the first item to the end of the last one
            // the load phase starts right after the 12 zero-state constants and is contiguous in records and cells
            const LoadItem &last = items.back();
            uint64_t last_nrec = last.kind == 3 ? 0 : 1, last_ncell = last.kind == 0 ? (uint64_t)pl->tt.ncells(T_LOADW) : last.kind == 1 ? 1 : last.kind == 2 ? 4 : 1;
            pl->load_nrec = last.rec + last_nrec - items.front().rec; pl->load_ncell = last.cell + last_ncell - items.front().cell;
        }
        pl->nrec = sink.nrec; pl->ncells = sink.cell_off; pl->nunit = sink.nunit; pl->st.total_unit = sink.nunit;
        for (uint64_t m : meta) pl->rec_cells += (uint64_t)pl->tt.ncells((int)meta_tmpl(m));
        pl->st.pro_nrec = pl->st.q_rec0[0]; pl->st.pro_ncell = pl->st.q_cell0[0]; pl->st.total_rec = sink.nrec; pl->st.total_cell = sink.cell_off;
        if (s.num_queries == 1) {
            pl->st.q_unit0[1] = pl->st.q_unit0[0]; pl->st.q_nunit[1] = pl->st.q_nunit[0];
            for (int k2 = 0; k2 < MK_KINDS; k2++) pl->st.mk_unit_rel[1][k2] = pl->st.mk_unit_rel[0][k2];
            pl->st.q_rec0[1] = pl->st.q_rec0[0]; pl->st.q_cell0[1] = pl->st.q_cell0[0]; pl->st.q_nrec[1] = pl->st.q_nrec[0]; pl->st.q_ncell[1] = pl->st.q_ncell[0];
            for (int k2 = 0; k2 < MK_KINDS; k2++) { pl->st.mk_rec_rel[1][k2] = pl->st.mk_rec_rel[0][k2]; pl->st.mk_cell_rel[1][k2] = pl->st.mk_cell_rel[0][k2]; pl->st.mk_nrec[1][k2] = pl->st.mk_nrec[0][k2]; pl->st.mk_ncell[1][k2] = pl->st.mk_ncell[0][k2]; }
        }
        if (pl->st.first_zero_kind == -2) { set_error("h2w_plan_compile: internal: first load_zero outside query 0"); delete pl; return nullptr; }
        // the two-phase strands' static tables: permutation list slots, emission work items of a query
        pl->st.total_glp = (uint32_t)sink.nglp;
        if (s.hash_mode == 1) pl->st.q_nglp = 0;
        if ((uint64_t)pl->st.pro_nglp + (uint64_t)s.num_queries * pl->st.q_nglp != sink.nglp) { set_error("h2w_plan_compile: internal: permutation list layout"); delete pl; return nullptr; }
        uint32_t it = 0;
        for (int k2 = 0; k2 < MK_KINDS; k2++) {
            pl->st.mk_item0[k2] = it;
            const bool exists = k2 < 3 ? k2 < pl->d.n_oracles : k2 - 3 < pl->d.n_steps;
            if (exists) it += pl->st.mk_nunit[k2] ? pl->st.mk_nunit[k2] : 1;
        }
        pl->st.mk_item0[MK_KINDS] = it;
        pl->small_mds = glp_small_mds(*consts);
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        // no GPU: the plan is still usable for layout queries (cells, records, proof words); batch calls fail.
        pl->device = -1; return pl;
    }
    if (device_id < 0 || device_id >= ndev) { set_error("h2w_plan_compile: device_id out of range"); delete pl; return nullptr; }
    DeviceGuard dg(device_id);
    auto up = [&]() -> int {
        if (pl->dt.upload(pl->tt) != 0) return -1;
        H2W_HIP(hipMalloc((void **)&pl->d_meta, meta.size() * sizeof(uint64_t)));
        H2W_HIP(hipMemcpy(pl->d_meta, meta.data(), meta.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
        if (pl->n_cap_items) items.insert(items.end(), cap_items.begin(), cap_items.end());
        if (!items.empty()) {
            H2W_HIP(hipMalloc((void **)&pl->d_items, items.size() * sizeof(LoadItem)));
            H2W_HIP(hipMemcpy(pl->d_items, items.data(), items.size() * sizeof(LoadItem), hipMemcpyHostToDevice));
        }
        {   // PoseidonBN254 tables: canonical and R-premultiplied, per plan (two plans with different tables never share state)
            std::vector<fr_t> tab(BK_ALL); bn_table_build(*consts, pl->P, tab.data());
            H2W_HIP(hipMalloc((void **)&pl->d_bn_tab, tab.size() * sizeof(fr_t)));
            H2W_HIP(hipMemcpy(pl->d_bn_tab, tab.data(), tab.size() * sizeof(fr_t), hipMemcpyHostToDevice));
            std::vector<uint32_t> tab9((size_t)BK9_N * BK9_W); bn_table9_build(tab.data(), tab9.data());      // the values pass' limb-form copy
            H2W_HIP(hipMalloc((void **)&pl->d_bn_tab9, tab9.size() * sizeof(uint32_t)));
            H2W_HIP(hipMemcpy(pl->d_bn_tab9, tab9.data(), tab9.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            rf::RowConst rk; rf::rowconst_init(rk, pl->P);                                      // the row-cooperative values pass' constants (rowfr.h)
            H2W_HIP(hipMalloc((void **)&pl->d_rowk, sizeof(rk)));
            H2W_HIP(hipMemcpy(pl->d_rowk, &rk, sizeof(rk), hipMemcpyHostToDevice));
        }
        {   // the FRI gadgets' shape constants (valbackend.h FriTab): per-call host work in the reference, a table here
            FriTab ft; fri_tab_build(ft, pl->shape.degree_bits + pl->shape.rate_bits);
            H2W_HIP(hipMalloc((void **)&pl->d_fri, sizeof(FriTab)));
            H2W_HIP(hipMemcpy(pl->d_fri, &ft, sizeof(FriTab), hipMemcpyHostToDevice));
        }
        H2W_HIP(hipMalloc((void **)&pl->d_st, sizeof(StrandTable)));
        H2W_HIP(hipMemcpy(pl->d_st, &pl->st, sizeof(StrandTable), hipMemcpyHostToDevice));
        {   // the constants, and behind them the derived tables of the values phase (coop.h glp_aux_tables)
            std::vector<uint64_t> aux(GLP_AUX_WORDS); glp_aux_tables(*consts, aux.data());
            H2W_HIP(hipMalloc((void **)&pl->d_consts, sizeof(h2w_poseidon_consts_t) + aux.size() * sizeof(uint64_t)));
            H2W_HIP(hipMemcpy(pl->d_consts, consts, sizeof(h2w_poseidon_consts_t), hipMemcpyHostToDevice));
            H2W_HIP(hipMemcpy(pl->d_consts + 1, aux.data(), aux.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
        }
        std::vector<uint16_t> nc(T_MAX, 0); for (size_t i = 0; i < pl->tt.info.size(); i++) nc[i] = pl->tt.info[i].ncells;
        H2W_HIP(hipMalloc((void **)&pl->d_ncells, nc.size() * sizeof(uint16_t)));
        H2W_HIP(hipMemcpy(pl->d_ncells, nc.data(), nc.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        H2W_HIP(hipMalloc((void **)&pl->d_inv, inv.size() * sizeof(fr_t)));
        H2W_HIP(hipMemcpy(pl->d_inv, inv.data(), inv.size() * sizeof(fr_t), hipMemcpyHostToDevice));
        for (int r = 0; r < h2w_plan::EV_RING; r++) for (int i = 0; i < h2w_plan::N_EV; i++) H2W_HIP(hipEventCreate(&pl->evr[r][i]));
        pl->ev_ready = true;
        return 0;
    };
    if (up() != 0) { h2w_plan_free(pl); return nullptr; }      // (frees whatever the partial upload allocated)
    return pl;
}
void h2w_plan_free(h2w_plan *p) {
    if (!p) return;
    DeviceGuard dg(p->device);
    if (p->traced) traced_free(p);
    if (p->d_meta) (void)hipFree(p->d_meta);
    if (p->d_items) (void)hipFree(p->d_items);
    if (p->d_bn_tab) (void)hipFree(p->d_bn_tab);
    if (p->d_bn_tab9) (void)hipFree(p->d_bn_tab9);
    if (p->d_rowk) (void)hipFree(p->d_rowk);
    if (p->d_fri) (void)hipFree(p->d_fri);
    if (p->d_consts) (void)hipFree(p->d_consts);
    if (p->d_st) (void)hipFree(p->d_st);
    if (p->d_ncells) (void)hipFree(p->d_ncells);
    if (p->d_inv) (void)hipFree(p->d_inv);
    if (p->d_lookup_cells) (void)hipFree(p->d_lookup_cells);
    if (p->d_sel_bits) (void)hipFree(p->d_sel_bits);
    if (p->d_col_tab) (void)hipFree(p->d_col_tab);
    if (p->ev_ready) for (int r = 0; r < h2w_plan::EV_RING; r++) for (int i = 0; i < h2w_plan::N_EV; i++) (void)hipEventDestroy(p->evr[r][i]);
    for (int i = 0; i < p->n_side; i++) (void)hipStreamDestroy(p->side[i]);
    p->dt.free();
    delete p;
}
uint64_t h2w_plan_num_cells(const h2w_plan *p) { return p ? p->ncells : 0; }
uint64_t h2w_plan_proof_words(const h2w_plan *p) { return p ? p->pl.total : 0; }
uint64_t h2w_plan_num_records(const h2w_plan *p) { return p ? p->nrec : 0; }
uint64_t h2w_plan_num_record_cells(const h2w_plan *p) { return p ? p->rec_cells : 0; }
int h2w_plan_strand_layout(const h2w_plan *p, uint64_t out[4]) {
    if (!p || !out) { set_error("h2w_plan_strand_layout: null argument"); return -1; }
    out[0] = p->st.pro_ncell; out[1] = p->st.q_ncell[0]; out[2] = p->shape.num_queries > 1 ? p->st.q_ncell[1] : p->st.q_ncell[0]; out[3] = p->ncells;
    return 0;
}
uint64_t h2w_plan_num_chain_cells(const h2w_plan *p) {      // cells of the Merkle strands: what k_merkle_bn_fused writes per proof (hash_mode 1)
    if (!p) return 0;
    uint64_t n = 0;
    for (int k = 0; k < MK_KINDS; k++) n += p->st.mk_ncell[0][k] + (uint64_t)(p->shape.num_queries - 1) * p->st.mk_ncell[1][k];
    return n;
}
struct ShardSpec { int rank = 0, world = 1, compact = 0; };
static uint64_t own_count(uint64_t total, int rank, int world) { return total > (uint64_t)rank ? (total - (uint64_t)rank + (uint64_t)world - 1) / (uint64_t)world : 0; }
static uint32_t unit_slot_of(const h2w_plan *p) { return (uint32_t)(p->st.q_nunit[0] > p->st.q_nunit[1] ? p->st.q_nunit[0] : p->st.q_nunit[1]); }
// Per-proof pieces first (their offsets do not depend on the sharding: h2w_plan_status finds the status words whatever call filled them), then the
// PoseidonBN254 unit buffers, which hold the units of THIS RANK's (proof, query) units only.
struct WsLayout { size_t recs, cbs, status, lflag, units, sbox, sbox9, glp, ctr, total; };
static WsLayout ws_layout(const h2w_plan *p, uint64_t n, const ShardSpec &sh = ShardSpec()) {
    WsLayout w; size_t o = 0;
    w.recs = o; o += align_up((size_t)n * p->nrec * sizeof(rec_t), 256);
    w.cbs = o; o += align_up((size_t)n * sizeof(DevCB), 256);
    w.status = o; o += align_up((size_t)n * sizeof(uint32_t), 256);
    w.lflag = o; o += align_up((size_t)n * sizeof(uint32_t), 256);      // per proof: a word outside its field's range (k_prologue_load)
    w.glp = o; o += align_up((size_t)n * p->st.total_glp * GLP_LIST_WORDS * sizeof(uint64_t), 256);     // listed Goldilocks-Poseidon permutations
    w.ctr = o; o += align_up((size_t)n * sizeof(uint32_t), 256);                                       // expansion kernel's per-proof tile counters
    const size_t own_units = (size_t)own_count(n * (uint64_t)p->shape.num_queries, sh.rank, sh.world) * unit_slot_of(p);
    w.units = o; o += align_up(own_units * 4 * sizeof(fr_t), 256);                                     // PoseidonBN254 unit states (values phase -> emission)
    w.sbox = o; o += align_up(own_units * BN_PARTIAL_ROUNDS * 3 * sizeof(fr_t), 256);                  // ... and the S-box values of their partial rounds
    w.sbox9 = o; o += align_up(own_units * BN_PARTIAL_ROUNDS * 3 * rf::SBX9_W * sizeof(uint32_t), 256);  // ... in the limb form the row-cooperative values pass leaves them in
    w.total = o;
    return w;
}
uint64_t h2w_plan_workspace_bytes(const h2w_plan *p, uint64_t n_proofs) { return !p ? 0 : p->traced ? traced_workspace_bytes(p, n_proofs) : ws_layout(p, n_proofs).total; }
uint64_t h2w_plan_shard_workspace_bytes(const h2w_plan *p, uint64_t n_proofs, int rank, int world) {
    if (!p || world < 1 || rank < 0 || rank >= world) return 0;
    ShardSpec sh; sh.rank = rank; sh.world = world;
    return ws_layout(p, n_proofs, sh).total;
}
static int run_batch(h2w_plan *p, const uint64_t *proofs_dev, uint64_t n_proofs, void *advice_dev, void *workspace_dev, void *stream_, void *emit_stream_, ColMap cm, uint64_t cell_stride, ShardSpec sh = ShardSpec());
int h2w_fri_witness_batch(h2w_plan *p, const uint64_t *proofs_dev, uint64_t n_proofs, void *advice_dev, void *workspace_dev, void *stream_) {
    ColMap flat; flat.starts = nullptr; flat.ncols = 0; flat.k = 0;
    return run_batch(p, proofs_dev, n_proofs, advice_dev, workspace_dev, stream_, stream_, flat, p ? p->ncells : 0);
}
int h2w_fri_witness_batch2(h2w_plan *p, const uint64_t *proofs_dev, uint64_t n_proofs, void *advice_dev, void *workspace_dev, void *stream_, void *emit_stream_) {
    ColMap flat; flat.starts = nullptr; flat.ncols = 0; flat.k = 0;
    return run_batch(p, proofs_dev, n_proofs, advice_dev, workspace_dev, stream_, emit_stream_, flat, p ? p->ncells : 0);
}
// column-major emission: boundary cells repeated in the previous column, unused rows zeroed
__global__ void k_columns_fixup(ulonglong2 *cols, const uint64_t *lens, uint32_t ncols, uint32_t k) {
    const uint64_t rows2 = (uint64_t)2 << k; const uint32_t p = blockIdx.z, c = blockIdx.y;
    ulonglong2 *col = cols + ((uint64_t)p * ncols + c) * rows2; const uint64_t len2 = lens[c] * 2;
    if (c + 1 < ncols && blockIdx.x == 0 && threadIdx.x < 2) col[len2 - 2 + threadIdx.x] = col[rows2 + threadIdx.x];       // (c, len-1) <- (c+1, 0)
    for (uint64_t h = len2 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < rows2; h += (uint64_t)gridDim.x * blockDim.x) col[h] = make_ulonglong2(0, 0);
}
int h2w_fri_witness_batch_columns(h2w_plan *p, const uint64_t *proofs_dev, uint64_t n_proofs, const uint64_t *break_points, uint64_t n_bp, int k,
                                  void *columns_dev, void *workspace_dev, void *stream_) {
    if (!p || (!break_points && n_bp) || !columns_dev) { set_error("h2w_fri_witness_batch_columns: null argument"); return -1; }
    if (k < 12 || k > 34 || n_bp + 1 > 4096) { set_error("h2w_fri_witness_batch_columns: k must be in [12, 34] and at most 4096 columns"); return -1; }
    if (p->device < 0) { set_error("h2w_fri_witness_batch_columns: no HIP device — the hot path only runs on the GPU (no CPU fallback)"); return -1; }
    const uint64_t ncols = n_bp + 1; std::vector<uint64_t> h(2 * ncols); uint64_t start = 0;
    for (uint64_t c = 0; c < ncols; c++) {
        const uint64_t len = c < n_bp ? break_points[c] + 1 : p->ncells - start;
        if (start + len > p->ncells || len > ((uint64_t)1 << k) || len < 2) { set_error("h2w_fri_witness_batch_columns: break points do not fit the stream"); return -1; }
        h[c] = start; h[ncols + c] = len; start += len - (c < n_bp ? 1 : 0);
    }
    DeviceGuard dg(p->device);
    hipStream_t stream = (hipStream_t)stream_;
    if (h != p->h_col_tab || k != p->col_k) {      // (re)upload the column table; plans are single-threaded handles (include/h2w.h)
        H2W_HIP(hipDeviceSynchronize());             // a previous call on another stream may still read the old table
        if (p->d_col_tab) { (void)hipFree(p->d_col_tab); p->d_col_tab = nullptr; }
        H2W_HIP(hipMalloc((void **)&p->d_col_tab, h.size() * sizeof(uint64_t)));
        H2W_HIP(hipMemcpy(p->d_col_tab, h.data(), h.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
        p->h_col_tab = h; p->col_k = k;
    }
    ColMap cm; cm.starts = p->d_col_tab; cm.ncols = (uint32_t)ncols; cm.k = (uint32_t)k;
    if (run_batch(p, proofs_dev, n_proofs, columns_dev, workspace_dev, stream_, stream_, cm, ncols << k) != 0) return -1;
    if (n_proofs) {
        if (n_proofs > 65535) { set_error("h2w_fri_witness_batch_columns: too many proofs per call"); return -1; }
        hipLaunchKernelGGL(k_columns_fixup, dim3(64, (unsigned)ncols, (unsigned)n_proofs), dim3(256), 0, stream, (ulonglong2 *)columns_dev, p->d_col_tab + ncols, (uint32_t)ncols, (uint32_t)k);
        H2W_HIP(hipGetLastError());
    }
    return 0;
}
// (proof, query) sharding: this rank LAUNCHES only what it owns - the prologue block of the proofs p % world == rank and the
// query blocks of the units (proof * num_queries + query) % world == rank (every rank runs every prologue's values: it needs the
// challenges) - at their global offsets in advice_dev[n_proofs][num_cells]; the other blocks are left untouched.
int h2w_fri_witness_batch_shard(h2w_plan *p, const uint64_t *proofs_dev, uint64_t n_proofs, void *advice_dev, void *workspace_dev, void *stream_, int rank, int world) {
    if (world < 1 || rank < 0 || rank >= world) { set_error("h2w_fri_witness_batch_shard: bad rank / world"); return -1; }
    ColMap flat; flat.starts = nullptr; flat.ncols = 0; flat.k = 0;
    ShardSpec sh; sh.rank = rank; sh.world = world;
    return run_batch(p, proofs_dev, n_proofs, advice_dev, workspace_dev, stream_, stream_, flat, p ? p->ncells : 0, sh);
}
// The same blocks packed into a buffer of h2w_plan_shard_cells cells: the rank's owned blocks back to back in (proof, block) order.
static uint64_t shard_q_slot(const h2w_plan *p) { return p->st.q_ncell[0] > p->st.q_ncell[1] ? p->st.q_ncell[0] : p->st.q_ncell[1]; }
uint64_t h2w_plan_shard_cells(const h2w_plan *p, uint64_t n_proofs, int rank, int world) {
    if (!p || world < 1 || rank < 0 || rank >= world) return 0;
    return own_count(n_proofs, rank, world) * p->st.pro_ncell + own_count(n_proofs * (uint64_t)p->shape.num_queries, rank, world) * shard_q_slot(p);
}
int h2w_plan_shard_block(const h2w_plan *p, int rank, int world, uint64_t proof, int query, uint64_t *local_cell, uint64_t *n_cells, uint64_t *global_cell) {
    if (!p || world < 1 || rank < 0 || rank >= world || query >= p->shape.num_queries) { set_error("h2w_plan_shard_block: bad argument"); return -1; }
    const uint64_t W = (uint64_t)world, r = (uint64_t)rank, nq = (uint64_t)p->shape.num_queries, u0 = proof * nq;
    const bool owned = query < 0 ? proof % W == r : (u0 + (uint64_t)query) % W == r;
    if (!owned) return 1;
    const uint64_t pro_before = (proof + W - 1 - r) / W, units_before = (u0 + W - 1 - r) / W;
    uint64_t local = pro_before * p->st.pro_ncell + units_before * shard_q_slot(p), n = p->st.pro_ncell, g = 0;
    if (query >= 0) {
        if (proof % W == r) local += p->st.pro_ncell;
        local += ((u0 + (uint64_t)query + W - 1 - r) / W - units_before) * shard_q_slot(p);
        n = p->st.q_ncell[query == 0 ? 0 : 1]; g = strand_q_cell(p->st, query);
    }
    if (local_cell) *local_cell = local; if (n_cells) *n_cells = n; if (global_cell) *global_cell = g;
    return 0;
}
int h2w_fri_witness_batch_shard_compact(h2w_plan *p, const uint64_t *proofs_dev, uint64_t n_proofs, void *shard_advice_dev, void *workspace_dev, void *stream_, int rank, int world) {
    if (world < 1 || rank < 0 || rank >= world) { set_error("h2w_fri_witness_batch_shard_compact: bad rank / world"); return -1; }
    if (p && !(p->shape.lookup_bits == 21 || p->shape.lookup_bits == 13 || p->shape.lookup_bits == 8)) { set_error("h2w_fri_witness_batch_shard_compact: lookup_bits must be 21, 13 or 8 (the packed layout is written by expand_fast)"); return -1; }
    ColMap flat; flat.starts = nullptr; flat.ncols = 0; flat.k = 0;
    ShardSpec sh; sh.rank = rank; sh.world = world; sh.compact = 1;
    return run_batch(p, proofs_dev, n_proofs, shard_advice_dev, workspace_dev, stream_, stream_, flat, 0, sh);
}
static void fill_expand_shard(const h2w_plan *p, ExpandArgs &E, const ShardSpec &sh) {
    E.shard_rank = (uint32_t)sh.rank; E.shard_world = (uint32_t)sh.world; E.shard_compact = (uint32_t)sh.compact; E.nq = (uint32_t)p->shape.num_queries;
    E.pro_nrec = p->st.pro_nrec; E.q_rec0_first = p->st.q_rec0[0]; E.q_rec0_rest = p->st.q_rec0[1]; E.q_nrec_first = p->st.q_nrec[0]; E.q_nrec_rest = p->st.q_nrec[1] ? p->st.q_nrec[1] : 1;
    if (p->shape.num_queries == 1) E.q_rec0_rest = ~0ull;
    E.pro_ncell = p->st.pro_ncell; E.q_cell0_first = p->st.q_cell0[0]; E.q_cell0_rest = p->st.q_cell0[1]; E.q_ncell_rest = p->st.q_ncell[1]; E.q_slot = shard_q_slot(p);
}
static int run_batch(h2w_plan *p, const uint64_t *proofs_dev, uint64_t n_proofs, void *advice_dev, void *workspace_dev, void *stream_, void *emit_stream_, ColMap cm, uint64_t cell_stride, ShardSpec sh) {
    if (!p) { set_error("h2w_fri_witness_batch: null plan"); return -1; }
    if (p->device < 0) { set_error("h2w_fri_witness_batch: no HIP device — the hot path only runs on the GPU (no CPU fallback)"); return -1; }
    if (!proofs_dev || !advice_dev || !workspace_dev) { set_error("h2w_fri_witness_batch: null buffer"); return -1; }
    if (n_proofs == 0) return 0;
    if (p->traced) {      // a recorded run (replay.hip): the flat stream or the FlexGate columns, unsharded
        if (sh.world > 1 || emit_stream_ != stream_) { set_error("h2w_fri_witness_batch: a traced plan runs on one stream, unsharded"); return -1; }
        return traced_run(p, proofs_dev, n_proofs, advice_dev, workspace_dev, stream_, cm, cell_stride);
    }
    if (n_proofs * (uint64_t)p->shape.num_queries * (p->st.mk_item0[MK_KINDS] ? p->st.mk_item0[MK_KINDS] : 1) > 0x3fffffffull) { set_error("h2w_fri_witness_batch: batch too large"); return -1; }
    if (n_proofs > 65535) { set_error("h2w_fri_witness_batch: more than 65535 proofs per call (the proof index is a grid dimension of the load and expansion kernels); split the batch"); return -1; }
    DeviceGuard dg(p->device);
    hipStream_t stream = (hipStream_t)stream_, estream = (hipStream_t)emit_stream_;
    const WsLayout wl = ws_layout(p, n_proofs, sh);
    char *ws = (char *)workspace_dev;
    BatchArgs A;
    A.shape = p->shape; A.consts = p->d_consts; A.proofs = proofs_dev; A.proof_words = p->pl.total;
    A.recs = (rec_t *)(ws + wl.recs); A.rec_stride = p->nrec; A.out = (fr_t *)advice_dev; A.cell_stride = cell_stride; A.cm = cm;
    A.cbs = (DevCB *)(ws + wl.cbs); A.status = (uint32_t *)(ws + wl.status);
    A.unit_state = (fr_t *)(ws + wl.units); A.unit_sbox = (fr_t *)(ws + wl.sbox); A.unit_sbox9 = (uint32_t *)(ws + wl.sbox9); A.rowk = p->d_rowk; A.glp_list = (uint64_t *)(ws + wl.glp); A.glp_small_mds = p->small_mds ? 1 : 0;
    A.bn_tab = p->d_bn_tab; A.bn_tab9 = p->d_bn_tab9; A.fri = p->d_fri;
    A.load_items = p->d_items; A.n_load_items = p->n_items; A.n_cap_items = p->n_cap_items; A.load_nrec = p->load_nrec; A.load_ncell = p->load_ncell; A.load_flag = (uint32_t *)(ws + wl.lflag);
    A.ncells = p->d_ncells; A.inv_pos = p->d_inv; A.inv_neg = p->d_inv + INV_TAB; A.st = p->d_st; A.P = p->P; A.nproofs = (int)n_proofs;
    A.sh.rank = sh.rank; A.sh.world = sh.world; A.sh.compact = sh.compact; A.sh.q_slot = shard_q_slot(p); A.sh.unit_slot = unit_slot_of(p);
    A.sh.n_own_units = (uint32_t)own_count(n_proofs * (uint64_t)p->shape.num_queries, sh.rank, sh.world);
    A.sh.n_own_proofs = (uint32_t)own_count(n_proofs, sh.rank, sh.world);
    hipStream_t cstream = stream;      // chain kernels' stream
    if (p->fork_chains) {
        int k = 0; while (k < p->n_side && p->side_of[k] != stream) k++;
        if (k == p->n_side && p->n_side < h2w_plan::N_SIDE) {
            int lo_pri = 0, hi_pri = 0; (void)hipDeviceGetStreamPriorityRange(&lo_pri, &hi_pri);      // the chain is the longest dependent piece of a launch: let its blocks be placed first
            H2W_HIP(hipStreamCreateWithPriority(&p->side[k], hipStreamNonBlocking, hi_pri)); p->side_of[k] = stream; p->n_side++;
        }
        if (k < p->n_side) cstream = p->side[k];      // (more caller streams than side streams: the extra ones do not fork)
    }
    hipEvent_t *prev_ev = (p->n_batches && p->ev_recorded) ? p->evr[(p->n_batches - 1) % h2w_plan::EV_RING] : nullptr;
    hipEvent_t *ev = p->evr[p->n_batches % h2w_plan::EV_RING];
    const unsigned nunits = A.sh.n_own_units;
    const unsigned nkinds = (unsigned)(p->d.n_oracles + p->d.n_steps);
    bool forked = false;
    auto body = [&]() -> int {
        H2W_HIP(hipEventRecord(ev[0], stream));
        // 1. prologue strands, values: one wavefront per proof (witness load, Fiat-Shamir sponge, PoW, reduced openings) -> challenge blocks.  FIRST: everything
        //    else of the launch waits for the challenges, nothing for the load cells
        launch_prologue_values(A, stream);
        H2W_HIP(hipEventRecord(ev[9], stream));
        // 0. witness load + cap-hash limb decompositions: one lane per (proof, item).  Cells and a flag word only (nobody's input): behind the prologue, beside the
        //    Merkle chains that fork off at this point
        H2W_HIP(hipMemsetAsync(A.load_flag, 0, n_proofs * sizeof(uint32_t), stream));
        if (p->n_items + p->n_cap_items) {
            const dim3 lgrid((p->n_items + p->n_cap_items + 255) / 256, (unsigned)n_proofs);
            if (cm.starts) hipLaunchKernelGGL(k_prologue_load<true>, lgrid, dim3(256), 0, stream, A); else hipLaunchKernelGGL(k_prologue_load<false>, lgrid, dim3(256), 0, stream, A);
        }
        // 2. the records of the listed Goldilocks-Poseidon permutations: with PoseidonBN254 caps the prologues' (now); with Goldilocks caps
        //    together with the Merkle strands' (below)
        const unsigned n_pro_perms = A.sh.n_own_proofs * p->st.pro_nglp, n_mk_perms = nunits * p->st.q_nglp;
        auto glp_emit = [&](unsigned n) -> int {
            H2W_HIP(hipEventRecord(ev[11], stream));
            if (n) { if (cm.starts) hipLaunchKernelGGL(k_glp_emit<true>, dim3(n), dim3(64), 0, stream, A); else hipLaunchKernelGGL(k_glp_emit<false>, dim3(n), dim3(64), 0, stream, A); }
            H2W_HIP(hipEventRecord(ev[12], stream));
            return 0;
        };
        if (p->shape.hash_mode == 1) { if (glp_emit(n_pro_perms) != 0) return -1; }
        H2W_HIP(hipEventRecord(ev[1], stream));
        // 3. PoseidonBN254 Merkle chains (hash_mode 1): values, then one quad per permutation unit.  They write their cells themselves and no
        //    block records, so nothing but the challenge blocks orders them against the other kernels of the batch: they run on a side
        //    stream of the plan and rejoin at the end of the call.
        if (p->shape.hash_mode == 1) {
            if (cstream != stream) { H2W_HIP(hipStreamWaitEvent(cstream, ev[9], 0)); forked = true; }
            H2W_HIP(hipEventRecord(ev[4], cstream));
            const dim3 sgrid((nunits * 4 + QUAD_BLOCK - 1) / QUAD_BLOCK, nkinds);
            // one pass or two (include/h2w.h H2W_OPT_CHAIN_PASSES): a launch whose paths do not fill the chip is bound by the depth of a path - split it
            const int passes = p->chain_passes ? p->chain_passes : (nunits <= 512 ? 2 : 1);
            p->passes_of[p->n_batches % h2w_plan::EV_RING] = passes;
            if (nunits && passes == 1) { if (cm.starts) hipLaunchKernelGGL(k_merkle_bn_fused<true>, sgrid, dim3(QUAD_BLOCK), 0, cstream, A); else hipLaunchKernelGGL(k_merkle_bn_fused<false>, sgrid, dim3(QUAD_BLOCK), 0, cstream, A); }
            if (nunits && passes != 1) {
                // the values of the paths: one wavefront per path (rowperm.h: a path's 18 permutations in 1.3 ms instead of 2.6) for a launch of a few proofs - the form for
                // one proof's latency; four lanes per path (coop.h bn_values: a sixth of the instructions per path) beyond that: alone on the chip the wavefront form
                // wins up to ~2,500 paths (profiles/r04_values_forms.jsonl), but a caller with several launches in flight pays for its instructions (cfg 1 pipelined,
                // 1,024 paths per launch: 160 G cells/s against 175-190), so the default switches early and H2W_OPT_VALUES_FORM = 2 is the caller's to set
                const bool rows = p->values_form ? p->values_form == 2 : (uint64_t)nunits * nkinds <= 784;
                const uint32_t upq = unit_slot_of(p);
                const uint64_t nval = (uint64_t)nunits * upq * (BN_PARTIAL_ROUNDS * 3);
                if (rows) {
                    launch_merkle_bn_values_row(A, nkinds, cstream);
                    if (nval) hipLaunchKernelGGL(k_sbox_canon9, dim3((unsigned)((nval + 255) / 256)), dim3(256), 0, cstream, A, upq);
                } else {
                    launch_merkle_bn_values(A, sgrid, cstream);
                    if (nval) hipLaunchKernelGGL(k_sbox_canon, dim3((unsigned)((nval + 255) / 256)), dim3(256), 0, cstream, A, upq);
                }
            }
            H2W_HIP(hipEventRecord(ev[10], cstream));
            const unsigned long long items = passes == 1 ? 0ull : (unsigned long long)((nunits + 15u) & ~15u) * p->st.mk_item0[MK_KINDS];
            if (items) {
                const dim3 egrid((unsigned)((items * 4 + QUAD_BLOCK - 1) / QUAD_BLOCK));
                launch_merkle_bn_emit(A, egrid, cstream);
            }
            H2W_HIP(hipEventRecord(ev[5], cstream));
        }
        // 4. query glue strands (FriChip::verify_query_round minus its Merkle proofs): one lane per owned (proof, query);
        //    Goldilocks-Poseidon Merkle strands (hash_mode 0), values: one cooperating wavefront per (proof, query, tree)
        //    The two read the challenge blocks and write disjoint records (and list entries): with Goldilocks caps the Merkle strands run on the
        //    plan's side stream beside the glue strands and rejoin before the permutations' records are emitted.
        H2W_HIP(hipEventRecord(ev[7], stream));
        if (p->shape.hash_mode == 0) {
            if (cstream != stream) { H2W_HIP(hipStreamWaitEvent(cstream, ev[9], 0)); forked = true; }
            H2W_HIP(hipEventRecord(ev[4], cstream));
            if (nunits) launch_merkle_gl_values(A, nunits, nkinds, cstream);
            H2W_HIP(hipEventRecord(ev[10], cstream)); H2W_HIP(hipEventRecord(ev[5], cstream));
        }
        if (nunits) launch_glue_strands(A, stream);
        if (p->shape.hash_mode == 0) {
            if (forked) { H2W_HIP(hipStreamWaitEvent(stream, ev[5], 0)); forked = false; }
            if (glp_emit(n_pro_perms + n_mk_perms) != 0) return -1;
        }
        H2W_HIP(hipEventRecord(ev[2], stream));
        // 5. expansion of the block records (HBM-write-bound)
        ExpandArgs E;
        E.meta = p->d_meta; E.recs = A.recs; E.nrec = p->nrec; E.rec_stride = p->nrec; E.out = A.out; E.cell_stride = cell_stride; E.pool = nullptr; E.cm = cm;
        fill_expand_shard(p, E, sh);
        p->dt.fill(E);
        E.tile_ctr = (uint32_t *)(ws + wl.ctr);
        E.roam_per_cu = p->shape.hash_mode == 0 ? 2 : 1;      // (profiles/r02_expand_grid.txt)
        int gx = (int)(2048 / (n_proofs < 2048 ? n_proofs : 2048)); if (gx < 8) gx = 8;
        if (estream != stream) H2W_HIP(hipStreamWaitEvent(estream, ev[2], 0));    // value strands done -> expansion on the emit stream
        H2W_HIP(hipMemsetAsync(E.tile_ctr, 0, n_proofs * sizeof(uint32_t), estream));
        // One expansion kernel at a time over all the streams a plan is driven on: its blocks are persistent and hold their CUs until the
        // launch is written, so two of them side by side keep the latency-bound strands of the other launches in flight off the chip
        // (measured: +8..12 % for the whole job with PoseidonBN254 caps, profiles/r02_sweep5.txt; +5 % with Goldilocks caps since that kernel
        // runs on a grid of resident blocks there and reaches its rate alone, profiles/r02_expand_grid.txt).
        if (p->serial_expand != 0 && prev_ev) H2W_HIP(hipStreamWaitEvent(estream, prev_ev[3], 0));
        H2W_HIP(hipEventRecord(ev[8], estream));
        if (launch_expand(E, n_proofs, gx, estream) != 0) return -1;
        H2W_HIP(hipEventRecord(ev[3], estream));
        if (estream != stream) H2W_HIP(hipStreamWaitEvent(stream, ev[3], 0));    // the caller's stream completes when the advice is complete
        if (forked) { H2W_HIP(hipStreamWaitEvent(stream, ev[5], 0)); forked = false; }
        H2W_HIP(hipEventRecord(ev[6], stream));
        H2W_HIP(hipGetLastError());
        return 0;
    };
    if (body() != 0) {
        // a failed enqueue after the fork: the chain kernels may still be writing advice / workspace on the side stream and nothing orders the
        // caller's stream behind them any more - wait for them here, so that the caller may release the buffers when it sees the error
        if (forked) (void)hipStreamSynchronize(cstream);
        return -1;
    }
    p->ev = ev; p->n_batches++; p->ev_recorded = true;
    return 0;
}
// Re-expands the block records a previous h2w_fri_witness_batch* call left in `workspace_dev` (same n_proofs) into advice_dev: the
// expansion kernel alone, for measuring its streaming rate (bench.py roofline).  Cells written by the value kernels are not touched.
int h2w_fri_expand_records(h2w_plan *p, uint64_t n_proofs, void *advice_dev, void *workspace_dev, void *stream_) {
    if (!p || p->device < 0) { set_error("h2w_fri_expand_records: no HIP device"); return -1; }
    if (p->traced) { set_error("h2w_fri_expand_records: not for traced plans"); return -1; }
    if (!advice_dev || !workspace_dev) { set_error("h2w_fri_expand_records: null buffer"); return -1; }
    if (n_proofs == 0) return 0;
    DeviceGuard dg(p->device);
    hipStream_t stream = (hipStream_t)stream_;
    const WsLayout wl = ws_layout(p, n_proofs);
    char *ws = (char *)workspace_dev;
    ExpandArgs E;
    E.meta = p->d_meta; E.recs = (rec_t *)(ws + wl.recs); E.nrec = p->nrec; E.rec_stride = p->nrec; E.out = (fr_t *)advice_dev; E.cell_stride = p->ncells; E.pool = nullptr;
    E.cm = ColMap{nullptr, 0, 0};
    fill_expand_shard(p, E, ShardSpec());
    p->dt.fill(E);
    E.tile_ctr = (uint32_t *)(ws + wl.ctr);
    E.roam_per_cu = p->shape.hash_mode == 0 ? 2 : 1;      // (profiles/r02_expand_grid.txt)
    H2W_HIP(hipMemsetAsync(E.tile_ctr, 0, n_proofs * sizeof(uint32_t), stream));
    int gx = (int)(2048 / (n_proofs < 2048 ? n_proofs : 2048)); if (gx < 8) gx = 8;
    if (launch_expand(E, n_proofs, gx, stream) != 0) return -1;
    H2W_HIP(hipGetLastError());
    return 0;
}
// ---- keygen-side metadata of the cell stream: static per shape, computed by a second host replay
int h2w_plan_metadata(h2w_plan *pl) {
    if (!pl) { set_error("h2w_plan_metadata: null plan"); return -1; }
    if (pl->traced) { set_error("h2w_plan_metadata: a traced plan has no shape to replay on the host (the tracing context recorded the keygen lists: h2w_ctx_* with witness_gen_only = 0)"); return -1; }
    if (pl->meta_ready) return 0;
    if (pl->shape.lookup_bits >= 48) { set_error("h2w_plan_metadata: lookup_bits >= 48 makes a single-limb range check look up its SOURCE cell; not tracked"); return -1; }
    std::vector<fr_t> inv(2 * INV_TAB, fr_zero());
    for (int k2 = 1; k2 < INV_TAB; k2++) { inv[k2] = fr_inv(fr_from_u64((uint64_t)k2), pl->P); inv[INV_TAB + k2] = fr_neg(inv[k2]); }
    std::vector<uint64_t> zero_proof(pl->pl.total, 0), unit_cell; std::vector<LoadItem> items; StrandTable st; memset(&st, 0, sizeof(st));
    pl->sel_bits.assign((size_t)(pl->ncells + 7) / 8, 0); pl->lk_bits.assign((size_t)(pl->ncells + 7) / 8, 0);
    PlanSink sink; sink.meta = nullptr; sink.tt = &pl->tt; sink.st = &st; sink.unit_cell = &unit_cell; sink.items = &items;
    sink.sel_bits = &pl->sel_bits; sink.lk_bits = &pl->lk_bits;
    ValCfg cfg; cfg.proof = zero_proof.data(); cfg.mode = pl->shape.hash_mode; cfg.L = pl->shape.lookup_bits; cfg.P = pl->P;
    cfg.inv_pos = inv.data(); cfg.inv_neg = inv.data() + INV_TAB; cfg.st = nullptr; cfg.split = false; cfg.split_bn = false; cfg.load_items = nullptr; cfg.n_load_items = 0; cfg.load_nrec = cfg.load_ncell = 0;
    ValBackend<PlanSink> be(sink, cfg, false);
    Verifier<ValBackend<PlanSink>> V(be, pl->shape, &pl->h_consts);
    ChallengeBlock<ValBackend<PlanSink>> *cb = new ChallengeBlock<ValBackend<PlanSink>>();
    V.run_all(*cb);
    delete cb;
    if (sink.cell_off != pl->ncells) { set_error("h2w_plan_metadata: internal: replay length mismatch"); return -1; }
    pl->sel_bits.resize((size_t)(pl->ncells + 7) / 8); pl->lk_bits.resize((size_t)(pl->ncells + 7) / 8);
    pl->n_gates = pl->n_lookups = 0;
    for (uint8_t b : pl->sel_bits) pl->n_gates += (uint64_t)__builtin_popcount(b);
    for (uint8_t b : pl->lk_bits) pl->n_lookups += (uint64_t)__builtin_popcount(b);
    pl->meta_ready = true;
    return 0;
}
uint64_t h2w_plan_num_gates(h2w_plan *p) { return p && h2w_plan_metadata(p) == 0 ? p->n_gates : 0; }
uint64_t h2w_plan_num_lookups(h2w_plan *p) { return p && h2w_plan_metadata(p) == 0 ? p->n_lookups : 0; }
int h2w_plan_selectors(h2w_plan *p, uint8_t *bitmap) {
    if (!p || !bitmap) { set_error("h2w_plan_selectors: null argument"); return -1; }
    if (h2w_plan_metadata(p) != 0) return -1;
    memcpy(bitmap, p->sel_bits.data(), p->sel_bits.size()); return 0;
}
int h2w_plan_lookup_cells(h2w_plan *p, uint64_t *cells) {
    if (!p || !cells) { set_error("h2w_plan_lookup_cells: null argument"); return -1; }
    if (h2w_plan_metadata(p) != 0) return -1;
    uint64_t k2 = 0;
    for (uint64_t i = 0; i < p->ncells; i++) if (p->lk_bits[i / 8] >> (i & 7) & 1) cells[k2++] = i;     // registration order = stream order (RangeChip::range_check)
    return 0;
}
// FlexGate break points (halo2-base assign_with_constraints, ROTATIONS = 4 [R]): walk the stream down a column of
// max_rows = 2^k - unusable_rows; break when a gate would not fit or the column is full; the breaking cell is assigned twice
// (last row of the old column, row 0 of the new one).
int h2w_break_points(const uint8_t *selectors, uint64_t n_cells, int k, int unusable_rows, uint64_t *out, uint64_t cap, uint64_t *n_out) {
    if (!selectors || !n_out || k < 3 || k > 40 || unusable_rows < 0 || ((uint64_t)1 << k) <= (uint64_t)unusable_rows + 4) { set_error("h2w_break_points: bad argument"); return -1; }
    const uint64_t max_rows = ((uint64_t)1 << k) - (uint64_t)unusable_rows; uint64_t row = 0, n = 0;
    for (uint64_t i = 0; i < n_cells; i++) {
        const bool q = selectors[i / 8] >> (i & 7) & 1;
        if ((q && row + 4 > max_rows) || row >= max_rows - 1) { if (out && n < cap) out[n] = row; n++; row = 0; }
        row++;
    }
    *n_out = n;
    if (out && n > cap) { set_error("h2w_break_points: output too small"); return -1; }
    return 0;
}
// advice -> FlexGate columns on the device: columns[p][c][r], c < n_bp + 1, r < 2^k (unassigned rows zero), 32-byte cells
__global__ void k_layout_columns(const ulonglong2 *advice, uint64_t proof_stride, const uint64_t *starts, const uint64_t *lens, uint32_t ncols, uint32_t k, ulonglong2 *out) {
    const uint64_t rows2 = (uint64_t)2 << k;                      // 16-byte halves per column
    const uint32_t p = blockIdx.z, c = blockIdx.y;
    const uint64_t start = starts[c], len2 = lens[c] * 2;
    const ulonglong2 *src = advice + ((uint64_t)p * proof_stride + start) * 2;
    ulonglong2 *dst = out + ((uint64_t)p * ncols + c) * rows2;
    for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < rows2; h += (uint64_t)gridDim.x * blockDim.x)
        dst[h] = h < len2 ? src[h] : make_ulonglong2(0, 0);
}
int h2w_layout_columns(const void *advice_dev, uint64_t n_cells, uint64_t proof_stride_cells, uint64_t n_proofs, const uint64_t *break_points, uint64_t n_bp, int k, void *columns_dev, void *stream_) {
    if (!advice_dev || !columns_dev || (!break_points && n_bp) || k < 3 || k > 34) { set_error("h2w_layout_columns: bad argument"); return -1; }
    if (n_proofs == 0) return 0;
    const uint64_t ncols = n_bp + 1; std::vector<uint64_t> h(2 * ncols); uint64_t start = 0;
    for (uint64_t c = 0; c < ncols; c++) {      // column c holds cells [start, start + len); consecutive columns share their boundary cell
        const uint64_t len = c < n_bp ? break_points[c] + 1 : n_cells - start;
        if (start + len > n_cells || len > ((uint64_t)1 << k)) { set_error("h2w_layout_columns: break points do not fit the stream"); return -1; }
        h[c] = start; h[ncols + c] = len; start += len - (c < n_bp ? 1 : 0);
    }
    if (ncols > 65535 || n_proofs > 65535) { set_error("h2w_layout_columns: too many columns / proofs per call"); return -1; }
    DeviceGuard dg(device_of(advice_dev));
    hipStream_t stream = (hipStream_t)stream_; uint64_t *d = nullptr;
    H2W_HIP(hipMallocAsync((void **)&d, h.size() * sizeof(uint64_t), stream));
    H2W_HIP(hipMemcpyAsync(d, h.data(), h.size() * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
    H2W_HIP(hipStreamSynchronize(stream));      // h is a local
    const unsigned gx = (unsigned)std::min<uint64_t>((((uint64_t)2 << k) + 255) / 256, 4096);
    hipLaunchKernelGGL(k_layout_columns, dim3(gx, (unsigned)ncols, (unsigned)n_proofs), dim3(256), 0, stream, (const ulonglong2 *)advice_dev, proof_stride_cells, d, d + ncols, (uint32_t)ncols, (uint32_t)k, (ulonglong2 *)columns_dev);
    H2W_HIP(hipFreeAsync(d, stream));
    H2W_HIP(hipGetLastError());
    return 0;
}
// lookup advice: the looked-up cells, in registration order, down columns of max_rows rows [R]; out[p][c][r], r < 2^k
__global__ void k_layout_lookup(const ulonglong2 *advice, uint64_t proof_stride, const uint32_t *cells, uint64_t n_lookups, uint64_t max_rows, uint32_t ncols, uint32_t k, ulonglong2 *out) {
    const uint64_t rows = (uint64_t)1 << k, total2 = (uint64_t)ncols * rows * 2; const uint32_t p = blockIdx.y;
    for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < total2; h += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t cellpos = h >> 1, c = cellpos / rows, r = cellpos % rows, j = c * max_rows + r;
        ulonglong2 v = make_ulonglong2(0, 0);
        if (r < max_rows && j < n_lookups) v = advice[((uint64_t)p * proof_stride + cells[j]) * 2 + (h & 1)];
        out[(uint64_t)p * total2 + h] = v;
    }
}
static int ensure_lookup_cells(h2w_plan *p);
int h2w_layout_lookup_columns(h2w_plan *p, const void *advice_dev, uint64_t proof_stride_cells, uint64_t n_proofs, int k, int unusable_rows, void *out_dev, uint64_t *n_cols_out, void *stream_) {
    if (!p || k < 3 || k > 34 || unusable_rows < 0) { set_error("h2w_layout_lookup_columns: bad argument"); return -1; }
    if (h2w_plan_metadata(p) != 0) return -1;
    const uint64_t max_rows = ((uint64_t)1 << k) - (uint64_t)unusable_rows, ncols = (p->n_lookups + max_rows - 1) / max_rows;
    if (n_cols_out) *n_cols_out = ncols;
    if (!out_dev) return 0;                    // size query
    if (!advice_dev) { set_error("h2w_layout_lookup_columns: null advice"); return -1; }
    if (p->device < 0) { set_error("h2w_layout_lookup_columns: no HIP device"); return -1; }
    DeviceGuard dg(p->device);
    hipStream_t stream = (hipStream_t)stream_;
    if (ensure_lookup_cells(p) != 0) return -1;
    if (n_proofs == 0 || ncols == 0) return 0;
    hipLaunchKernelGGL(k_layout_lookup, dim3(4096, (unsigned)n_proofs), dim3(256), 0, stream, (const ulonglong2 *)advice_dev, proof_stride_cells, p->d_lookup_cells, p->n_lookups, max_rows, (uint32_t)ncols, (uint32_t)k, (ulonglong2 *)out_dev);
    H2W_HIP(hipGetLastError());
    return 0;
}
// Device-side constraint check of an advice stream (the MockProver's gate and lookup checks, restated): every vertical gate
// a[i] + a[i+1]*a[i+2] = a[i+3] at a selector-enabled cell i, and every looked-up cell < 2^lookup_bits.  Size-independent: it
// covers every cell of a full-size stream without the CPU oracle.  Copy constraints are not checked (no equality lists yet).
__global__ void k_check_gates(const fr_t *advice, uint64_t proof_stride, uint64_t n_cells, const uint8_t *sel, FrParams P, unsigned long long *bad) {
    const uint32_t p = blockIdx.y; const fr_t *adv = advice + (uint64_t)p * proof_stride; unsigned long long nb = 0;
    for (uint64_t byte = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; byte < (n_cells + 7) / 8; byte += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t m = sel[byte];
        while (m) {
            const int b = __ffs((int)m) - 1; m &= m - 1; const uint64_t i = byte * 8 + (uint64_t)b;
            if (i + 3 >= n_cells) { nb++; continue; }
            const fr_t a = g_load_fr(adv + i), x = g_load_fr(adv + i + 1), y = g_load_fr(adv + i + 2), d = g_load_fr(adv + i + 3);
            if (!fr_eq(fr_add(a, fr_mul(x, y, P)), d)) nb++;
        }
    }
    if (nb) atomicAdd(bad, nb);
}
__global__ void k_check_lookups(const fr_t *advice, uint64_t proof_stride, const uint32_t *cells, uint64_t n_lookups, int lookup_bits, unsigned long long *bad) {
    const uint32_t p = blockIdx.y; const fr_t *adv = advice + (uint64_t)p * proof_stride; unsigned long long nb = 0;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_lookups; j += (uint64_t)gridDim.x * blockDim.x) {
        const fr_t v = g_load_fr(adv + cells[j]);
        if ((v.l[1] | v.l[2] | v.l[3]) != 0 || (v.l[0] >> lookup_bits) != 0) nb++;
    }
    if (nb) atomicAdd(bad + 1, nb);
}
static int ensure_lookup_cells(h2w_plan *p) {
    if (p->d_lookup_cells || !p->n_lookups) return 0;
    if (p->ncells >> 32) { set_error("lookup cells: stream longer than 2^32 cells"); return -1; }
    std::vector<uint32_t> h((size_t)p->n_lookups); uint64_t k2 = 0;
    for (uint64_t i = 0; i < p->ncells; i++) if (p->lk_bits[i / 8] >> (i & 7) & 1) h[k2++] = (uint32_t)i;
    H2W_HIP(hipMalloc((void **)&p->d_lookup_cells, h.size() * sizeof(uint32_t)));
    H2W_HIP(hipMemcpy(p->d_lookup_cells, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    return 0;
}
int h2w_check_constraints(h2w_plan *p, const void *advice_dev, uint64_t proof_stride_cells, uint64_t n_proofs, uint64_t bad_out[2], void *stream_) {
    if (!p || !advice_dev || !bad_out) { set_error("h2w_check_constraints: null argument"); return -1; }
    if (p->device < 0) { set_error("h2w_check_constraints: no HIP device"); return -1; }
    DeviceGuard dg(p->device);
    if (h2w_plan_metadata(p) != 0 || ensure_lookup_cells(p) != 0) return -1;
    bad_out[0] = bad_out[1] = 0;
    if (n_proofs == 0) return 0;
    if (n_proofs > 65535) { set_error("h2w_check_constraints: too many proofs per call"); return -1; }
    hipStream_t stream = (hipStream_t)stream_;
    if (!p->d_sel_bits) {
        H2W_HIP(hipMalloc((void **)&p->d_sel_bits, p->sel_bits.size()));
        H2W_HIP(hipMemcpy(p->d_sel_bits, p->sel_bits.data(), p->sel_bits.size(), hipMemcpyHostToDevice));
    }
    unsigned long long *d_bad = nullptr;
    H2W_HIP(hipMallocAsync((void **)&d_bad, 16, stream));
    unsigned long long h[2] = {0, 0};
    auto run = [&]() -> int {
        H2W_HIP(hipMemsetAsync(d_bad, 0, 16, stream));
        hipLaunchKernelGGL(k_check_gates, dim3(2048, (unsigned)n_proofs), dim3(256), 0, stream, (const fr_t *)advice_dev, proof_stride_cells, p->ncells, p->d_sel_bits, p->P, d_bad);
        if (p->n_lookups) hipLaunchKernelGGL(k_check_lookups, dim3(1024, (unsigned)n_proofs), dim3(256), 0, stream, (const fr_t *)advice_dev, proof_stride_cells, p->d_lookup_cells, p->n_lookups, (int)p->shape.lookup_bits, d_bad);
        H2W_HIP(hipMemcpyAsync(h, d_bad, 16, hipMemcpyDeviceToHost, stream));
        H2W_HIP(hipStreamSynchronize(stream));
        return 0;
    };
    const int rc = run();
    (void)hipFreeAsync(d_bad, stream);
    if (rc != 0) return -1;
    bad_out[0] = h[0]; bad_out[1] = h[1];
    return 0;
}
// copy constraints and constant equalities over device advice streams (the rest of the restated MockProver): the lists are static
// per shape and come from an eager keygen context (h2w_ctx_equalities / h2w_ctx_const_equalities)
__global__ void k_check_equalities(const fr_t *advice, uint64_t proof_stride, const uint64_t *pairs, uint64_t n_pairs, const uint64_t *ccells, const fr_t *cvals, uint64_t n_const, unsigned long long *bad) {
    const uint32_t p = blockIdx.y; const fr_t *adv = advice + (uint64_t)p * proof_stride; unsigned long long b0 = 0, b1 = 0;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_pairs + n_const; j += (uint64_t)gridDim.x * blockDim.x) {
        if (j < n_pairs) { if (!fr_eq(g_load_fr(adv + pairs[2 * j]), g_load_fr(adv + pairs[2 * j + 1]))) b0++; }
        else { const uint64_t t = j - n_pairs; if (!fr_eq(g_load_fr(adv + ccells[t]), g_load_fr(cvals + t))) b1++; }
    }
    if (b0) atomicAdd(bad, b0);
    if (b1) atomicAdd(bad + 1, b1);
}
int h2w_check_equalities(const void *advice_dev, uint64_t n_cells, uint64_t proof_stride_cells, uint64_t n_proofs, const uint64_t *pairs, uint64_t n_pairs,
                         const uint64_t *const_cells, const h2w_fr_t *const_values, uint64_t n_const, uint64_t bad_out[2], void *stream_) {
    if (!advice_dev || !bad_out || (n_pairs && !pairs) || (n_const && (!const_cells || !const_values))) { set_error("h2w_check_equalities: null argument"); return -1; }
    bad_out[0] = bad_out[1] = 0;
    if (n_proofs == 0 || n_pairs + n_const == 0) return 0;
    if (n_proofs > 65535) { set_error("h2w_check_equalities: too many proofs per call"); return -1; }
    for (uint64_t i = 0; i < 2 * n_pairs; i++) if (pairs[i] >= n_cells) { set_error("h2w_check_equalities: equality refers to a cell outside the stream"); return -1; }
    for (uint64_t i = 0; i < n_const; i++) if (const_cells[i] >= n_cells) { set_error("h2w_check_equalities: constant equality refers to a cell outside the stream"); return -1; }
    DeviceGuard dg(device_of(advice_dev));
    hipStream_t stream = (hipStream_t)stream_;
    // one stream-ordered allocation for the three lists and the counters: nothing to leak on an error path, nothing synchronous
    const size_t b_pairs = (n_pairs ? 2 * n_pairs : 1) * 8, b_cc = (n_const ? n_const : 1) * 8, b_cv = (n_const ? n_const : 1) * sizeof(fr_t);
    char *d = nullptr;
    H2W_HIP(hipMallocAsync((void **)&d, b_pairs + b_cc + b_cv + 16, stream));
    uint64_t *d_pairs = (uint64_t *)d, *d_cc = (uint64_t *)(d + b_pairs); fr_t *d_cv = (fr_t *)(d + b_pairs + b_cc); unsigned long long *d_bad = (unsigned long long *)(d + b_pairs + b_cc + b_cv);
    unsigned long long h[2] = {0, 0};
    auto run = [&]() -> int {
        if (n_pairs) H2W_HIP(hipMemcpyAsync(d_pairs, pairs, 2 * n_pairs * 8, hipMemcpyHostToDevice, stream));
        if (n_const) { H2W_HIP(hipMemcpyAsync(d_cc, const_cells, n_const * 8, hipMemcpyHostToDevice, stream)); H2W_HIP(hipMemcpyAsync(d_cv, const_values, n_const * sizeof(fr_t), hipMemcpyHostToDevice, stream)); }
        H2W_HIP(hipMemsetAsync(d_bad, 0, 16, stream));
        hipLaunchKernelGGL(k_check_equalities, dim3(1024, (unsigned)n_proofs), dim3(256), 0, stream, (const fr_t *)advice_dev, proof_stride_cells, d_pairs, n_pairs, d_cc, d_cv, n_const, d_bad);
        H2W_HIP(hipMemcpyAsync(h, d_bad, 16, hipMemcpyDeviceToHost, stream));
        H2W_HIP(hipStreamSynchronize(stream));
        return 0;
    };
    const int rc = run();
    (void)hipFreeAsync(d, stream);
    if (rc != 0) return -1;
    bad_out[0] = h[0]; bad_out[1] = h[1];
    return 0;
}
int h2w_plan_status(h2w_plan *p, const void *workspace_dev, uint64_t n_proofs, uint32_t *host_status, void *stream_) {
    if (!p || !workspace_dev || !host_status) { set_error("h2w_plan_status: null argument"); return -1; }
    WsLayout wl = ws_layout(p, n_proofs);
    if (p->traced) { wl.status = traced_status_offset(p, n_proofs, false); wl.lflag = traced_status_offset(p, n_proofs, true); }
    DeviceGuard dg(p->device);
    hipStream_t stream = (hipStream_t)stream_;
    std::vector<uint32_t> flag((size_t)n_proofs);
    H2W_HIP(hipMemcpyAsync(host_status, (const char *)workspace_dev + wl.status, n_proofs * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    H2W_HIP(hipMemcpyAsync(flag.data(), (const char *)workspace_dev + wl.lflag, n_proofs * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    H2W_HIP(hipStreamSynchronize(stream));
    for (uint64_t i = 0; i < n_proofs; i++) if (!host_status[i]) host_status[i] = flag[i];      // (the strands' conditions come first, as in round 2)
    return 0;
}
int h2w_advice_digest(const void *advice_dev, uint64_t n_cells, uint64_t *digest4_dev, void *stream_) {
    if (!advice_dev || !digest4_dev) { set_error("h2w_advice_digest: null argument"); return -1; }
    DeviceGuard dg(device_of(advice_dev));
    hipStream_t stream = (hipStream_t)stream_;
    H2W_HIP(hipMemsetAsync(digest4_dev, 0, 32, stream));
    if (n_cells) hipLaunchKernelGGL(k_digest, dim3(2048), dim3(256), 0, stream, (const ulonglong4 *)advice_dev, n_cells, (unsigned long long *)digest4_dev);
    H2W_HIP(hipGetLastError());
    return 0;
}
// canonical -> Montgomery form (halo2curves bn256::Fr in memory: v * 2^256 mod r, little-endian limbs), in place.  One Montgomery
// product per cell with the constant 2^(256+261) mod r (the device product divides by 2^261).
__global__ void k_to_montgomery(fr_t *cells, uint64_t n, fr_t kconst, uint64_t ninv) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        g_store_fr(cells + i, fr_mont_mul(g_load_fr(cells + i), kconst, ninv));
}
int h2w_advice_to_montgomery(void *cells_dev, uint64_t n_cells, void *stream_) {
    if (!cells_dev) { set_error("h2w_advice_to_montgomery: null argument"); return -1; }
    if (n_cells == 0) return 0;
    DeviceGuard dg(device_of(cells_dev));
    static const FrParams P = fr_params_init();
    static const fr_t K = [] { fr_t x = fr_from_u64(1); for (int i = 0; i < 256 + FR_MONT_BITS; i++) x = fr_add(x, x); return x; }();
    hipLaunchKernelGGL(k_to_montgomery, dim3(4096), dim3(256), 0, (hipStream_t)stream_, (fr_t *)cells_dev, n_cells, K, P.ninv);
    H2W_HIP(hipGetLastError());
    return 0;
}
int h2w_plan_timing(h2w_plan *p, uint64_t back, float ms[5]) {   // `back` batches before the last one (ring of 64)
    if (!p || !p->ev_recorded || back >= p->n_batches || back >= (uint64_t)h2w_plan::EV_RING) { set_error("h2w_plan_timing: no such batch"); return -1; }
    hipEvent_t *ev = p->evr[(p->n_batches - 1 - back) % h2w_plan::EV_RING];
    DeviceGuard dg(p->device);
    H2W_HIP(hipEventSynchronize(ev[6]));
    H2W_HIP(hipEventElapsedTime(&ms[0], ev[0], ev[1]));   // prologue strands: values (+ with PoseidonBN254 caps: their permutations' records)
    H2W_HIP(hipEventElapsedTime(&ms[1], ev[7], ev[2]));   // query glue strands (+ Goldilocks caps: Merkle strands, values, and every listed permutation's records)
    H2W_HIP(hipEventElapsedTime(&ms[2], ev[4], ev[5]));   // PoseidonBN254 Merkle chains: values + emission (on their side stream; 0 for Goldilocks-Poseidon Merkle)
    H2W_HIP(hipEventElapsedTime(&ms[3], ev[8], ev[3]));   // expansion kernel
    H2W_HIP(hipEventElapsedTime(&ms[4], ev[0], ev[6]));   // whole call
    return 0;
}
int h2w_plan_timing_ex(h2w_plan *p, uint64_t back, float ms[8]) {
    if (!p || !p->ev_recorded || back >= p->n_batches || back >= (uint64_t)h2w_plan::EV_RING) { set_error("h2w_plan_timing_ex: no such batch"); return -1; }
    hipEvent_t *ev = p->evr[(p->n_batches - 1 - back) % h2w_plan::EV_RING];
    DeviceGuard dg(p->device);
    H2W_HIP(hipEventSynchronize(ev[6]));
    H2W_HIP(hipEventElapsedTime(&ms[0], ev[0], ev[9]));    // k_prologue_values
    H2W_HIP(hipEventElapsedTime(&ms[1], ev[11], ev[12]));  // k_glp_emit
    H2W_HIP(hipEventElapsedTime(&ms[2], ev[7], p->shape.hash_mode == 0 ? ev[11] : ev[2]));   // k_strands (+ k_merkle_gl_values)
    H2W_HIP(hipEventElapsedTime(&ms[3], ev[4], ev[10]));   // k_merkle_bn_values
    H2W_HIP(hipEventElapsedTime(&ms[4], ev[10], ev[5]));   // k_merkle_bn_emit
    H2W_HIP(hipEventElapsedTime(&ms[5], ev[8], ev[3]));    // expansion kernel
    H2W_HIP(hipEventElapsedTime(&ms[6], ev[0], ev[6]));    // whole call
    ms[7] = p->shape.hash_mode == 1 ? (float)p->passes_of[(p->n_batches - 1 - back) % h2w_plan::EV_RING] : 0.f;
    return 0;
}
int h2w_plan_last_timing(h2w_plan *p, float ms[5]) { return h2w_plan_timing(p, 0, ms); }
int h2w_plan_event_gap(h2w_plan *p, uint64_t back_a, int which_a, uint64_t back_b, int which_b, float *ms) {
    static const int idx[H2W_EV_COUNT] = {0, 1, 7, 2, 4, 5, 8, 3, 6};
    if (!p || !ms || !p->ev_recorded || back_a >= p->n_batches || back_b >= p->n_batches || back_a >= (uint64_t)h2w_plan::EV_RING || back_b >= (uint64_t)h2w_plan::EV_RING ||
        which_a < 0 || which_a >= H2W_EV_COUNT || which_b < 0 || which_b >= H2W_EV_COUNT) { set_error("h2w_plan_event_gap: no such batch / event"); return -1; }
    if (p->shape.hash_mode == 0 && (which_a == H2W_EV_CHAINS_START || which_a == H2W_EV_CHAINS_END || which_b == H2W_EV_CHAINS_START || which_b == H2W_EV_CHAINS_END)) { set_error("h2w_plan_event_gap: no chain kernel with Goldilocks-Poseidon caps"); return -1; }
    hipEvent_t a = p->evr[(p->n_batches - 1 - back_a) % h2w_plan::EV_RING][idx[which_a]], b = p->evr[(p->n_batches - 1 - back_b) % h2w_plan::EV_RING][idx[which_b]];
    DeviceGuard dg(p->device);
    H2W_HIP(hipEventSynchronize(a)); H2W_HIP(hipEventSynchronize(b));
    H2W_HIP(hipEventElapsedTime(ms, a, b));
    return 0;
}
int h2w_plan_configure(h2w_plan *p, int option, int value) {
    if (!p) { set_error("h2w_plan_configure: null plan"); return -1; }
    if (option == H2W_OPT_FORK_CHAINS) { p->fork_chains = value != 0; return 0; }
    if (option == H2W_OPT_SERIAL_EXPAND) { p->serial_expand = value != 0; return 0; }      // (negative: the default, on)
    if (option == H2W_OPT_VALUES_FORM) { if (value < 0 || value > 2) { set_error("h2w_plan_configure: H2W_OPT_VALUES_FORM is 0 (by launch size), 1 (four lanes per path) or 2 (one wavefront per path)"); return -1; } p->values_form = value; return 0; }
    if (option == H2W_OPT_CHAIN_PASSES) { if (value < 0 || value > 2) { set_error("h2w_plan_configure: H2W_OPT_CHAIN_PASSES is 0 (by launch size), 1 or 2"); return -1; } p->chain_passes = value; return 0; }
    set_error("h2w_plan_configure: unknown option"); return -1;
}

}  // extern "C"
