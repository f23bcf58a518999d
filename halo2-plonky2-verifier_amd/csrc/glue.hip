// glue.hip - the two strand kernels that are serial programs of Goldilocks gadget operations:
//   k_strands          the query glue: FriChip::verify_query_round minus its Merkle proofs (fri/mod.rs:338-444: index bits, subgroup_x,
//                      combine_initial, per fold step the consistency select, compute_evaluation / interpolate_coset and x^arity,
//                      final-polynomial evaluation), one lane per (proof, query);
//   k_merkle_bn_emit   the level-parallel PoseidonBN254 emission: a quad steps over its strand to its own unit (merkle/mod.rs:57-78) - out of line that walk
//                      kept a 3 KB stack per lane, and 10752 wavefronts x 198 KB of it was 3 GB of scratch traffic per launch beside 22 GB of cells;
//   k_prologue_values  the prologue on values (stark/mod.rs:497-508, challenger/mod.rs:168-222, fri/mod.rs:45-62,130-145), one wavefront per proof.
//
// Their own translation unit because the gadget stack is compiled FLATTENED here (H2W_FLATTEN_CHIPS): a strand is a serial program of
// a few thousand Goldilocks ops, each of which appends one 32-byte block record.  Out of line, every op is an AMDGPU function call:
// it starts with s_waitcnt vmcnt(0) - i.e. it waits for the record store of the PREVIOUS op to be acknowledged by memory - and
// reaches the sink's cursor (record index, cell offset) through flat loads of the backend object (profiles/: 5.3 ms for 5.5 k ops
// per lane, ~1 us per op).  Flattened, the cursor lives in registers and nothing waits for a store.
#define H2W_FLATTEN_CHIPS 1
#include <hip/hip_runtime.h>
#include "common.h"
#include "batchargs.h"
#define RF_TAB9 s_bn_tab9
#include "rowperm.h"

namespace h2w {

// ---- the row-cooperative values pass (rowfr.h, rowperm.h): ONE wavefront per Merkle path, the four rows of the wavefront = the four elements of the
// PoseidonBN254 state, one 29-bit limb per lane.  The wavefront walks its strand wave-uniformly (every lane runs the same gadget code on the same
// values and writes nothing: the walk between two permutation units is a few selects); a permutation unit is rowperm.h bn_permute_rows.
struct RowSink {
    static constexpr bool kCoop = false, kSplitOnly = false, kBnUnits = true, kDevSponge = false; static constexpr int kHashMode = 1;
    fr_t *ustate;                  // output states of this strand's permutation units, [unit][4]
    uint32_t *sbx9;                // their partial rounds' S-box values in limb form, [unit][56][3][12]
    const rf::RowConst *rowk; int unit_local = 0;
    __device__ __forceinline__ void rec(int, uint64_t, uint64_t, uint64_t, uint64_t) {}
    __device__ __forceinline__ void cell(const fr_t &) {}
    __device__ __forceinline__ void gate() {}
    __device__ __forceinline__ void lookup() {}
    __device__ void note_cap_hash(uint64_t) {}
    __device__ __forceinline__ void skip(uint64_t, uint64_t) {}
    __device__ void merkle_begin(int, int, bool, uint64_t) {}
    __device__ void merkle_end(int, int, bool) {}
    __device__ void query_begin(int, uint64_t) {}
    __device__ void query_end(int, uint64_t) {}
    __device__ void bn_perm_begin(bool) {}
    __device__ void bn_perm_end(bool) {}
    __device__ void glp_note() {}
    __device__ void note_load(uint64_t, int) {}
    __device__ bool coop_load_proof(const ValCfg &) { return false; }
    __device__ void coop_poseidon_permute(uint64_t *, const h2w_poseidon_consts_t *) {}
    __device__ __forceinline__ bool level_skip(fr_t &, bool &) { return false; }
    __device__ __forceinline__ bool tail_skip() const { return true; }      // the cap lookup: cells only (no value anyone uses)
    __device__ __forceinline__ void permute_unit(fr_t *st) {
        rf::RowConst K;
        {   // wave-uniform: scalar loads
            const uint32_t *src = reinterpret_cast<const uint32_t *>(rowk); uint32_t *dst = reinterpret_cast<uint32_t *>(&K);
#pragma unroll
            for (unsigned i = 0; i < sizeof(rf::RowConst) / 4; i++) dst[i] = *(const __attribute__((address_space(4))) uint32_t *)(src + i);
        }
        const rf::LaneK L = rf::lane_consts();
        rf::bn_permute_rows(st, K, L, sbx9 + (size_t)unit_local * (BN_PARTIAL_ROUNDS * 3 * rf::SBX9_W));
        const unsigned lane = threadIdx.x & 63u;
#pragma unroll
        for (int i = 0; i < 4; i++) if (lane == (unsigned)i) g_store_fr(ustate + (uint64_t)unit_local * 4 + i, st[i]);
        unit_local++;
    }
    __device__ __forceinline__ bool bn_emit_inline(fr_t *st, const ValCfg &, bool &) { permute_unit(st); return true; }
};
__global__ __launch_bounds__(QUAD_BLOCK) __attribute__((flatten)) void k_merkle_bn_values_row(BatchArgs A) {
    typedef ValBackend<RowSink> RowB;
    stage_bn_consts9(A.bn_tab9, threadIdx.x, QUAD_BLOCK);    // (block-wide barrier inside: before any wavefront leaves)
    const unsigned idx = blockIdx.x * QUAD_WAVES + (threadIdx.x >> 6);      // this wavefront's (proof, query) unit
    int p, q;
    if (!own_unit_at(A, idx, p, q)) return;
    const int n_or = A.shape.n_perm_z > 0 ? 3 : 2;
    const int slot = blockIdx.y, kind = slot < n_or ? slot : 3 + (slot - n_or);
    const int sq = q == 0 ? 0 : 1;
    RowSink sink;
    const uint64_t unit0 = (uint64_t)idx * A.sh.unit_slot + A.st->mk_unit_rel[sq][kind];
    sink.ustate = A.unit_state + unit0 * 4; sink.sbx9 = A.unit_sbox9 + unit0 * (BN_PARTIAL_ROUNDS * 3 * rf::SBX9_W); sink.rowk = A.rowk;
    ValCfg mc = make_cfg(A, p); mc.split_bn = true;
    RowB be(sink, mc, !(q == 0 && kind == A.st->first_zero_kind));
    const h2w_shape_t shp = A.shape;
    Verifier<RowB> V(be, shp, A.consts);
    const uint64_t x = A.cbs[p].fri_query_indices[q];
    const int lde = V.d.lde_bits; int lo = 0;
    if (kind >= 3) for (int i = 0; i <= kind - 3; i++) lo += V.d.arity[i];
    const uint64_t cap_index = (x >> (lde - A.shape.cap_height)) & ((1ull << A.shape.cap_height) - 1);
    V.merkle_strand(q, kind, PackedBits{x, lo}, lde - lo, cap_index);
    if ((threadIdx.x & 63) == 0 && be.status) atomicCAS(&A.status[p], 0u, be.status);
}
void launch_merkle_bn_values_row(const BatchArgs &A, unsigned nkinds, hipStream_t stream) {
    const dim3 grid((A.sh.n_own_units + QUAD_WAVES - 1) / QUAD_WAVES, nkinds);
    hipLaunchKernelGGL(k_merkle_bn_values_row, grid, dim3(QUAD_BLOCK), 0, stream, A);
}

// (the backends of this unit - DevSinkT<COLS, true> and CoopSinkT<COLS, true, -1> - are instantiated nowhere else: field.h, HNI)
template <bool COLS> __global__ __launch_bounds__(64) __attribute__((flatten)) void k_strands(BatchArgs A) {
    typedef DevSinkT<COLS, true> GlueSink; typedef ValBackend<GlueSink> GlueB;
    int p, q;
    if (!own_unit_at(A, blockIdx.x * blockDim.x + threadIdx.x, p, q)) return;      // lane i of the launch = this rank's i-th (proof, query) unit
    GlueSink sink; sink.recs = A.recs + (uint64_t)p * A.rec_stride; sink.out = block_out(A, p, q); sink.ncells = A.ncells; sink.cc.init(A.cm);
    sink.nrec = strand_q_rec(*A.st, q); sink.cell_off = strand_q_cell(*A.st, q);
    const ChallengeBlock<GlueB> &cb = *reinterpret_cast<const ChallengeBlock<GlueB> *>(&A.cbs[p]);
    GlueB be(sink, make_cfg(A, p), true);
    const h2w_shape_t shp = A.shape;      // (a reference into the kernel arguments would put all of them on every lane's stack)
    Verifier<GlueB> V(be, shp, A.consts);
    V.query_round(q, cb);
    if (be.status) atomicCAS(&A.status[p], 0u, be.status);
}

// one wavefront per proof: wave-uniform gadget code; every rank of a sharded run computes every prologue's VALUES (it needs the
// challenges); only the proof's owner writes the block (direct cells, records, permutation list)
template <bool COLS> __global__ __launch_bounds__(64) __attribute__((flatten)) void k_prologue_values(BatchArgs A) {
    typedef CoopSinkT<COLS, true> Sink; typedef ValBackend<Sink> CoopB;
    __builtin_amdgcn_s_setprio(3);   // latency-bound serial strand: win issue arbitration against co-resident streaming waves
    stage_glp_consts<true>(A.consts, threadIdx.x, 64);
    const int p = blockIdx.x;
#ifdef H2W_EXP_GLP_CLOCK
    const long long k0 = clock64();
#endif
    Sink sink; coop_sink_init(sink, A, p, -1); sink.nrec = 0; sink.cell_off = 0; sink.glp_slot = 0;
    sink.emit = own_prologue(A, p);
    CoopB be(sink, make_cfg(A, p), true);
    const h2w_shape_t shp = A.shape;      // (a reference into the kernel arguments would put all of them on every lane's stack)
    Verifier<CoopB> V(be, shp, A.consts);
    V.prologue(*reinterpret_cast<ChallengeBlock<CoopB> *>(&A.cbs[p]));
#ifdef H2W_EXP_GLP_CLOCK
    if (p == 0 && threadIdx.x == 0) {      // marks (verifier.h prologue): load_proof | caps and alphas | zeta and openings | fri alpha and betas | final poly and pow | query indices | reduced openings
        const long long total = (long long)clock64() - k0;
        for (int i = 0; i < sink.dbg_k; i++) printf("mark %d at %lld cycles %lld in %d permutations\n", i, sink.dbg_t[i], sink.dbg_c[i], sink.dbg_m[i]);
        printf("kernel %lld cycles\n", total);
    }
#endif
    if (threadIdx.x == 0) A.status[p] = be.status;
}

// emission: four lanes per work item = (strand kind, permutation unit of that strand; owned unit).  The quads of a wavefront share
// the (kind, unit) and differ in the (proof, query): they reach their unit's permutation together (a wavefront whose quads emitted
// at different levels would run every one of those emissions with four lanes active).
template <bool COLS> __global__ __launch_bounds__(QUAD_BLOCK) H2W_QUAD_ATTR __attribute__((flatten)) void k_merkle_bn_emit(BatchArgs A) {
    typedef QuadSinkT<COLS, QUAD_EMIT> Sink; typedef ValBackend<Sink> QuadB;
    stage_bn_consts(A.bn_tab, threadIdx.x, QUAD_BLOCK);
    const unsigned per_q = A.st->mk_item0[MK_KINDS], upad = (A.sh.n_own_units + 15u) & ~15u;      // 16 quads = one wavefront per 16 units of an item
    const unsigned long long total = (unsigned long long)per_q * upad;
    const unsigned long long g = ((unsigned long long)blockIdx.x * QUAD_BLOCK + threadIdx.x) >> 2;
    if (g >= total) return;                                       // (whole wavefronts: total is a multiple of 16)
    // items outermost: the wavefronts in flight at one time work on the same few levels of every path of the launch (grouping all levels of a
    // few hundred units instead - a compact part of the advice - was slower: 6.65 against 5.74 ms, profiles/r03_emit_store_bound.txt)
    const unsigned item = (unsigned)(g / upad); unsigned ui = (unsigned)(g % upad);
    if (ui >= A.sh.n_own_units) ui = A.sh.n_own_units - 1;      // tail quads of an item redo its last unit (identical bytes)
    int p, q; own_unit_at(A, ui, p, q);
    int kind = 0;
#pragma unroll 1
    for (int k = 1; k < MK_KINDS; k++) if (item >= A.st->mk_item0[k]) kind = k;      // (kinds a shape does not have own no items)
    Sink sink; sink.set_window((int)(item - A.st->mk_item0[kind]), (int)A.st->mk_nunit[kind]);
    quad_strand<QuadB>(A, sink, ui, p, q, kind);
}

// values phase of the two-pass paths: four lanes per (owned unit, kind); blockIdx.y = kind slot.  In this (flattened) unit: the walk between two
// permutation units - selects, the state's constants, the index bits - is inlined instead of a chain of calls through a 1.3 KB stack frame
__global__ __launch_bounds__(QUAD_BLOCK) __attribute__((flatten)) void k_merkle_bn_values(BatchArgs A) {
    typedef QuadSinkT<false, QUAD_VALUES> Sink; typedef ValBackend<Sink> QuadB;
    stage_bn_consts9(A.bn_tab9, threadIdx.x, QUAD_BLOCK);    // (block-wide barrier inside: before any wavefront leaves)
    const unsigned total = A.sh.n_own_units;
    if (((blockIdx.x * QUAD_BLOCK + (threadIdx.x & ~63u)) >> 2) >= total) return;      // a wavefront past the last strand
    unsigned idx = (blockIdx.x * QUAD_BLOCK + threadIdx.x) >> 2;
    if (idx >= total) idx = total - 1;                  // tail quads redo the last strand (identical bytes)
    int p, q; own_unit_at(A, idx, p, q);
    const int n_or = A.shape.n_perm_z > 0 ? 3 : 2;
    const int slot = blockIdx.y, kind = slot < n_or ? slot : 3 + (slot - n_or);
    Sink sink;
    quad_strand<QuadB>(A, sink, idx, p, q, kind);
}
void launch_merkle_bn_values(const BatchArgs &A, dim3 grid, hipStream_t stream) { hipLaunchKernelGGL(k_merkle_bn_values, grid, dim3(QUAD_BLOCK), 0, stream, A); }

void launch_merkle_bn_emit(const BatchArgs &A, dim3 grid, hipStream_t stream) {
    if (A.cm.starts) hipLaunchKernelGGL(k_merkle_bn_emit<true>, grid, dim3(QUAD_BLOCK), 0, stream, A);
    else hipLaunchKernelGGL(k_merkle_bn_emit<false>, grid, dim3(QUAD_BLOCK), 0, stream, A);
}

// Goldilocks-Poseidon Merkle strands (hash_mode 0), values phase: one wavefront per (owned unit, kind); blockIdx.y = kind slot.  Flattened like the
// prologue (round 4): out of line, every hash of a path handed its state to the permutation through scratch memory and the sink lived there
template <bool COLS> __global__ __launch_bounds__(64) __attribute__((flatten)) void k_merkle_gl_values(BatchArgs A) {
    typedef CoopSinkT<COLS, true, 0> Sink; typedef ValBackend<Sink> CoopB;
    __builtin_amdgcn_s_setprio(3);
    stage_glp_consts<true>(A.consts, threadIdx.x, 64);
    int p, q;
    if (!own_unit_at(A, blockIdx.x, p, q)) return;
    const int sq = q == 0 ? 0 : 1;
    const int n_or = A.shape.n_perm_z > 0 ? 3 : 2;
    const int slot = blockIdx.y, kind = slot < n_or ? slot : 3 + (slot - n_or);
    Sink sink; coop_sink_init(sink, A, p, q); sink.emit = true;
    sink.nrec = strand_q_rec(*A.st, q) + A.st->mk_rec_rel[sq][kind]; sink.cell_off = strand_q_cell(*A.st, q) + A.st->mk_cell_rel[sq][kind];
    sink.glp_slot = A.st->pro_nglp + (uint32_t)q * A.st->q_nglp + A.st->mk_glp_rel[kind];
    CoopB be(sink, make_cfg(A, p), true);
    const h2w_shape_t shp = A.shape;      // (a reference into the kernel arguments would put all of them on every lane's stack)
    Verifier<CoopB> V(be, shp, A.consts);
    const uint64_t x = A.cbs[p].fri_query_indices[q];
    const int lde = V.d.lde_bits; int lo = 0;
    if (kind >= 3) for (int i = 0; i <= kind - 3; i++) lo += V.d.arity[i];
    const uint64_t cap_index = (x >> (lde - A.shape.cap_height)) & ((1ull << A.shape.cap_height) - 1);
    V.merkle_strand(q, kind, PackedBits{x, lo}, lde - lo, cap_index);
    if (threadIdx.x == 0 && be.status) atomicCAS(&A.status[p], 0u, be.status);
}

void launch_merkle_gl_values(const BatchArgs &A, unsigned nunits, unsigned nkinds, hipStream_t stream) {
    if (A.cm.starts) hipLaunchKernelGGL(k_merkle_gl_values<true>, dim3(nunits, nkinds), dim3(64), 0, stream, A); else hipLaunchKernelGGL(k_merkle_gl_values<false>, dim3(nunits, nkinds), dim3(64), 0, stream, A);
}
void launch_prologue_values(const BatchArgs &A, hipStream_t stream) {
    if (A.cm.starts) hipLaunchKernelGGL(k_prologue_values<true>, dim3((unsigned)A.nproofs), dim3(64), 0, stream, A);
    else hipLaunchKernelGGL(k_prologue_values<false>, dim3((unsigned)A.nproofs), dim3(64), 0, stream, A);
}

void launch_glue_strands(const BatchArgs &A, hipStream_t stream) {
    const unsigned nlanes = A.sh.n_own_units;
    if (A.cm.starts) hipLaunchKernelGGL(k_strands<true>, dim3((nlanes + 63) / 64, 1), dim3(64), 0, stream, A);
    else hipLaunchKernelGGL(k_strands<false>, dim3((nlanes + 63) / 64, 1), dim3(64), 0, stream, A);
}

}  // namespace h2w
