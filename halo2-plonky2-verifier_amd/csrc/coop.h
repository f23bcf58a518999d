// coop.h — wavefront-cooperative value kernels (device only).
//
// CoopSink: one 64-lane wavefront executes one strand of verifier.h *wave-uniformly* (every lane runs the same
// gadget code on the same values; lane 0 alone writes records / direct cells), and the lanes split up the one
// wide thing in it: the Goldilocks Poseidon permutation (hash/poseidon/permutation.rs:43-284), 12 lanes = the 12
// state elements / MDS rows, state exchanged through LDS.  It emits exactly the 2,604 records the sequential
// template code (chips.h PoseidonPermutationChip) emits, at the same indices — tests/test_gpu_batch.py compares
// the resulting advice with the oracle byte for byte.
#pragma once
#include "valbackend.h"

namespace h2w {

// PoseidonBN254 constants for the unit kernel, canonical [0] and Montgomery form [1], in CONSTANT address space:
// wave-uniform reads become scalar loads (lgkmcnt), which — unlike vector loads — do not queue behind the kernel's own
// outstanding cell stores on vmcnt.  One table per device context; re-uploaded on the stream when a plan with
// different constants runs (g_const_owner).
struct BnConsts { h2w_fr_t c[88], s[392], m[4][4], p[4][4]; };
__constant__ BnConsts c_bn[2];

constexpr int GLP_RECS_FULL = 12 + 48 + 1 + 12 * 14;           // constant_layer, sbox_layer, mds_layer
constexpr int GLP_RECS_PARTIAL_ROUND = 4 + 1 + 1 + 11 + 1 + 11; // sbox, +const, d = m00*s0, d chain, zeros, v row
constexpr int GLP_RECS_PARTIAL = 12 + 1 + 121 + N_PARTIAL_ROUNDS * GLP_RECS_PARTIAL_ROUND;
constexpr int GLP_CONST_WORDS = 360 + 12 + 12 + 12 + 22 + 121 + 242 + 242;   // u64 words of the Goldilocks block of h2w_poseidon_consts_t
constexpr int GLP_RECS = 2 * HALF_N_FULL_ROUNDS * GLP_RECS_FULL + GLP_RECS_PARTIAL;   // 2604

// The cooperating block is ONE wavefront: its LDS operations execute in program order and its lanes run in lockstep, so
// lanes exchange the Poseidon state through LDS with no barrier.  __syncthreads() would also wait for vmcnt(0), i.e. for
// every outstanding record STORE to be acknowledged by memory (~1-2 us) at each of the ~90 exchange points of a
// permutation; this waits for LDS only.
__device__ __forceinline__ void wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }

// Goldilocks-Poseidon round constants + MDS, staged in LDS by the cooperative kernels (stage_glp_consts) and read with
// ds_read (a namespace-scope __shared__ array keeps the LDS address space; a pointer member would decay to flat).
__shared__ uint64_t s_glp_k[GLP_CONST_WORDS];
constexpr int KO_ARC = 0, KO_CIRC = 360, KO_DIAG = 372, KO_FIRST = 384, KO_PRC = 396, KO_INIT = 418, KO_WHAT = 539, KO_VS = 781;
static_assert(KO_VS + 242 == GLP_CONST_WORDS, "Goldilocks constant block layout");
__device__ __forceinline__ void stage_glp_consts(const h2w_poseidon_consts_t *k, int tid, int nthreads) {
    const uint64_t *src = reinterpret_cast<const uint64_t *>(k);
    for (int i = tid; i < GLP_CONST_WORDS; i += nthreads) s_glp_k[i] = g_load_u64(src + i);
    __syncthreads();
}

__device__ __forceinline__ uint64_t gl_add_c(uint64_t x, uint64_t y) {   // canonical x + y mod p
    uint64_t t = x + y;
    if (t < x || t >= GL_P) t -= GL_P;
    return t;
}
__device__ __forceinline__ uint64_t shfl_up64(uint64_t v, int d) { return __shfl_up(v, d, 64); }

template <bool COLS> struct CoopSinkT {
    static constexpr bool kCoop = true;
    rec_t *recs; uint64_t nrec; fr_t *out; uint64_t cell_off; const uint16_t *ncells; int lane; int dbg_skip_perm = 0; ColPolicy<COLS> cc;
    __device__ __forceinline__ void rec(int t, uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
        if (lane == 0) g_store_rec(recs + nrec, a, b, c, d);
        nrec++; cell_off += ncells[t];
    }
    bool lane_mode = false, lane_on = false;     // lane_mode: every enabled lane writes cells at its OWN offset (coop_decompose_hashes)
    __device__ __forceinline__ void cell(const fr_t &v) { if (lane_mode ? lane_on : lane == 0) g_store_fr(out + cc.map(cell_off), v); cell_off++; }
    __device__ __forceinline__ void gate() {}
    __device__ __forceinline__ void lookup() {}
    __device__ bool unit_writer() const { return lane == 0; }
    __device__ int coop_lanes() { return 64; }
    __device__ int coop_lane() { return lane; }
    __device__ uint64_t lane_bcast(uint64_t v, int src) { return __shfl(v, src, 64); }
    __device__ void begin_lane_cells(uint64_t off, bool on) { cell_off = off; lane_mode = true; lane_on = on; }
    __device__ void end_lane_cells(uint64_t off) { cell_off = off; lane_mode = false; }
    __device__ __forceinline__ void skip(uint64_t nr, uint64_t nc) { nrec += nr; cell_off += nc; }
    __device__ void merkle_begin(int, int, bool, uint64_t) {}
    __device__ void merkle_end(int, int, bool) {}
    __device__ void query_begin(int, uint64_t) {}
    __device__ void query_end(int, uint64_t) {}
    __device__ void bn_perm_begin(bool) {}
    __device__ void bn_perm_end(bool) {}
    __device__ void note_load(uint64_t, int) {}
    __device__ void bn_native(fr_t *st, const h2w_poseidon_consts_t *km, const FrParams &P) { bn_poseidon_native(st, km, P); }
    __device__ bool bn_emit_inline(fr_t *, const ValCfg &, bool &) { return false; }
    // WitnessChip::load_proof_with_pis (witness/mod.rs:267-294): every item is independent -> striped over the lanes
    uint32_t load_flag = 0;      // set by coop_load_proof: some word is outside its field's canonical range (status 4)
    __device__ __noinline__ bool coop_load_proof(const ValCfg &cfg) {
        bool bad = false;
        for (uint32_t i = lane; i < cfg.n_load_items; i += 64) {
            const uint64_t *ip = reinterpret_cast<const uint64_t *>(cfg.load_items + i);
            const uint64_t wk = g_load_u64(ip), irec = g_load_u64(ip + 1), icell = g_load_u64(ip + 2);
            const uint32_t word = (uint32_t)wk, kind = (uint32_t)(wk >> 32);
            const uint64_t *w = cfg.proof + word; const uint64_t w0 = g_load_u64(w);
            if (kind <= 1) { g_store_rec(recs + irec, w0, 0, 0, 0); bad |= w0 >= GL_P; }
            else if (kind == 2) { const uint64_t w1 = g_load_u64(w + 1), w2 = g_load_u64(w + 2), w3 = g_load_u64(w + 3); g_store_rec(recs + irec, w0, w1, w2, w3); bad |= w0 >= GL_P || w1 >= GL_P || w2 >= GL_P || w3 >= GL_P; }
            else { fr_t v; v.l[0] = w0; v.l[1] = g_load_u64(w + 1); v.l[2] = g_load_u64(w + 2); v.l[3] = g_load_u64(w + 3); g_store_fr(out + cc.map(icell), v); bad |= fr_geq_mod(v); }
        }
        if (__any(bad)) load_flag = 4;
        nrec += cfg.load_nrec; cell_off += cfg.load_ncell;
        return true;
    }

    __device__ __noinline__ void coop_poseidon_permute(uint64_t *st, const h2w_poseidon_consts_t *) {
        __shared__ uint64_t s_a[SPONGE_WIDTH], s_b[SPONGE_WIDTH];
        rec_t *R = recs + nrec;
        if (dbg_skip_perm) {   // timing-only diagnostic (H2W_DBG_SKIP_PERM=1): advance the counters, skip the arithmetic
            const uint64_t nG0 = ncells[T_GLOP], nKA0 = ncells[T_KA_GLOP];
            nrec += GLP_RECS; cell_off += 2 * HALF_N_FULL_ROUNDS * (12 * nKA0 + 48 * nG0 + 12 + 12 * (1 + 13 * nKA0)) + 12 * nKA0 + 12 + 121 * nKA0 + (uint64_t)N_PARTIAL_ROUNDS * (4 * nG0 + nKA0 + nKA0 + 11 * nKA0 + 12 + 11 * nKA0);
            return;
        }
        const int l = lane, grp13 = lane / 13, idx13 = lane % 13;
        auto W = [&](int idx, uint64_t A, uint64_t B, uint64_t C) { g_store_rec(R + idx, A, B, C, 0); };
        if (l < SPONGE_WIDTH) s_a[l] = st[l];
        wave_sync();
        int base = 0, round_ctr = 0;
        auto full_round = [&]() {
            if (l < SPONGE_WIDTH) {
                uint64_t x = s_a[l]; const uint64_t rc = s_glp_k[KO_ARC + l + SPONGE_WIDTH * round_ctr];
                W(base + l, rc, 1, x); x = gl_add(x, rc);                                   // constant_layer
                const int sb = base + 12 + 4 * l;                                            // sbox_monomial: x^7
                const uint64_t x2 = gl_mul(x, x); W(sb, x, x, 0);
                const uint64_t x4 = gl_mul(x2, x2); W(sb + 1, x2, x2, 0);
                const uint64_t x6 = gl_mul(x4, x2); W(sb + 2, x4, x2, 0);
                const uint64_t x7 = gl_mul(x6, x); W(sb + 3, x6, x, 0);
                s_b[l] = x7;
            }
            wave_sync();
            const int mb = base + 60;                                                        // mds_layer
            // 12 rows x 13 terms: res_i = (sum_{j<=i} c_j v_j) mod p.  Products in parallel (lane = row_in_pass*13 + i, 4 rows
            // per pass), then a segmented inclusive scan over each 13-lane group: 1 product + 4 adds deep instead of 13 mul-adds.
            if (l == 0) W(mb, 0, 0, 0);
            uint64_t rowres = 0;
#pragma unroll
            for (int pass = 0; pass < 3; pass++) {
                const int rr = pass * 4 + grp13;
                uint64_t c = 0, v = 0, x = 0;
                if (l < 52) {
                    if (idx13 < 12) { c = s_glp_k[KO_CIRC + idx13]; int vi = idx13 + rr; if (vi >= 12) vi -= 12; v = s_b[vi]; }
                    else { c = s_glp_k[KO_DIAG + rr]; v = s_b[rr]; }
                    x = gl_mul(c, v);
                }
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) { const uint64_t y = shfl_up64(x, d); if (idx13 >= d) x = gl_add_c(x, y); }
                uint64_t prev = shfl_up64(x, 1); if (idx13 == 0) prev = 0;
                if (l < 52) {
                    const int rb = mb + 1 + 14 * rr;
                    if (idx13 == 0) W(rb, 0, 0, 0);
                    W(rb + 1 + idx13, c, v, prev);
                    if (idx13 == 12) rowres = x;
                }
                if (l < 52 && idx13 == 12) s_a[rr] = rowres;
            }
            wave_sync();
            base += GLP_RECS_FULL; round_ctr++;
        };
        for (int i = 0; i < HALF_N_FULL_ROUNDS; i++) full_round();
        // ---- partial rounds (hash/poseidon/permutation.rs:216-239)
        if (l < SPONGE_WIDTH) {                                                              // partial_first_constant_layer
            const uint64_t x = s_a[l], c = s_glp_k[KO_FIRST + l];
            W(base + l, c, 1, x); s_b[l] = gl_add(x, c);
        }
        wave_sync();
        base += 12;
        if (l == 0) { W(base, 0, 0, 0); s_a[0] = s_b[0]; }                                   // mds_partial_layer_init
        else if (l < SPONGE_WIDTH) {
            uint64_t res = 0;
            for (int r = 1; r < SPONGE_WIDTH; r++) {
                const uint64_t t = s_glp_k[KO_INIT + (r - 1) * 11 + (l - 1)], v = s_b[r];
                W(base + 1 + (r - 1) * 11 + (l - 1), t, v, res); res = gl_muladd(t, v, res);
            }
            s_a[l] = res;
        }
        wave_sync();
        base += 122;
        for (int r = 0; r < N_PARTIAL_ROUNDS; r++) {
            if (l == 0) {
                const uint64_t x = s_a[0];
                const uint64_t x2 = gl_mul(x, x); W(base, x, x, 0);
                const uint64_t x4 = gl_mul(x2, x2); W(base + 1, x2, x2, 0);
                const uint64_t x6 = gl_mul(x4, x2); W(base + 2, x4, x2, 0);
                const uint64_t x7 = gl_mul(x6, x); W(base + 3, x6, x, 0);
                const uint64_t c = s_glp_k[KO_PRC + r];
                W(base + 4, c, 1, x7); s_a[0] = gl_add(x7, c);
            }
            wave_sync();
            const uint64_t s0 = s_a[0];                                                      // mds_partial_layer_fast
            {   // d = m00*s0 + sum_i w_hat_i * st_i as an inclusive scan over lanes 0..11 (records need every partial sum)
                uint64_t t = 0, v = 0, x = 0;
                if (l == 0) { t = s_glp_k[KO_CIRC] + s_glp_k[KO_DIAG]; v = s0; }
                else if (l < SPONGE_WIDTH) { t = s_glp_k[KO_WHAT + r * 11 + (l - 1)]; v = s_a[l]; }
                if (l < SPONGE_WIDTH) x = gl_mul(t, v);
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) { const uint64_t y = shfl_up64(x, d); if (l >= d) x = gl_add_c(x, y); }
                uint64_t prev = shfl_up64(x, 1); if (l == 0) prev = 0;
                if (l < SPONGE_WIDTH) W(base + 5 + l, t, v, prev);
                if (l == 0) W(base + 17, 0, 0, 0);
                if (l == SPONGE_WIDTH - 1) s_b[0] = x;
                if (l > 0 && l < SPONGE_WIDTH) {
                    const uint64_t tv = s_glp_k[KO_VS + r * 11 + (l - 1)];
                    W(base + 17 + l, tv, s0, v); s_b[l] = gl_muladd(tv, s0, v);
                }
            }
            wave_sync();
            if (l < SPONGE_WIDTH) s_a[l] = s_b[l];
            wave_sync();
            base += GLP_RECS_PARTIAL_ROUND;
        }
        round_ctr += N_PARTIAL_ROUNDS;
        for (int i = 0; i < HALF_N_FULL_ROUNDS; i++) full_round();
        for (int i = 0; i < SPONGE_WIDTH; i++) st[i] = s_a[i];
        wave_sync();
        const uint64_t nG = ncells[T_GLOP], nKA = ncells[T_KA_GLOP];
        const uint64_t full_cells = 12 * nKA + 48 * nG + 12 + 12 * (1 + 13 * nKA);
        const uint64_t part_cells = 12 * nKA + 12 + 121 * nKA + (uint64_t)N_PARTIAL_ROUNDS * (4 * nG + nKA + nKA + 11 * nKA + 12 + 11 * nKA);
        nrec += GLP_RECS; cell_off += 2 * HALF_N_FULL_ROUNDS * full_cells + part_cells;
    }
};
typedef CoopSinkT<false> CoopSink;

// ---------------------------------------------------------------------------------------------------------------
// QuadSink: four adjacent lanes execute one BN254 Merkle chain strand quad-uniformly (lane 0 of the quad writes the
// strand's direct cells) and split the width-4 PoseidonBN254 state between them for the value-domain permutation:
// full rounds 4-way parallel (x^5, then each lane one output row of the mix), partial rounds: lane 0 does the S-box,
// all four lanes one product of the sparse row, lanes 1-3 their column update (5 instead of 14 dependent products).
// Montgomery product as a real call, for code that is not store-bound.  NOT for the quad emitter: every AMDGPU function entry starts
// with s_waitcnt vmcnt(0), i.e. waits for all cell stores in flight.
__device__ __attribute__((noinline)) fr_t mont_call(fr_t a, fr_t b, uint64_t ninv) { return fr_mont_mul(a, b, ninv); }

// per-quad LDS staging regions of the BN254 quad emitter (QuadSink::stage / flush): 16 quads x QST cells x 32 B per wavefront
// (QST 64: 32 KB, five wavefronts per CU by LDS; QST 32: 16 KB, and the 4-lane layers are staged in two passes)
#ifndef H2W_QST
#define H2W_QST 64
#endif
#ifndef H2W_QUAD_BLOCK
#define H2W_QUAD_BLOCK 64
#endif
#ifndef H2W_BN_LDS
#define H2W_BN_LDS 0
#endif
constexpr int QST = H2W_QST;
static_assert(QST == 32 || QST == 64, "the emitter stages a 64-cell mix layer whole (QST 64) or in two halves (QST 32)");
constexpr int QHALVES = QST >= 64 ? 1 : 2;                  // staging passes per 4-lane layer: with QST 32 lanes 0-1 stage and flush first, then lanes 2-3
constexpr int QUAD_BLOCK = H2W_QUAD_BLOCK;                   // threads per block of k_merkle_bn_quad: 4 wavefronts share one LDS copy of the constants
struct __attribute__((aligned(16))) sq16_t { unsigned long long x, y; };
__shared__ sq16_t s_quad_stage[(QUAD_BLOCK / 4) * QST * 2];
// The PoseidonBN254 tables (c_bn[0] canonical for the cells, c_bn[1] pre-multiplied by R for the products) in LDS.  The emitter
// indexes them by lane; from constant/global memory that is a vector load, and on gfx9-family parts loads and stores share
// vmcnt in order, so waiting for a constant means waiting for every cell store issued before it (microseconds of HBM write
// latency per flush).  LDS reads count on lgkmcnt and leave the stores in flight.
enum { BK_C = 0, BK_S = 88, BK_M = 88 + 392, BK_P = 88 + 392 + 16, BK_N = 88 + 392 + 32 };
#if H2W_BN_LDS
__shared__ sq16_t s_bn_k[2 * BK_N * 2];
__device__ __forceinline__ void stage_bn_consts(int tid, int nthreads) {
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(c_bn);
    for (int i = tid; i < 2 * BK_N * 2; i += nthreads) s_bn_k[i] = sq16_t{src[2 * i], src[2 * i + 1]};
    __syncthreads();
}
__device__ __forceinline__ fr_t bnk(int which, int idx) {
    const sq16_t a = s_bn_k[(which * BK_N + idx) * 2], b = s_bn_k[(which * BK_N + idx) * 2 + 1];
    fr_t r; r.l[0] = a.x; r.l[1] = a.y; r.l[2] = b.x; r.l[3] = b.y; return r;
}
#else
__device__ __forceinline__ void stage_bn_consts(int, int) {}
__device__ __forceinline__ fr_t bnk(int which, int idx) { return c_bn[which].c[idx]; }   // c, s, m, p are contiguous: one flat index (address space stays constant memory)
#endif

template <bool COLS> struct QuadSinkT {
    static constexpr bool kCoop = false;
    rec_t *recs; uint64_t nrec; fr_t *out; uint64_t cell_off; const uint16_t *ncells; int l4; ColPolicy<COLS> cc;
    __device__ __forceinline__ void rec(int t, uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
        if (l4 == 0) g_store_rec(recs + nrec, a, b, c, d);
        nrec++; cell_off += ncells[t];
    }
    __device__ __forceinline__ void cell(const fr_t &v) { if (l4 == 0) g_store_fr(out + cc.map(cell_off), v); cell_off++; }
    __device__ __forceinline__ void gate() {}
    __device__ __forceinline__ void lookup() {}
    __device__ bool unit_writer() const { return l4 == 0; }
    __device__ int coop_lanes() { return 1; }
    __device__ int coop_lane() { return 0; }
    __device__ uint64_t lane_bcast(uint64_t v, int) { return v; }
    __device__ void begin_lane_cells(uint64_t, bool) {}
    __device__ void end_lane_cells(uint64_t) {}
    __device__ __forceinline__ void skip(uint64_t nr, uint64_t nc) { nrec += nr; cell_off += nc; }
    __device__ void merkle_begin(int, int, bool, uint64_t) {}
    __device__ void merkle_end(int, int, bool) {}
    __device__ void query_begin(int, uint64_t) {}
    __device__ void query_end(int, uint64_t) {}
    __device__ void bn_perm_begin(bool) {}
    __device__ void bn_perm_end(bool) {}
    __device__ void note_load(uint64_t, int) {}
    __device__ bool coop_load_proof(const ValCfg &) { return false; }
    __device__ void coop_poseidon_permute(uint64_t *, const h2w_poseidon_consts_t *) {}
    static __device__ __forceinline__ fr_t shfl4(const fr_t &v, int src) {     // value of lane `src` of this quad
        fr_t r;
#pragma unroll
        for (int i = 0; i < 4; i++) r.l[i] = __shfl(v.l[i], src, 4);
        return r;
    }
    static __device__ __forceinline__ fr_t shfl4_xor(const fr_t &v, int m) {
        fr_t r;
#pragma unroll
        for (int i = 0; i < 4; i++) r.l[i] = __shfl_xor(v.l[i], m, 4);
        return r;
    }
    // Staged cell store.  A lane that writes a whole 32-byte cell alone issues two 16-byte stores into its own cache line; 64
    // such lines per store instruction hold the write path to ~3 TB/s device-wide (tools/ubench_store2.hip), while a quad writing
    // 64 contiguous bytes per instruction streams at the HBM ceiling.  So the owner lanes put their cells into the quad's LDS
    // region (s_quad_stage, QST cells) and the quad flushes the region, which is contiguous in the advice stream, with all four
    // lanes writing adjacent 16-byte pieces.  `off` is the cell offset inside the region, negative when this lane has no cell.
    static __device__ __forceinline__ void stage(int off, const fr_t &v) {
        if (off >= 0) { sq16_t *q = reinterpret_cast<sq16_t *>(s_quad_stage) + ((threadIdx.x >> 2) * QST + off) * 2; q[0] = sq16_t{v.l[0], v.l[1]}; q[1] = sq16_t{v.l[2], v.l[3]}; }
    }
    // (static, lane index / cursor by value: a member read after wave_sync's memory clobber is a FLAT load from the sink object, and a
    // flat load waits for vmcnt(0), i.e. for every cell store still in flight)
    static __device__ __forceinline__ void flush_run(fr_t *dst_cells, int first, int n, const int l4) {      // staged cells [first, first + n) -> dst_cells[0..n)
        const sq16_t *src = reinterpret_cast<const sq16_t *>(s_quad_stage) + ((threadIdx.x >> 2) * QST + first) * 2;
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(dst_cells);
        const int np = 2 * n;                                        // 16-byte pieces; lane k of the quad takes pieces k, k+4, ...
#pragma unroll
        for (int u = 0; u < QST / 2; u += 4) {                       // 4 LDS reads in flight, then their 4 stores
            sq16_t t[4];
#pragma unroll
            for (int v = 0; v < 4; v++) { const int pc = (u + v) * 4 + l4; if (pc < np) t[v] = src[pc]; }
#pragma unroll
#ifdef H2W_ABL_NOFLUSH      // timing-only ablation: the staged cells are read back but not written to memory
            for (int v = 0; v < 4; v++) { const int pc = (u + v) * 4 + l4; if (pc < np && t[v].x == 0x123456789abcull) { H2W_GSTORE64(dst + 2 * pc, t[v].x); H2W_GSTORE64(dst + 2 * pc + 1, t[v].y); } }
#else
            for (int v = 0; v < 4; v++) { const int pc = (u + v) * 4 + l4; if (pc < np) { H2W_GSTORE64(dst + 2 * pc, t[v].x); H2W_GSTORE64(dst + 2 * pc + 1, t[v].y); } }
#endif
            if ((u + 4) * 4 >= np) break;
        }
    }
    // column-major layout: a region can straddle one column boundary (out of line: only this mode pays for it)
    static __device__ __noinline__ void flush_cols(fr_t *out, ColPolicy<true> &cc, uint64_t cell0, int n, const int l4) {
        const uint64_t a0 = cc.map(cell0);
        const int n0 = (cc.hi - cell0 < (uint64_t)n) ? (int)(cc.hi - cell0) : n;
        flush_run(out + a0, 0, n0, l4);
        if (n0 < n) { const uint64_t a1 = cc.map(cell0 + (uint64_t)n0); flush_run(out + a1, n0, n - n0, l4); }
    }
    // flush the n staged cells to the advice cells [cell0, cell0 + n)
    static __device__ __forceinline__ void flush(fr_t *out, ColPolicy<COLS> &cc, uint64_t &cell0, int &n, const int l4) {
        wave_sync();
        if constexpr (!COLS) flush_run(out + cell0, 0, n, l4);
        else flush_cols(out, cc, cell0, n, l4);
        wave_sync();
        cell0 += (uint64_t)n; n = 0;
    }
    // PoseidonBN254 permutation with its 4,032 cells emitted by the quad itself (hash/poseidon_bn254/permutation.rs:48-203):
    // lane i owns state element i and the cells of "its" ops (x^5 of element i, ark i, row i of the mix, term j of the
    // sparse row, column update k), each a contiguous run of 5-16 cells.  Hybrid arithmetic: canonical state, constants
    // pre-multiplied by R for const*var products, x^5 in 5 Montgomery products.  No second pass over the permutations.
    __device__ __noinline__ bool bn_emit_inline(fr_t *st, const ValCfg &cfg, bool &zc_ref) {
        bool zc = zc_ref;                      // by value: a reference would be re-read with a flat load (vmcnt(0)) at every mix
        const int l = l4; const uint64_t ninv = cfg.P.ninv; const fr_t r2 = cfg.P.r2;
        uint64_t base = cell_off;             // flat cell index of the staging region's first cell
        fr_t *const outp = out;      // (by value: see flush)
        int so = 0;                           // cells staged (quad-uniform)
        fr_t s = l == 0 ? st[0] : l == 1 ? st[1] : l == 2 ? st[2] : st[3];
#ifdef H2W_ABL_NOSTORE
        auto W = [&](int off, const fr_t &v) { if (cfg.P.ninv == 12345) stage(off < 0 ? -1 : so + off, v); };
#else
        auto W = [&](int off, const fr_t &v) { stage(off < 0 ? -1 : so + off, v); };
#endif
        auto W64 = [&](int off, uint64_t v) { W(off, fr_from_u64(v)); };
        auto need = [&](int n) { if (so + n > QST) flush(outp, cc, base, so, l); };
        // x^5: 12 cells at region offset p (p < 0: this lane has no S-box here and keeps its s)
        auto exp5_all = [&]() {                          // S-box on every lane: 4 x 12 cells, lane l's at 12 l
            const fr_t X = fr_mont_mul(s, r2, ninv);
            const fr_t x2 = fr_mont_mul(s, X, ninv), X2 = fr_mont_mul(X, X, ninv);
            const fr_t x4 = fr_mont_mul(x2, X2, ninv), x5 = fr_mont_mul(x4, X, ninv);
#pragma unroll
            for (int h = 0; h < QHALVES; h++) {
                need(48 / QHALVES);
                const int q = QHALVES == 1 ? 12 * l : ((l >> 1) == h ? 12 * (l & 1) : -64);      // q + i stays negative for a lane without cells in this pass
                W64(q, 0); W(q + 1, s); W(q + 2, s); W(q + 3, x2);
                W64(q + 4, 0); W(q + 5, x2); W(q + 6, x2); W(q + 7, x4);
                W64(q + 8, 0); W(q + 9, x4); W(q + 10, s); W(q + 11, x5);
                so += 48 / QHALVES;
            }
            s = x5;
        };
        auto add_const = [&](int p, const fr_t &c) {                  // [c][s, c, 1, s+c]
            const int q = p < 0 ? -64 : p; const fr_t ns = fr_add(s, c);
            W(q, c); W(q + 1, s); W(q + 2, c); W64(q + 3, 1); W(q + 4, ns);
            if (p >= 0) s = ns;
        };
        // (constants are requested before a possible flush, see the partial rounds)
        auto ark = [&](int it) { const fr_t kc = bnk(0, BK_C + it + l); need(20); add_const(5 * l, kc); so += 20; };
        auto mix = [&](int which) {                      // which 0: M, 1: P
            if (!zc) { need(1); W64(l == 0 ? 0 : -64, 0); so += 1; zc = true; }
            if constexpr (QHALVES == 1) {
                need(64);
                fr_t acc = fr_zero(); const int p = 16 * l;
                for (int j = 0; j < 4; j++) {
                    const fr_t sj = shfl4(s, j);
                    const fr_t nacc = fr_add(fr_mont_mul(sj, bnk(1, (which ? BK_P : BK_M) + 4 * j + l), ninv), acc);
                    W(p + 4 * j, acc); W(p + 4 * j + 1, bnk(0, (which ? BK_P : BK_M) + 4 * j + l)); W(p + 4 * j + 2, sj); W(p + 4 * j + 3, nacc); acc = nacc;
                }
                s = acc; so += 64;
            } else {
                // the canonical constants are requested before the first flush (loads and stores share vmcnt in order); the row's
                // partial sums stay in registers across the two staging passes
                fr_t kc[4], acc[5];
#pragma unroll
                for (int j = 0; j < 4; j++) kc[j] = bnk(0, (which ? BK_P : BK_M) + 4 * j + l);
                acc[0] = fr_zero();
#pragma unroll
                for (int j = 0; j < 4; j++) acc[j + 1] = fr_add(fr_mont_mul(shfl4(s, j), bnk(1, (which ? BK_P : BK_M) + 4 * j + l), ninv), acc[j]);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    need(32);
                    const int p = (l >> 1) == h ? 16 * (l & 1) : -64;
#pragma unroll
                    for (int j = 0; j < 4; j++) { const fr_t sj = shfl4(s, j); W(p + 4 * j, acc[j]); W(p + 4 * j + 1, kc[j]); W(p + 4 * j + 2, sj); W(p + 4 * j + 3, acc[j + 1]); }
                    so += 32;
                }
                s = acc[4];
            }
        };
        auto consts32 = [&]() {                          // load_constant of M then P (full_rounds prologue)
            need(32);
            for (int u = 0; u < 8; u++) { const int t = 8 * l + u; W(t, bnk(0, BK_M + t)); }
            so += 32;
        };
        ark(0);
        for (int half = 0; half < 2; half++) {
            if (half == 1) {
                for (int r = 0; r < BN_PARTIAL_ROUNDS; r++) {
                    // This round's constants are fetched BEFORE the flush of the previous round's cells: loads and stores share vmcnt
                    // in order, so a constant requested after the flush's stores would wait for all of them to reach memory.
                    const int ix = (BN_WIDTH * 2 - 1) * r + l, lm = l > 0 ? l - 1 : 0, iy = (BN_WIDTH * 2 - 1) * r + BN_WIDTH + lm;
                    const fr_t kc = bnk(0, BK_C + (BN_FULL_ROUNDS / 2 + 1) * BN_WIDTH + r);
                    const fr_t ksx = bnk(0, BK_S + ix), ksxm = bnk(1, BK_S + ix), ksy = bnk(0, BK_S + iy), ksym = bnk(1, BK_S + iy);
                    need(QST >= 64 ? 52 : 17);             // (QST 32: the round's three cell groups - 17, 20, 15 - are placed one by one)
                    // Five wavefront-level Montgomery products per partial round instead of seven: the lanes that have no S-box do
                    // their own work inside the S-box's instruction stream.
                    //   A: lane 0  X = s0 R          | lanes 1-3  S_j * s_j          (their terms of the sparse row; s_j is not touched by the S-box)
                    //   B: lane 0  x2 = s0 X         | lane 1     X2 = X X
                    //   C: lane 0  x4 = x2 X2 ;  D: lane 0  x5 = x4 X ;  then s0' = x5 + c
                    //   E: lane 0  S_0 * s0'         | lanes 1-3  S'_k * s0'         (column update)
                    const fr_t A_ = fr_mont_mul(s, l == 0 ? r2 : ksxm, ninv);
                    const fr_t Xb = shfl4(A_, 0);
                    const fr_t B_ = fr_mont_mul(l == 0 ? s : Xb, Xb, ninv);
                    const fr_t X2b = shfl4(B_, 1);
                    const fr_t x4 = fr_mont_mul(B_, X2b, ninv), x5 = fr_mont_mul(x4, Xb, ninv);
                    {   // lane 0's S-box cells [0,s,s,x2, 0,x2,x2,x4, 0,x4,s,x5] and +c [c][x5, c, 1, x5+c]
                        const int q = l == 0 ? 0 : -64; const fr_t ns = fr_add(x5, kc);
                        W64(q, 0); W(q + 1, s); W(q + 2, s); W(q + 3, B_);
                        W64(q + 4, 0); W(q + 5, B_); W(q + 6, B_); W(q + 7, x4);
                        W64(q + 8, 0); W(q + 9, x4); W(q + 10, s); W(q + 11, x5);
                        W(q + 12, kc); W(q + 13, x5); W(q + 14, kc); W64(q + 15, 1); W(q + 16, ns);
                        if (l == 0) s = ns;
                    }
                    so += 17;
                    if constexpr (QST < 64) need(20);
                    const fr_t s0 = shfl4(s, 0);
                    const fr_t E_ = fr_mont_mul(s0, l == 0 ? ksxm : ksym, ninv);
                    const fr_t pr = l == 0 ? E_ : A_;                                    // S[j] * s_j
                    fr_t incl = pr, t = shfl4_up(incl, 1); if (l >= 1) incl = fr_add(incl, t);
                    t = shfl4_up(incl, 2); if (l >= 2) incl = fr_add(incl, t);
                    fr_t excl = shfl4_up(incl, 1); if (l == 0) excl = fr_zero();
                    { const int p = 5 * l; W(p, ksx); W(p + 1, excl); W(p + 2, ksx); W(p + 3, s); W(p + 4, incl); }
                    so += 20;
                    if constexpr (QST < 64) need(15);
                    const fr_t ns0 = shfl4(incl, 3);
                    {
                        const int p = l > 0 ? 5 * lm : -64;
                        const fr_t nv = fr_add(E_, s);
                        W(p, ksy); W(p + 1, s); W(p + 2, ksy); W(p + 3, s0); W(p + 4, nv);
                        s = l > 0 ? nv : ns0;
                    }
                    so += 15;
                }
            }
            consts32();
            for (int r = 0; r < BN_FULL_ROUNDS / 2 - 1; r++) {
                exp5_all();
                ark(half == 0 ? (r + 1) * BN_WIDTH : (BN_FULL_ROUNDS / 2 + 1) * BN_WIDTH + BN_PARTIAL_ROUNDS + r * BN_WIDTH);
                mix(0);
            }
            exp5_all();
            if (half == 0) { ark((BN_FULL_ROUNDS / 2) * BN_WIDTH); mix(1); } else mix(0);
        }
        flush(outp, cc, base, so, l);
        for (int j = 0; j < 4; j++) st[j] = shfl4(s, j);
        cell_off = base; zc_ref = zc;
        return true;
    }
    static __device__ __forceinline__ fr_t shfl4_up(const fr_t &v, int d) {
        fr_t r;
#pragma unroll
        for (int i = 0; i < 4; i++) r.l[i] = __shfl_up(v.l[i], d, 4);
        return r;
    }
    __device__ __noinline__ void bn_native(fr_t *st, const h2w_poseidon_consts_t *km, const FrParams &P) {
        const int l = l4; const uint64_t ninv = P.ninv;
        fr_t mine = l == 0 ? st[0] : l == 1 ? st[1] : l == 2 ? st[2] : st[3];
        fr_t s = fr_mont_mul(mine, P.r2, ninv);
        auto exp5 = [&](const fr_t &x) { fr_t x2 = fr_mont_mul(x, x, ninv), x4 = fr_mont_mul(x2, x2, ninv); return fr_mont_mul(x4, x, ninv); };
        auto mix = [&](const h2w_fr_t (*m)[4]) {
            fr_t acc = fr_zero();
            for (int j = 0; j < 4; j++) { const fr_t sj = shfl4(s, j); acc = fr_add(acc, fr_mont_mul(m[j][l], sj, ninv)); }
            s = acc;
        };
        s = fr_add(s, km->bn_c[l]);
        for (int r = 0; r < BN_FULL_ROUNDS / 2 - 1; r++) { s = exp5(s); s = fr_add(s, km->bn_c[(r + 1) * BN_WIDTH + l]); mix(km->bn_m); }
        s = exp5(s); s = fr_add(s, km->bn_c[(BN_FULL_ROUNDS / 2) * BN_WIDTH + l]); mix(km->bn_p);
        for (int r = 0; r < BN_PARTIAL_ROUNDS; r++) {
            if (l == 0) s = fr_add(exp5(s), km->bn_c[(BN_FULL_ROUNDS / 2 + 1) * BN_WIDTH + r]);
            const fr_t s0 = shfl4(s, 0);
            fr_t p = fr_mont_mul(km->bn_s[(BN_WIDTH * 2 - 1) * r + l], s, ninv);            // S[j] * s_j
            if (l > 0) s = fr_add(s, fr_mont_mul(km->bn_s[(BN_WIDTH * 2 - 1) * r + BN_WIDTH + l - 1], s0, ninv));
            p = fr_add(p, shfl4_xor(p, 1)); p = fr_add(p, shfl4_xor(p, 2));                     // sum over the quad
            if (l == 0) s = p;
        }
        for (int r = 0; r < BN_FULL_ROUNDS / 2 - 1; r++) { s = exp5(s); s = fr_add(s, km->bn_c[(BN_FULL_ROUNDS / 2 + 1) * BN_WIDTH + BN_PARTIAL_ROUNDS + r * BN_WIDTH + l]); mix(km->bn_m); }
        s = exp5(s); mix(km->bn_m);
        const fr_t o = fr_mont_mul(s, fr_from_u64(1), ninv);
        for (int j = 0; j < 4; j++) st[j] = shfl4(o, j);
    }
};
typedef QuadSinkT<false> QuadSink;

}  // namespace h2w
