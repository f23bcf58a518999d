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

struct CoopSink {
    static constexpr bool kCoop = true;
    rec_t *recs; uint64_t nrec; fr_t *out; uint64_t cell_off; const uint16_t *ncells; int lane; int dbg_skip_perm = 0;
    __device__ __forceinline__ void rec(int t, uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
        if (lane == 0) g_store_rec(recs + nrec, a, b, c, d);
        nrec++; cell_off += ncells[t];
    }
    bool lane_mode = false, lane_on = false;     // lane_mode: every enabled lane writes cells at its OWN offset (coop_decompose_hashes)
    __device__ __forceinline__ void cell(const fr_t &v) { if (lane_mode ? lane_on : lane == 0) g_store_fr(out + cell_off, v); cell_off++; }
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
    __device__ __noinline__ bool coop_load_proof(const ValCfg &cfg) {
        for (uint32_t i = lane; i < cfg.n_load_items; i += 64) {
            const uint64_t *ip = reinterpret_cast<const uint64_t *>(cfg.load_items + i);
            const uint64_t wk = g_load_u64(ip), irec = g_load_u64(ip + 1), icell = g_load_u64(ip + 2);
            const uint32_t word = (uint32_t)wk, kind = (uint32_t)(wk >> 32);
            const uint64_t *w = cfg.proof + word; const uint64_t w0 = g_load_u64(w);
            if (kind <= 1) g_store_rec(recs + irec, w0, 0, 0, 0);
            else if (kind == 2) g_store_rec(recs + irec, w0, g_load_u64(w + 1), g_load_u64(w + 2), g_load_u64(w + 3));
            else { fr_t v; v.l[0] = w0; v.l[1] = g_load_u64(w + 1); v.l[2] = g_load_u64(w + 2); v.l[3] = g_load_u64(w + 3); g_store_fr(out + icell, v); }
        }
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

// ---------------------------------------------------------------------------------------------------------------
// QuadSink: four adjacent lanes execute one BN254 Merkle chain strand quad-uniformly (lane 0 of the quad writes the
// strand's direct cells) and split the width-4 PoseidonBN254 state between them for the value-domain permutation:
// full rounds 4-way parallel (x^5, then each lane one output row of the mix), partial rounds: lane 0 does the S-box,
// all four lanes one product of the sparse row, lanes 1-3 their column update (5 instead of 14 dependent products).
struct QuadSink {
    static constexpr bool kCoop = false;
    rec_t *recs; uint64_t nrec; fr_t *out; uint64_t cell_off; const uint16_t *ncells; int l4;
    __device__ __forceinline__ void rec(int t, uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
        if (l4 == 0) g_store_rec(recs + nrec, a, b, c, d);
        nrec++; cell_off += ncells[t];
    }
    __device__ __forceinline__ void cell(const fr_t &v) { if (l4 == 0) g_store_fr(out + cell_off, v); cell_off++; }
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
    // PoseidonBN254 permutation with its 4,032 cells emitted by the quad itself (hash/poseidon_bn254/permutation.rs:48-203):
    // lane i owns state element i and writes the cells of "its" ops (x^5 of element i, ark i, row i of the mix, term j of the
    // sparse row, column update k), each a contiguous run of 5-16 cells.  Hybrid arithmetic: canonical state, constants
    // pre-multiplied by R for const*var products, x^5 in 5 Montgomery products.  No second pass over the permutations.
    __device__ __noinline__ bool bn_emit_inline(fr_t *st, const ValCfg &cfg, bool &zc) {
        const int l = l4; const uint64_t ninv = cfg.P.ninv; const fr_t r2 = cfg.P.r2;
        fr_t *base = out + cell_off;
        fr_t s = l == 0 ? st[0] : l == 1 ? st[1] : l == 2 ? st[2] : st[3];
        auto W = [&](fr_t *p, const fr_t &v) { g_store_fr(p, v); };
        auto W64 = [&](fr_t *p, uint64_t v) { g_store_fr(p, fr_from_u64(v)); };
        auto exp5 = [&](fr_t *p) {                      // 12 cells at p
            const fr_t X = fr_mont_mul(s, r2, ninv);
            const fr_t x2 = fr_mont_mul(s, X, ninv), X2 = fr_mont_mul(X, X, ninv);
            const fr_t x4 = fr_mont_mul(x2, X2, ninv), x5 = fr_mont_mul(x4, X, ninv);
            W64(p, 0); W(p + 1, s); W(p + 2, s); W(p + 3, x2);
            W64(p + 4, 0); W(p + 5, x2); W(p + 6, x2); W(p + 7, x4);
            W64(p + 8, 0); W(p + 9, x4); W(p + 10, s); W(p + 11, x5);
            s = x5;
        };
        auto add_const = [&](fr_t *p, const fr_t &c) { W(p, c); W(p + 1, s); W(p + 2, c); W64(p + 3, 1); s = fr_add(s, c); W(p + 4, s); };   // [c][s, c, 1, s+c]
        auto ark = [&](int it) { add_const(base + 5 * l, c_bn[0].c[it + l]); base += 20; };
        auto mix = [&](int which) {                      // which 0: M, 1: P
            if (!zc) { if (l == 0) W64(base, 0); base += 1; zc = true; }
            fr_t acc = fr_zero(); fr_t *p = base + 16 * l;
            for (int j = 0; j < 4; j++) {
                const fr_t sj = shfl4(s, j);
                const fr_t mc = which ? c_bn[0].p[j][l] : c_bn[0].m[j][l], mm = which ? c_bn[1].p[j][l] : c_bn[1].m[j][l];
                const fr_t nacc = fr_add(fr_mont_mul(sj, mm, ninv), acc);
                W(p + 4 * j, acc); W(p + 4 * j + 1, mc); W(p + 4 * j + 2, sj); W(p + 4 * j + 3, nacc); acc = nacc;
            }
            s = acc; base += 64;
        };
        auto consts32 = [&]() {                          // load_constant of M then P (full_rounds prologue)
            for (int u = 0; u < 8; u++) { const int t = 8 * l + u; W(base + t, t < 16 ? c_bn[0].m[t >> 2][t & 3] : c_bn[0].p[(t - 16) >> 2][t & 3]); }
            base += 32;
        };
        ark(0);
        for (int half = 0; half < 2; half++) {
            if (half == 1) {
                for (int r = 0; r < BN_PARTIAL_ROUNDS; r++) {
                    if (l == 0) { exp5(base); add_const(base + 12, c_bn[0].c[(BN_FULL_ROUNDS / 2 + 1) * BN_WIDTH + r]); }
                    base += 17;
                    const fr_t s0 = shfl4(s, 0);
                    const int ix = (BN_WIDTH * 2 - 1) * r + l;
                    const fr_t pr = fr_mont_mul(s, c_bn[1].s[ix], ninv);            // S[j] * s_j (lane 0: new s0)
                    fr_t incl = pr, t = shfl4_up(incl, 1); if (l >= 1) incl = fr_add(incl, t);
                    t = shfl4_up(incl, 2); if (l >= 2) incl = fr_add(incl, t);
                    fr_t excl = shfl4_up(incl, 1); if (l == 0) excl = fr_zero();
                    { fr_t *p = base + 5 * l; W(p, c_bn[0].s[ix]); W(p + 1, excl); W(p + 2, c_bn[0].s[ix]); W(p + 3, s); W(p + 4, incl); }
                    base += 20;
                    const fr_t ns0 = shfl4(incl, 3);
                    if (l > 0) {
                        const int iy = (BN_WIDTH * 2 - 1) * r + BN_WIDTH + l - 1; fr_t *p = base + 5 * (l - 1);
                        const fr_t nv = fr_add(fr_mont_mul(s0, c_bn[1].s[iy], ninv), s);
                        W(p, c_bn[0].s[iy]); W(p + 1, s); W(p + 2, c_bn[0].s[iy]); W(p + 3, s0); W(p + 4, nv); s = nv;
                    } else s = ns0;
                    base += 15;
                }
            }
            consts32();
            for (int r = 0; r < BN_FULL_ROUNDS / 2 - 1; r++) {
                exp5(base + 12 * l); base += 48;
                ark(half == 0 ? (r + 1) * BN_WIDTH : (BN_FULL_ROUNDS / 2 + 1) * BN_WIDTH + BN_PARTIAL_ROUNDS + r * BN_WIDTH);
                mix(0);
            }
            exp5(base + 12 * l); base += 48;
            if (half == 0) { ark((BN_FULL_ROUNDS / 2) * BN_WIDTH); mix(1); } else mix(0);
        }
        for (int j = 0; j < 4; j++) st[j] = shfl4(s, j);
        cell_off = (uint64_t)(base - out);
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

}  // namespace h2w
