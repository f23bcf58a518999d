// glperm.h — the Goldilocks Poseidon permutation on VALUES, the lanes of one wavefront cooperating (device only; included by coop.h).
//
// A wavefront that has its SIMD to itself issues one instruction every ~4 cycles, a wait state (s_nop) costs the same 4 cycles, and a DEPENDENT
// instruction waits ~9 (tools/ubench/ubench_wave1.hip, profiles/r04_ubench_wave1.txt): a permutation that is one dependent chain is priced by its
// instruction count.  What keeps that count down:
//   * the reduction of a 128-bit product (glq_reduce) is written in assembly: one v_mad_u64_u32 folds the 2^64 word, its carry and the borrow of the
//     2^96 word are turned into ONE 64-bit correction (20 instructions and 3 wait states per product; the compiler's compare-and-select form: 25 and 5);
//   * the four 16-lane rows mirror each other, so x^3 (even rows) and x^4 (odd rows) are ONE product and v_permlane16_swap brings them together;
//   * a row of the small-entry MDS matrix is one hand-scheduled block (24 v_readlane, 24 v_mad_u64_u32, one reduction), with the next round's constant
//     layer in its sums;
//   * the partial rounds keep one accumulator per round instead of forming a row sum per round (glptab.h): a round is the S-box of one value and one
//     multiply-add per lane;
//   * the stretch from the fourth S-box layer to the partial rounds' lanes is ONE linear layer whose entries the host composed (glptab.h W): 36 products
//     of 22-bit limbs into two sums, one reduction.
// Values are ANY 64-bit representatives (x mod p for some x < 2^64) between operations; the permutation's output is canonical.
#pragma once

namespace h2w {

// One scheduling fence: the hazard recogniser does not look inside an asm block, so an asm result that the NEXT instruction reads through v_readlane or
// DPP needs its wait states spelled out (gfx940 family: 1 for v_readlane, 2 for DPP).  A lane swap (v_permlane16/32_swap of a value with itself) needs
// none: its operands are tied, one of them is always a fresh copy of the value, and the compiler itself puts the swap's two wait states behind that copy.
template <int N = 1> __device__ __forceinline__ void glq_lane_fence(uint64_t &v) { asm volatile("s_nop %1" : "+v"(v) : "n"(N)); }      // (tied to the value: stays between its producer and its readers)

// lo + p2 2^64 + p3 2^96 (mod p), 2^64 = 2^32 - 1 = eps, 2^96 = -1:  u = lo + p2 eps (carry c), v = u - p3 (borrow b), r = v + (c - b) eps.
// r does not wrap: c and not b: v + 2^64 is the true sum <= 2^65 - 2^33, so v + eps < 2^64;  b and not c: v - 2^64 > -2^32, so v - eps > 0.
// Scratch registers (caller-saved in the AMDGPU calling convention): v[48:55], s[20:21].
__device__ __forceinline__ uint64_t glq_reduce(uint64_t lo, uint32_t p2, uint32_t p3) {
    uint64_t r;
    asm("v_mad_u64_u32 v[48:49], s[20:21], %[p2], -1, %[lo]\n\t"
        "v_sub_co_u32_e32 v50, vcc, v48, %[p3]\n\t"
        "s_nop 0\n\t"
        "v_cndmask_b32_e64 v52, 0, -1, s[20:21]\n\t"           // c ? -1 : 0
        "v_subbrev_co_u32_e32 v51, vcc, 0, v49, vcc\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e64 v53, 0, -1, vcc\n\t"                // b ? -1 : 0
        "v_cndmask_b32_e64 v55, v53, 0, s[20:21]\n\t"          // high word of (c - b) eps: -1 for b and not c
        "v_sub_u32_e32 v54, v52, v53\n\t"                      // low word: -1 for c alone, +1 for b alone
        "v_lshl_add_u64 %[r], v[50:51], 0, v[54:55]"
        : [r] "=v"(r) : [lo] "v"(lo), [p2] "v"(p2), [p3] "v"(p3)
        : "vcc", "s20", "s21", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
    return r;
}
// lo + h0 2^32 + h1 2^64 with lo + h1 eps < 2^64 (the sums of a small-entry MDS row, the limb sums of a partial round): A = lo + h1 eps, then h0 into the
// high word, its carry = one eps (A + h0 2^32 - 2^64 <= 2^64 - 2^32 - 1: no second wrap)
__device__ __forceinline__ uint64_t glq_reduce96(uint64_t lo, uint32_t h0, uint32_t h1) {
    uint64_t r;
    asm("v_mad_u64_u32 v[48:49], s[20:21], %[h1], -1, %[lo]\n\t"
        "v_mov_b32_e32 v51, 0\n\t"
        "v_add_co_u32_e32 v49, vcc, v49, %[h0]\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e64 v50, 0, -1, vcc\n\t"
        "v_lshl_add_u64 %[r], v[48:49], 0, v[50:51]"
        : [r] "=v"(r) : [lo] "v"(lo), [h0] "v"(h0), [h1] "v"(h1) : "vcc", "s20", "s21", "v48", "v49", "v50", "v51");
    return r;
}
__device__ __forceinline__ uint64_t glq_mul(uint64_t a, uint64_t b) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const uint64_t t0 = (uint64_t)a0 * b0;
    const uint64_t t1 = (uint64_t)a0 * b1 + (t0 >> 32);
    const uint64_t t2 = (uint64_t)a1 * b0 + (uint32_t)t1;
    const uint64_t t3 = (uint64_t)a1 * b1 + (t1 >> 32) + (t2 >> 32);                  // <= (2^32-1)^2 + 2 (2^32-1) = 2^64 - 1
    return glq_reduce((t2 << 32) | (uint32_t)t0, (uint32_t)t3, (uint32_t)(t3 >> 32));
}
__device__ __forceinline__ uint64_t glq_muladd(uint64_t a, uint64_t b, uint64_t c) {      // a b + c < 2^128
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const uint64_t t0 = (uint64_t)a0 * b0 + (uint32_t)c;
    const uint64_t t1 = (uint64_t)a0 * b1 + (t0 >> 32) + (c >> 32);
    const uint64_t t2 = (uint64_t)a1 * b0 + (uint32_t)t1;
    const uint64_t t3 = (uint64_t)a1 * b1 + (t1 >> 32) + (t2 >> 32);
    return glq_reduce((t2 << 32) | (uint32_t)t0, (uint32_t)t3, (uint32_t)(t3 >> 32));
}
__device__ __forceinline__ uint64_t glq_add(uint64_t a, uint64_t b) {      // one of them canonical: the sum wraps at most once, and then r + eps does not wrap again
    uint64_t r = a + b; if (r < a) r += GL_EPS; return r;
}

// rows of v: [v0 v1 v2 v3] -> even = [v0 v0 v2 v2], odd = [v1 v1 v3 v3]   (v_permlane16_swap: odd rows of the first operand <-> even rows of the second)
__device__ __forceinline__ void glq_pair_rows(uint64_t v, uint64_t &even, uint64_t &odd) {
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(v >> 32), (unsigned)(v >> 32), false, false);
    even = ((uint64_t)hi[0] << 32) | lo[0]; odd = ((uint64_t)hi[1] << 32) | lo[1];
}
// every row <- row 0
__device__ __forceinline__ uint64_t glq_bcast_row0(uint64_t v) {
    auto lo = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);                 // [v0 v0 v2 v2]
    auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(v >> 32), (unsigned)(v >> 32), false, false);
    lo = __builtin_amdgcn_permlane32_swap(lo[0], lo[0], false, false);                                  // [v0 v0 v0 v0]
    hi = __builtin_amdgcn_permlane32_swap(hi[0], hi[0], false, false);
    return ((uint64_t)hi[0] << 32) | lo[0];
}

// One row of a small-entry MDS matrix times the state on the lanes, plus a constant word: sum_j m_j s_j + next, s_j = lane j's x (v_readlane), as two
// 64-bit sums of 32-bit halves (12 terms below 2^27 2^32 each and a dword: below 2^63) and ONE reduction.  In assembly for its schedule: a v_readlane
// result may be read two instructions later, not sooner - the pairs rotate through four scalar registers, each read behind the next pair's v_readlanes
// (left to the compiler, every pair was followed by two wait states: 24 per row).
__device__ __forceinline__ uint64_t glq_mds_small(uint64_t x, const uint32_t (&m)[SPONGE_WIDTH], uint64_t next) {
    uint64_t r; const uint64_t nlo = (uint32_t)next, nhi = next >> 32;
#define H2W_MDS_STEP(j, sa, sb, jn) \
        "v_mad_u64_u32 v[48:49], vcc, " sa ", %[m" #j "], v[48:49]\n\t" "v_mad_u64_u32 v[50:51], vcc, " sb ", %[m" #j "], v[50:51]\n\t" \
        "v_readlane_b32 " sa ", %[xl], " #jn "\n\t" "v_readlane_b32 " sb ", %[xh], " #jn "\n\t"
    asm("s_nop 0\n\t"
        "v_readlane_b32 s20, %[xl], 0\n\t" "v_readlane_b32 s21, %[xh], 0\n\t" "v_readlane_b32 s22, %[xl], 1\n\t" "v_readlane_b32 s23, %[xh], 1\n\t"
        "v_mad_u64_u32 v[48:49], vcc, s20, %[m0], %[nlo]\n\t" "v_mad_u64_u32 v[50:51], vcc, s21, %[m0], %[nhi]\n\t"
        "v_readlane_b32 s20, %[xl], 2\n\t" "v_readlane_b32 s21, %[xh], 2\n\t"
        H2W_MDS_STEP(1, "s22", "s23", 3) H2W_MDS_STEP(2, "s20", "s21", 4) H2W_MDS_STEP(3, "s22", "s23", 5) H2W_MDS_STEP(4, "s20", "s21", 6)
        H2W_MDS_STEP(5, "s22", "s23", 7) H2W_MDS_STEP(6, "s20", "s21", 8) H2W_MDS_STEP(7, "s22", "s23", 9) H2W_MDS_STEP(8, "s20", "s21", 10)
        H2W_MDS_STEP(9, "s22", "s23", 11)
        "v_mad_u64_u32 v[48:49], vcc, s20, %[m10], v[48:49]\n\t" "v_mad_u64_u32 v[50:51], vcc, s21, %[m10], v[50:51]\n\t"
        "v_mad_u64_u32 v[48:49], vcc, s22, %[m11], v[48:49]\n\t" "v_mad_u64_u32 v[50:51], vcc, s23, %[m11], v[50:51]\n\t"
        // glq_reduce96(lo = v[48:49], h0 = v50, h1 = v51)
        "v_mad_u64_u32 v[52:53], s[20:21], v51, -1, v[48:49]\n\t"
        "v_mov_b32_e32 v55, 0\n\t"
        "v_add_co_u32_e32 v53, vcc, v53, v50\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e64 v54, 0, -1, vcc\n\t"
        "v_lshl_add_u64 %[r], v[52:53], 0, v[54:55]"
        : [r] "=v"(r)
        : [xl] "v"((uint32_t)x), [xh] "v"((uint32_t)(x >> 32)), [nlo] "v"(nlo), [nhi] "v"(nhi), [m0] "v"(m[0]), [m1] "v"(m[1]), [m2] "v"(m[2]), [m3] "v"(m[3]), [m4] "v"(m[4]),
          [m5] "v"(m[5]), [m6] "v"(m[6]), [m7] "v"(m[7]), [m8] "v"(m[8]), [m9] "v"(m[9]), [m10] "v"(m[10]), [m11] "v"(m[11])
        : "vcc", "s20", "s21", "s22", "s23", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
#undef H2W_MDS_STEP
    return r;
}

// Twelve terms of the linear layer between the fourth S-box layer and the partial rounds (glptab.h W): a0 += lo32(w_i) y_i, a1 += hi32(w_i) y_i, where y_i is
// lane i's 22-bit limb (v_readlane; three scalar registers in rotation: a v_readlane result is read three instructions later) and w_i this lane's
// entries.  36 such terms stay below 2^59.2: no reduction between them.
__device__ __forceinline__ void glq_dense12(uint32_t limb, const uint32_t (&wl)[SPONGE_WIDTH], const uint32_t (&wh)[SPONGE_WIDTH], uint64_t &a0, uint64_t &a1) {
#define H2W_D12(i, sr) "v_mad_u64_u32 %[a0], vcc, " sr ", %[l" #i "], %[a0]\n\t" "v_mad_u64_u32 %[a1], vcc, " sr ", %[h" #i "], %[a1]\n\t"
#define H2W_D12R(i, sr, n) H2W_D12(i, sr) "v_readlane_b32 " sr ", %[y], " #n "\n\t"
    asm("s_nop 0\n\t"
        "v_readlane_b32 s20, %[y], 0\n\t" "v_readlane_b32 s21, %[y], 1\n\t" "v_readlane_b32 s22, %[y], 2\n\t"
        H2W_D12R(0, "s20", 3) H2W_D12R(1, "s21", 4) H2W_D12R(2, "s22", 5) H2W_D12R(3, "s20", 6) H2W_D12R(4, "s21", 7) H2W_D12R(5, "s22", 8)
        H2W_D12R(6, "s20", 9) H2W_D12R(7, "s21", 10) H2W_D12R(8, "s22", 11) H2W_D12(9, "s20") H2W_D12(10, "s21") H2W_D12(11, "s22")
        : [a0] "+v"(a0), [a1] "+v"(a1)
        : [y] "v"(limb), [l0] "v"(wl[0]), [l1] "v"(wl[1]), [l2] "v"(wl[2]), [l3] "v"(wl[3]), [l4] "v"(wl[4]), [l5] "v"(wl[5]), [l6] "v"(wl[6]), [l7] "v"(wl[7]), [l8] "v"(wl[8]),
          [l9] "v"(wl[9]), [l10] "v"(wl[10]), [l11] "v"(wl[11]), [h0] "v"(wh[0]), [h1] "v"(wh[1]), [h2] "v"(wh[2]), [h3] "v"(wh[3]), [h4] "v"(wh[4]), [h5] "v"(wh[5]), [h6] "v"(wh[6]),
          [h7] "v"(wh[7]), [h8] "v"(wh[8]), [h9] "v"(wh[9]), [h10] "v"(wh[10]), [h11] "v"(wh[11])
        : "vcc", "s20", "s21", "s22");
#undef H2W_D12R
#undef H2W_D12
}

// Goldilocks Poseidon (plonky2's fast form, hash/poseidon/permutation.rs:216-284) on VALUES.  Lane l of EVERY 16-lane row holds state element l & 15
// (l & 15 < 12; the callers keep the four rows alike: coop.h sponge_challenge, coop_poseidon_permute) and returns its element of the output on every row
// (lanes 12..15 of a row compute along on element 11 and are ignored).
// K: the constant block in LDS, M: the dense MDS rows, X: the derived tables of the partial rounds (stage_glp_consts<true>).
// list_at / list_word: a word this lane leaves in memory on the way in (the strand's permutation list, coop.h sponge_permute; null: none) - issued here, it
// has the whole permutation to complete in; issued by the caller, the entry of this function would wait for it.
__device__ __noinline__ uint64_t glp_permute_lanes(uint64_t x, lds64_t *K, lds64_t *M, lds64_t *X, int l, bool small, uint64_t *list_at = nullptr, uint64_t list_word = 0) {
    if (list_at) H2W_GSTORE64(reinterpret_cast<unsigned long long *>(list_at), list_word);
#ifdef H2W_EXP_GLP_STUB      // experiment (tools/experiments/variant.sh): what the values phase costs WITHOUT its permutations - the values are garbage
    return x + K[KO_ARC + (l & 7)];
#endif
    const int l15 = l & 15, lc = l15 < SPONGE_WIDTH ? l15 : SPONGE_WIDTH - 1;
    const bool odd_row = (l >> 4) & 1;
    // this lane's row of the dense MDS matrix sits in registers for all eight full rounds (small entries: one dword each); every round's table words
    // are read a round ahead (an LDS read is ~110 cycles for a wavefront with nothing else to run)
    uint32_t mrow[SPONGE_WIDTH];
    if (small) {
#pragma unroll
        for (int j = 0; j < SPONGE_WIDTH; j++) mrow[j] = (uint32_t)M[lc * SPONGE_WIDTH + j];
    }
    // x^7: x^2, then x^3 on the even rows and x^4 on the odd ones in one product, then their product on every row
    auto sbox = [&](uint64_t v) {
        const uint64_t v2 = glq_mul(v, v); uint64_t t = glq_mul(v2, odd_row ? v2 : v);
        uint64_t e, o; glq_pair_rows(t, e, o);
        return glq_mul(e, o);
    };
    // S-box layer and MDS layer of one full round, and the constant layer of what follows it (next: canonical) in the same sums
    auto full_round = [&](uint64_t next) {
        x = sbox(x);
        if (small) x = glq_mds_small(x, mrow, next);
        else {
            uint64_t acc = next;
            glq_lane_fence<0>(x);
#pragma unroll
            for (int j = 0; j < SPONGE_WIDTH; j++) acc = glq_muladd(M[lc * SPONGE_WIDTH + j], readlane64(x, j), acc);
            x = acc;
        }
    };
    uint64_t arc_n = K[KO_ARC + SPONGE_WIDTH + lc];
    x = glq_add(x, K[KO_ARC + lc]);
#pragma unroll 1
    for (int i = 0; i < HALF_N_FULL_ROUNDS - 1; i++) {
        const uint64_t next = arc_n;
        arc_n = K[KO_ARC + SPONGE_WIDTH * (i + 2 < HALF_N_FULL_ROUNDS ? i + 2 : i + 1) + lc];
        full_round(next);
    }
    x = sbox(x);      // the fourth full round's S-box layer; its MDS layer is part of what follows
    // ---- the partial rounds, their row sums unrolled.  Round k turns s0 into a_k = s0^7 + c_k, then s0 <- m00 a_k + sum_i w_hat[k][i] s_i and
    // s_i <- s_i + v[k][i] a_k: every s_i is its value at the start plus a combination of the a's so far, and so is every row sum.  Lane 16 + j keeps
    // A_j = (row sum of round j over the start values) + sum_{k < j} C[k][j] a_k, with C[j][j] = m00 its last term: after round j it IS the next s0.
    // So a round is the S-box of ONE value - computed by every lane, x^3 on the even rows and x^4 on the odd ones; the c_k are in the sums from the start
    // (glptab.h Q) - and ONE multiply-add on every
    // lane (lanes 1..11: s_i += v[k][i] a_k, lanes 16..37: A_j += C[k][j] a_k); no row sum, no sparse-row products.  The start values and the row sums
    // over them come out of ONE linear layer on the fourth S-box layer's outputs (glptab.h W).
    const bool st_lane = l >= 1 && l < SPONGE_WIDTH, acc_lane = l >= 16 && l < 16 + N_PARTIAL_ROUNDS;
    lds64_t *t_round = st_lane ? K + KO_VS + (l - 1) : acc_lane ? X + XO_C + (l - 16) : X + XO_C + N_PARTIAL_ROUNDS;      // (C[1][0] = 0: the lanes with no part in it)
    const int t_step = st_lane ? 11 : acc_lane ? N_PARTIAL_ROUNDS : 0;
    // the linear layer onto the lanes of the partial rounds (glptab.h W: MDS layer, first constants, initial matrix, every round's row sum over the start
    // values and the round constants' share, composed): 36 = 12 x 3 limbs of 22 bits, two sums, one reduction
    uint64_t acc;
    {
        const int slot = glp_slot_of_lane(l);
        lds64_t *wp = X + XO_W + slot;
        const uint64_t w0 = X[XO_W0 + slot];
        uint64_t a0 = (uint32_t)w0, a1 = w0 >> 32;
        const uint32_t y0 = (uint32_t)x & 0x3FFFFF, y1 = (uint32_t)(x >> 22) & 0x3FFFFF, y2 = (uint32_t)(x >> 44);
#pragma unroll
        for (int t = 0; t < 3; t++) {
            uint32_t wl[SPONGE_WIDTH], wh[SPONGE_WIDTH];
#pragma unroll
            for (int i = 0; i < SPONGE_WIDTH; i++) { const uint64_t w = wp[(SPONGE_WIDTH * t + i) * GLP_SLOTS]; wl[i] = (uint32_t)w; wh[i] = (uint32_t)(w >> 32); }
            glq_dense12(t == 0 ? y0 : t == 1 ? y1 : y2, wl, wh, a0, a1);
        }
        acc = glq_reduce96(a0, (uint32_t)a1, (uint32_t)(a1 >> 32));
    }
    glq_lane_fence<0>(acc);
    uint64_t s0 = readlane64(acc, 0);
    uint64_t tk_n = t_round[0];
#pragma unroll 2
    for (int k = 0; k < N_PARTIAL_ROUNDS; k++) {
        const uint64_t tk = tk_n;
        { const int kn = k + 1 < N_PARTIAL_ROUNDS ? k + 1 : k; tk_n = t_round[kn * t_step]; }
        asm("" : "+v"(s0));      // (a lane's copy, not a scalar: on the scalar unit a 64 x 64 product is ~35 instructions)
        const uint64_t s2 = glq_mul(s0, s0); uint64_t t = glq_mul(s2, odd_row ? s2 : s0);
        uint64_t e, o; glq_pair_rows(t, e, o);
        const uint64_t a = glq_mul(e, o);
        acc = glq_muladd(tk, a, acc);
        glq_lane_fence<0>(acc);
        s0 = readlane64(acc, 16 + k);
    }
    x = l == 0 ? s0 : acc;
    x = glq_bcast_row0(x);
    x = glq_add(x, K[KO_ARC + SPONGE_WIDTH * (HALF_N_FULL_ROUNDS + N_PARTIAL_ROUNDS) + lc]);
    arc_n = K[KO_ARC + SPONGE_WIDTH * (HALF_N_FULL_ROUNDS + N_PARTIAL_ROUNDS + 1) + lc];
#pragma unroll 1
    for (int i = 0; i < HALF_N_FULL_ROUNDS; i++) {
        const uint64_t next = i + 1 < HALF_N_FULL_ROUNDS ? arc_n : 0;
        { const int rn = HALF_N_FULL_ROUNDS + N_PARTIAL_ROUNDS + (i + 2 < HALF_N_FULL_ROUNDS ? i + 2 : i); arc_n = K[KO_ARC + SPONGE_WIDTH * rn + lc]; }
        full_round(next);
    }
    return x >= GL_P ? x - GL_P : x;
}

}      // namespace h2w
