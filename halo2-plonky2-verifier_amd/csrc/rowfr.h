// rowfr.h — lane-cooperative lazy Montgomery arithmetic in BN254 Fr: ONE 29-bit limb per lane.
//
// The values pass of the PoseidonBN254 Merkle paths (hash/poseidon_bn254/permutation.rs:83-203 under merkle/mod.rs:57-78) is a chain of
// ~300 DEPENDENT modular products per permutation, 18 permutations per path: with a whole product on one lane (field.h fr9_mont: ~330
// instructions, each ~5 cycles of a wavefront that has its SIMD to itself) a path takes 2.5 ms whatever the number of paths.  Here a value
// lives on the 16 lanes of a DPP row - lane k holds limb k (k < 9; lanes 9..15 hold 0), R = 2^261 as in field.h - and a product is ~100
// wavefront instructions:
//   T = a x b       lane k forms column k = sum_i a_i b_(k-i): the operand `a` REPLICATED (register i = limb i on every lane of the row),
//                   the operand b as its eight row shifts (DPP row_shr: 1 instruction each) - nine multiply-adds, no carries;
//   m = T N' mod R  N' = -N^-1 mod 2^261 in full (NOT the usual limb-by-limb m_i: that is a chain of nine broadcasts), the same column form on
//                   the low nine lanes;
//   U = m x N       again; (T + U) / R is exact.
// Columns are 64-bit sums of up to nine 2^60 products; between the three multiplications they are brought back to 30-bit limbs by a LOCAL
// carry step (split a column into bits 0..28 | 29..57 | 58.., hand the upper parts to the next two lanes): no ripple.  The exact division
// needs the carry out of the low half: because that half of T + U is 0 mod 2^261 and its limbs are below 2^31 after the local step, the carry
// out of limb k is ceil(limb / 2^29) whatever came in from below (rowfr_check.cpp walks through it), so it is read off lane 8 alone.
// Values are lazy (any representative below 2^261, limbs at most a few units above 2^29), as in field.h fr9_t.
//
// The four rows of a wavefront hold the four elements of the PoseidonBN254 state of ONE Merkle path (rowperm.h); rows exchange values with
// v_permlane16_swap / v_permlane32_swap (gfx950), a dynamic operand is replicated with nine v_readlane.
//
// The same source runs on the host with the lanes simulated (tests/cpp/rowfr_check.cpp pins the arithmetic, the bounds and the whole
// permutation against field.h and chips.h before any GPU sees it).
#pragma once
#include "field.h"

namespace h2w {
namespace rf {

constexpr uint32_t M29 = (1u << 29) - 1;
constexpr int SBX9_W = 12;      // dwords per S-box value the row-cooperative values pass hands to the emission pass (nine limbs used)

#if defined(__HIP_DEVICE_COMPILE__)
// ------------------------------------------------------------------------------------------------ device: a lane's view
typedef uint32_t V; typedef uint64_t W; typedef bool P;
#define RF_FN __device__ __forceinline__
RF_FN V lane_index() { return threadIdx.x & 63u; }
template <int N> RF_FN V shr(V v) { return (V)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + N, 0xf, 0xf, true); }      // lane k <- lane k - N of its row, 0 where there is none
template <int N> RF_FN V shl(V v) { return (V)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + N, 0xf, 0xf, true); }      // lane k <- lane k + N
template <int N> RF_FN V shr_keep(V old, V v) { return (V)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x110 + N, 0xf, 0xf, false); }      // ... `old` where there is none
RF_FN W mad(V a, V b, W c) { return (W)a * b + c; }
RF_FN W madk(uint32_t k, V b, W c) { return (W)k * b + c; }
RF_FN V lo32(W w) { return (V)w; }
RF_FN V hi32(W w) { return (V)(w >> 32); }
RF_FN V sel(P p, V a, V b) { return p ? a : b; }
RF_FN uint32_t rdlane(V v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
template <int J> RF_FN V quad_bcast(V v) { return (V)__builtin_amdgcn_update_dpp(0, (int)v, J * 0x55, 0xf, 0xf, false); }      // every lane <- lane J of its quad
template <int N, int BANKS> RF_FN V shr_banks(V old, V v) { return (V)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x110 + N, 0xf, BANKS, false); }      // the quads in BANKS <- lane k - N; `old` elsewhere
template <int N, int BANKS> RF_FN V shl_banks(V old, V v) { return (V)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x100 + N, 0xf, BANKS, false); }      // ... <- lane k + N
// lanes `rows` of dst <- the wave-uniform value s (one v_mov under the lanes' mask; a select would cost a move of s into a register first)
RF_FN void mov_rows(V &dst, P rows, uint32_t s) { if (rows) asm volatile("v_mov_b32 %0, %1" : "+v"(dst) : "s"(s)); }
// rows of v: [v0 v1 v2 v3] -> even = [v0 v0 v2 v2], odd = [v1 v1 v3 v3]   (v_permlane16_swap: odd rows of the first operand <-> even rows of the second)
RF_FN void pair_bcast(V v, V &even, V &odd) { auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); even = r[0]; odd = r[1]; }
// halves of v: [lo hi] -> low = [lo lo], high = [hi hi]                    (v_permlane32_swap: upper half of the first operand <-> lower half of the second)
RF_FN void half_bcast(V v, V &low, V &high) { auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); low = r[0]; high = r[1]; }
#else
// ------------------------------------------------------------------------------------------------ host: 64 simulated lanes
#define RF_FN inline
struct P { bool l[64]; };
struct V {
    uint32_t l[64];
    V() { for (int i = 0; i < 64; i++) l[i] = 0; }
    V(uint32_t x) { for (int i = 0; i < 64; i++) l[i] = x; }
};
struct W {
    uint64_t l[64];
    W() { for (int i = 0; i < 64; i++) l[i] = 0; }
    W(uint64_t x) { for (int i = 0; i < 64; i++) l[i] = x; }
    explicit W(const V &v) { for (int i = 0; i < 64; i++) l[i] = v.l[i]; }
};
extern bool g_overflow;      // a 64-bit column sum wrapped (the check reads it)
#define RF_LANES(...) for (int i = 0; i < 64; i++) { __VA_ARGS__; }
inline V operator+(const V &a, const V &b) { V r; RF_LANES(r.l[i] = a.l[i] + b.l[i]; if (r.l[i] < a.l[i]) g_overflow = true) return r; }
inline V operator&(const V &a, const V &b) { V r; RF_LANES(r.l[i] = a.l[i] & b.l[i]) return r; }
inline V operator>>(const V &a, int s) { V r; RF_LANES(r.l[i] = a.l[i] >> s) return r; }
inline W operator+(const W &a, const W &b) { W r; RF_LANES(r.l[i] = a.l[i] + b.l[i]; if (r.l[i] < a.l[i]) g_overflow = true) return r; }
inline W operator>>(const W &a, int s) { W r; RF_LANES(r.l[i] = a.l[i] >> s) return r; }
inline P operator==(const V &a, const V &b) { P r; RF_LANES(r.l[i] = a.l[i] == b.l[i]) return r; }
inline P operator<(const V &a, const V &b) { P r; RF_LANES(r.l[i] = a.l[i] < b.l[i]) return r; }
inline P operator!(const P &a) { P r; RF_LANES(r.l[i] = !a.l[i]) return r; }
inline P operator||(const P &a, const P &b) { P r; RF_LANES(r.l[i] = a.l[i] || b.l[i]) return r; }
inline P operator&&(const P &a, const P &b) { P r; RF_LANES(r.l[i] = a.l[i] && b.l[i]) return r; }
inline V lane_index() { V r; RF_LANES(r.l[i] = (uint32_t)i) return r; }
template <int N> inline V shr(const V &v) { V r; RF_LANES(const int k = i & 15; r.l[i] = k >= N ? v.l[i - N] : 0u) return r; }
template <int N> inline V shl(const V &v) { V r; RF_LANES(const int k = i & 15; r.l[i] = k + N <= 15 ? v.l[i + N] : 0u) return r; }
template <int N> inline V shr_keep(const V &old, const V &v) { V r; RF_LANES(const int k = i & 15; r.l[i] = k >= N ? v.l[i - N] : old.l[i]) return r; }
inline W mad(const V &a, const V &b, const W &c) { W r; RF_LANES(const uint64_t p = (uint64_t)a.l[i] * b.l[i]; r.l[i] = p + c.l[i]; if (r.l[i] < p) g_overflow = true) return r; }
inline W madk(uint32_t k, const V &b, const W &c) { W r; RF_LANES(const uint64_t p = (uint64_t)k * b.l[i]; r.l[i] = p + c.l[i]; if (r.l[i] < p) g_overflow = true) return r; }
inline V lo32(const W &w) { V r; RF_LANES(r.l[i] = (uint32_t)w.l[i]) return r; }
inline V hi32(const W &w) { V r; RF_LANES(r.l[i] = (uint32_t)(w.l[i] >> 32)) return r; }
inline V sel(const P &p, const V &a, const V &b) { V r; RF_LANES(r.l[i] = p.l[i] ? a.l[i] : b.l[i]) return r; }
inline uint32_t rdlane(const V &v, int lane) { return v.l[lane]; }
template <int J> inline V quad_bcast(const V &v) { V r; RF_LANES(r.l[i] = v.l[(i & ~3) + J]) return r; }
template <int N, int BANKS> inline V shr_banks(const V &old, const V &v) { V r; RF_LANES(const int k = i & 15; r.l[i] = ((BANKS >> (k >> 2)) & 1) && k >= N ? v.l[i - N] : old.l[i]) return r; }
template <int N, int BANKS> inline V shl_banks(const V &old, const V &v) { V r; RF_LANES(const int k = i & 15; r.l[i] = ((BANKS >> (k >> 2)) & 1) && k + N <= 15 ? v.l[i + N] : old.l[i]) return r; }
inline void mov_rows(V &dst, const P &rows, uint32_t s) { RF_LANES(if (rows.l[i]) dst.l[i] = s) }
inline void pair_bcast(const V &v, V &even, V &odd) { RF_LANES(const int row = i >> 4, k = i & 15; even.l[i] = v.l[(row & 2) * 16 + k]; odd.l[i] = v.l[((row & 2) + 1) * 16 + k]) }
inline void half_bcast(const V &v, V &low, V &high) { RF_LANES(low.l[i] = v.l[i & 31]; high.l[i] = v.l[32 + (i & 31)]) }
#endif

// wave-uniform constants: the modulus and N' = -N^-1 mod 2^261, nine 29-bit limbs each (built on the host: rowconst_init)
struct RowConst { uint32_t n[9], np[9], r2[9]; };      // r2 = R^2 mod N (to Montgomery form)
// per-lane constants of a wavefront
struct LaneK {
    V k;            // lane within its row
    P row0, row1, row23, low12;
    V m_mask;       // limbs of m = T N' mod 2^261: all bits on lanes 0..7, 29 bits on lane 8, nothing above
    V is8;          // all bits on lane 8 (where the carry out of the low half is read)
};
RF_FN LaneK lane_consts() {
    LaneK L; const V lane = lane_index();
    L.k = lane & V(15u);
    const V row = lane >> 4;
    L.row0 = row == V(0u); L.row1 = row == V(1u); L.row23 = !(row < V(2u)); L.low12 = L.k < V(12u);
    L.m_mask = sel(L.k < V(8u), V(~0u), sel(L.k == V(8u), V(M29), V(0u)));
    L.is8 = sel(L.k == V(8u), V(~0u), V(0u));
    return L;
}

struct A9 { V a[9]; };      // a replicated operand: a[i] = limb i on every lane of the row

// a 64-bit column -> 30-bit limb: its bits 0..28 stay, bits 29..57 go one lane up, the rest two lanes up.  The value sum_k col_k 2^(29 k) is
// unchanged (what leaves lane 15 is dropped: callers that need it take it from `mid` / `top`).
RF_FN V split(const W &c, V &mid, V &top) {
    const V lo = lo32(c) & V(M29);
    mid = lo32(c >> 29) & V(M29);
    top = hi32(c) >> 26;                                   // bits 58..63
    return lo + shr<1>(mid) + shr<2>(top);
}
// limbs up to 2^32 -> limbs below 2^29 + 8 (one local carry step; the value is below 2^261, so nothing leaves lane 8)
RF_FN V tighten(const V &v) { return (v & V(M29)) + shr<1>(v >> 29); }

// Montgomery product a b / R mod N of lazy operands: A replicated, b one limb per lane (0 on lanes 9..15); limbs of both below 2^30 + 2^7
// (a tight value, or a sum of two).  Result: one limb per lane, limbs below 2^29 + 8, value below a b / R + (1 + 2^-20) N.
RF_FN V mont(const A9 &A, const V &b, const RowConst &K, const LaneK &L) {
    // T = a x b: column k on lane k (k <= 15), column 16 on lane 0 of `y`
    const V B1 = shr<1>(b), B2 = shr<2>(b), B3 = shr<3>(b), B4 = shr<4>(b), B5 = shr<5>(b), B6 = shr<6>(b), B7 = shr<7>(b), B8 = shr<8>(b);
    W c0 = mad(A.a[0], b, W(0)), c1 = mad(A.a[1], B1, W(0)), c2 = mad(A.a[2], B2, W(0));
    c0 = mad(A.a[3], B3, c0); c1 = mad(A.a[4], B4, c1); c2 = mad(A.a[5], B5, c2);
    c0 = mad(A.a[6], B6, c0); c1 = mad(A.a[7], B7, c1); c2 = mad(A.a[8], B8, c2);
    const W c = c0 + c1 + c2;
    const W y = mad(A.a[8], shl<8>(b), W(0));
    // m = (T mod R) N' mod R, on the low nine lanes
    V mid, top;
    const V t = split(c, mid, top);
    const V T1 = shr<1>(t), T2 = shr<2>(t), T3 = shr<3>(t), T4 = shr<4>(t), T5 = shr<5>(t), T6 = shr<6>(t), T7 = shr<7>(t), T8 = shr<8>(t);
    W m0 = madk(K.np[0], t, W(0)), m1 = madk(K.np[1], T1, W(0)), m2 = madk(K.np[2], T2, W(0));
    m0 = madk(K.np[3], T3, m0); m1 = madk(K.np[4], T4, m1); m2 = madk(K.np[5], T5, m2);
    m0 = madk(K.np[6], T6, m0); m1 = madk(K.np[7], T7, m1); m2 = madk(K.np[8], T8, m2);
    V mm, mt;
    const V m = split(m0 + m1 + m2, mm, mt) & L.m_mask;
    // d = T + m N: its low nine columns sum to a multiple of R
    const V M1 = shr<1>(m), M2 = shr<2>(m), M3 = shr<3>(m), M4 = shr<4>(m), M5 = shr<5>(m), M6 = shr<6>(m), M7 = shr<7>(m), M8 = shr<8>(m);
    W d0 = madk(K.n[0], m, c0), d1 = madk(K.n[1], M1, c1), d2 = madk(K.n[2], M2, c2);
    d0 = madk(K.n[3], M3, d0); d1 = madk(K.n[4], M4, d1); d2 = madk(K.n[5], M5, d2);
    d0 = madk(K.n[6], M6, d0); d1 = madk(K.n[7], M7, d1); d2 = madk(K.n[8], M8, d2);
    const W d = d0 + d1 + d2;
    const W dy = madk(K.n[8], shl<8>(m), y);                 // column 16 (lane 0): below 2^60
    // (T + m N) / R: columns 9..17 plus the carry out of the low half
    V dm, dt;
    const V f = split(d, dm, dt);
    const V fy = (lo32(dy) & V(M29)) + shl<15>(dm) + shl<14>(dt) + shr<1>(lo32(dy >> 29));      // columns 16, 17 on lanes 0, 1
    const V carry = ((f + V(M29)) >> 29) & L.is8;            // ceil(limb 8 / 2^29): the carry out of the low half, whatever came in from below
    V res = shl<9>(f);                                       // columns 9..15 -> lanes 0..6
    res = shr_keep<7>(res, fy);                              // columns 16, 17 -> lanes 7, 8 (0 above)
    res = res + shl<8>(carry);
    return tighten(res);
}

// the operand `a` of a product from a value held one limb per lane: nine v_readlane (wave-uniform: row `row` of v)
RF_FN void replicate(const V &v, int row, uint32_t out[9]) {
    for (int i = 0; i < 9; i++) out[i] = rdlane(v, 16 * row + i);
}
RF_FN void put_rows(A9 &A, const P &rows, const uint32_t limbs[9]) {
    for (int i = 0; i < 9; i++) mov_rows(A.a[i], rows, limbs[i]);
}
// the same for every row at once, each row its own value: limb i of a row on the lanes i..15 of that row (the lanes below i multiply zeros: mont) and
// limb 8 on lane 0 as well (where mont forms column 16).  DPP only: the lane's quad gets the limb from a quad_perm, the other quads from bank-masked row
// shifts - 27 moves for the four rows of a wavefront (nine v_readlane + nine moves per ROW the other way).
template <int I> RF_FN V bcast_limb(const V &v) {
    const V x = quad_bcast<I & 3>(v);
    if (I < 4) { const V y = shr_banks<4, 0x2>(x, x); return shr_banks<8, 0xc>(y, y); }
    if (I < 8) { const V y = shr_banks<4, 0x4>(x, x); return shr_banks<8, 0x8>(y, x); }
    const V y = shr_banks<4, 0x8>(x, x); return shl_banks<8, 0x1>(y, x);
}
RF_FN A9 replicate_rows(const V &v) {
    A9 A;
    A.a[0] = bcast_limb<0>(v); A.a[1] = bcast_limb<1>(v); A.a[2] = bcast_limb<2>(v); A.a[3] = bcast_limb<3>(v); A.a[4] = bcast_limb<4>(v);
    A.a[5] = bcast_limb<5>(v); A.a[6] = bcast_limb<6>(v); A.a[7] = bcast_limb<7>(v); A.a[8] = bcast_limb<8>(v);
    return A;
}

inline void rowconst_init(RowConst &K, const FrParams &P) {
    // N in 29-bit limbs
    fr_t nn; nn.l[0] = H2W_FR_M0; nn.l[1] = H2W_FR_M1; nn.l[2] = H2W_FR_M2; nn.l[3] = H2W_FR_M3;
    const fr9_t n9 = fr9_from(nn); for (int i = 0; i < 9; i++) K.n[i] = n9.t[i];
    const fr9_t r29 = fr9_from(P.r2); for (int i = 0; i < 9; i++) K.r2[i] = r29.t[i];
    // N' = -N^-1 mod 2^261 by Newton's iteration on nine-limb integers: x <- x (2 - N x), doubling the correct low bits (N is odd: x = 1 is right mod 2)
    auto mul_lo = [](const uint32_t *a, const uint32_t *b, uint32_t *out) {      // a b mod 2^261
        unsigned __int128 acc = 0;
        for (int k = 0; k < 9; k++) { for (int i = 0; i <= k; i++) acc += (unsigned __int128)a[i] * b[k - i]; out[k] = (uint32_t)acc & M29; acc >>= 29; }
    };
    uint32_t x[9] = {1, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < 9; it++) {
        uint32_t nx[9], two_minus[9], nxt[9];
        mul_lo(K.n, x, nx);
        // 2 - nx mod 2^261
        int64_t borrow = 0;
        for (int k = 0; k < 9; k++) { int64_t v = (k == 0 ? 2 : 0) - (int64_t)nx[k] - borrow; borrow = 0; while (v < 0) { v += (int64_t)1 << 29; borrow++; } two_minus[k] = (uint32_t)v; }
        mul_lo(x, two_minus, nxt);
        for (int k = 0; k < 9; k++) x[k] = nxt[k];
    }
    // negate mod 2^261
    int64_t borrow = 0;
    for (int k = 0; k < 9; k++) { int64_t v = -(int64_t)x[k] - borrow; borrow = 0; while (v < 0) { v += (int64_t)1 << 29; borrow++; } K.np[k] = (uint32_t)v; }
}

}  // namespace rf
}  // namespace h2w
