// field.h — Goldilocks (u64) and BN254 Fr (4x64) arithmetic, host + device (gfx950).
//
// Goldilocks p = 2^64 - 2^32 + 1 (plonky2 GoldilocksField).  BN254 Fr per SURVEY App. A.
// Wide-integer work only: 64-bit integer MADs (v_mad_u64_u32 chains), no MFMA.
#pragma once
#include <stdint.h>
#include "../../include/h2w.h"

#if defined(__HIPCC__)
#define HD __host__ __device__ __forceinline__
#define HDN __host__ __device__
#define HF __host__ __device__ inline
// HNI: out-of-line by default (the gadget stack is instantiated over five backends; code size).  A translation unit that
// defines H2W_FLATTEN_CHIPS gets them inlinable: its kernel is then flattened so that the sink state lives in registers - every
// AMDGPU function call starts with s_waitcnt vmcnt(0) and reaches its object through flat loads (see glue.hip).
// HNI is for members of templates over a BACKEND only: the flattened unit instantiates backends no other unit does (glue.hip), so no inline
// function exists with two different definitions.  A non-template function that must stay out of line uses HOL, the same in every unit.
#if defined(H2W_FLATTEN_CHIPS)
#define HNI __host__ __device__ inline
#else
#define HNI __host__ __device__ __attribute__((noinline))
#endif
#define HOL __host__ __device__ __attribute__((noinline))
#else
#define HD inline
#define HDN
#define HF inline
#define HNI __attribute__((noinline))
#define HOL __attribute__((noinline))
#endif

namespace h2w {

typedef unsigned __int128 u128;

// Explicit global-address-space accesses.  Pointers that travel through structs / noinline calls lose their address
// space and hipcc falls back to flat_load / flat_store, which count on BOTH vmcnt and lgkmcnt: every dependent flat
// load then waits for all outstanding record stores (s_waitcnt vmcnt(0) lgkmcnt(0)) — ~1-2 us per exchange.  These
// helpers emit global_load / global_store.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(1))) unsigned long long gu64_t;
#define H2W_GLOAD64(p) (*(const ::h2w::gu64_t *)(p))
#define H2W_GSTORE64(p, v) (*(::h2w::gu64_t *)(p) = (v))
// a word of a table no kernel writes (constant address space): with a wavefront-uniform address it is a scalar load - outside the in-order
// queue of the vector loads and stores, where a load waits for every record store issued before it
typedef __attribute__((address_space(4))) const unsigned long long cu64_t;
#define H2W_CLOAD64(p) (*(const ::h2w::cu64_t *)(p))
#else
#define H2W_GLOAD64(p) (*(const unsigned long long *)(p))
#define H2W_GSTORE64(p, v) (*(unsigned long long *)(p) = (v))
#define H2W_CLOAD64(p) (*(const unsigned long long *)(p))
#endif
typedef h2w_fr_t fr_t;

constexpr uint64_t GL_P = 0xFFFFFFFF00000001ULL;
constexpr uint64_t GL_EPS = 0xFFFFFFFFULL;       // 2^64 mod p = 2^32 - 1
constexpr uint64_t GL_NEG_ONE = GL_P - 1;

// ---------------------------------------------------------------- Goldilocks
// x (u128) mod p using 2^64 = 2^32 - 1, 2^96 = -1 (mod p)
HD uint64_t gl_reduce128(u128 x) {
    uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    uint64_t hh = hi >> 32, hl = hi & GL_EPS;
    uint64_t t0 = lo - hh;
    if (lo < hh) t0 -= GL_EPS;                     // borrow: subtract 2^64 = eps (mod p) -> add p... (wraps) 
    uint64_t t1 = hl * GL_EPS;                     // < 2^64
    uint64_t r = t0 + t1;
    if (r < t1) r += GL_EPS;                       // carry: 2^64 = eps
    if (r >= GL_P) r -= GL_P;
    return r;
}
HD uint64_t gl_add(uint64_t a, uint64_t b) { return gl_reduce128((u128)a + b); }
HD uint64_t gl_sub(uint64_t a, uint64_t b) { return a >= b ? a - b : a + (GL_P - b); }
HD uint64_t gl_mul(uint64_t a, uint64_t b) { return gl_reduce128((u128)a * b); }
HD uint64_t gl_muladd(uint64_t a, uint64_t b, uint64_t c) { return gl_reduce128((u128)a * b + c); }
HD uint64_t gl_exp(uint64_t a, uint64_t e) {
    uint64_t acc = 1;
    while (e) { if (e & 1) acc = gl_mul(acc, a); a = gl_mul(a, a); e >>= 1; }
    return acc;
}
HD uint64_t gl_inv(uint64_t a) { return gl_exp(a, GL_P - 2); }
// exact quotient of v by p when v = A*B + C with A,B,C canonical (q < p < 2^64):
// q = (lo64(v) - r) * p^{-1} mod 2^64, p^{-1} = 1 + 2^32 (mod 2^64)
HD void gl_divmod128(u128 v, uint64_t &q, uint64_t &r) {
    r = gl_reduce128(v);
    uint64_t d = (uint64_t)v - r;
    q = d + (d << 32);
}
// plonky2: MULTIPLICATIVE_GROUP_GENERATOR = 7, TWO_ADICITY = 32, POWER_OF_TWO_GENERATOR = 7^((p-1)/2^32)
HD uint64_t gl_primitive_root_of_unity(int n_log) {
    uint64_t g = gl_exp(7, (GL_P - 1) >> 32);
    for (int i = 0; i < 32 - n_log; i++) g = gl_mul(g, g);
    return g;
}
struct gle_t { uint64_t c[2]; };
HD gle_t gle_mul(gle_t a, gle_t b) {
    gle_t r;
    r.c[0] = gl_add(gl_mul(a.c[0], b.c[0]), gl_mul(7, gl_mul(a.c[1], b.c[1])));
    r.c[1] = gl_add(gl_mul(a.c[0], b.c[1]), gl_mul(a.c[1], b.c[0]));
    return r;
}
HD gle_t gle_inv(gle_t a) {
    uint64_t n = gl_sub(gl_mul(a.c[0], a.c[0]), gl_mul(7, gl_mul(a.c[1], a.c[1])));
    uint64_t ni = gl_inv(n);
    gle_t r; r.c[0] = gl_mul(a.c[0], ni); r.c[1] = gl_mul(gl_sub(0, a.c[1]), ni);
    return r;
}

// ---------------------------------------------------------------- BN254 Fr
// r = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
#define H2W_FR_M0 0x43e1f593f0000001ULL
#define H2W_FR_M1 0x2833e84879b97091ULL
#define H2W_FR_M2 0xb85045b68181585dULL
#define H2W_FR_M3 0x30644e72e131a029ULL
HD uint64_t fr_mod_limb(int i) { return i == 0 ? H2W_FR_M0 : i == 1 ? H2W_FR_M1 : i == 2 ? H2W_FR_M2 : H2W_FR_M3; }

struct FrParams { fr_t r2; uint64_t ninv; };   // R^2 mod r (R = 2^261), -r^{-1} mod 2^64 (derived at init, never typed)

HD fr_t fr_zero() { fr_t z; z.l[0] = z.l[1] = z.l[2] = z.l[3] = 0; return z; }
HD fr_t fr_from_u64(uint64_t x) { fr_t z; z.l[0] = x; z.l[1] = z.l[2] = z.l[3] = 0; return z; }
HD fr_t fr_from_u128(u128 x) { fr_t z; z.l[0] = (uint64_t)x; z.l[1] = (uint64_t)(x >> 64); z.l[2] = z.l[3] = 0; return z; }
HD bool fr_is_zero(const fr_t &a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
HD bool fr_eq(const fr_t &a, const fr_t &b) { return a.l[0] == b.l[0] && a.l[1] == b.l[1] && a.l[2] == b.l[2] && a.l[3] == b.l[3]; }
HD bool fr_geq_mod(const fr_t &a) {
    if (a.l[3] != H2W_FR_M3) return a.l[3] > H2W_FR_M3;
    if (a.l[2] != H2W_FR_M2) return a.l[2] > H2W_FR_M2;
    if (a.l[1] != H2W_FR_M1) return a.l[1] > H2W_FR_M1;
    return a.l[0] >= H2W_FR_M0;
}
HD fr_t fr_sub_mod_raw(const fr_t &a) {  // a - r
    fr_t r; u128 t; uint64_t bw = 0;
    t = (u128)a.l[0] - H2W_FR_M0 - bw; r.l[0] = (uint64_t)t; bw = (uint64_t)(t >> 64) & 1;
    t = (u128)a.l[1] - H2W_FR_M1 - bw; r.l[1] = (uint64_t)t; bw = (uint64_t)(t >> 64) & 1;
    t = (u128)a.l[2] - H2W_FR_M2 - bw; r.l[2] = (uint64_t)t; bw = (uint64_t)(t >> 64) & 1;
    t = (u128)a.l[3] - H2W_FR_M3 - bw; r.l[3] = (uint64_t)t;
    return r;
}
HD fr_t fr_add(const fr_t &a, const fr_t &b) {
    fr_t r; u128 t; uint64_t c = 0;
    for (int i = 0; i < 4; i++) { t = (u128)a.l[i] + b.l[i] + c; r.l[i] = (uint64_t)t; c = (uint64_t)(t >> 64); }
    if (fr_geq_mod(r)) r = fr_sub_mod_raw(r);
    return r;
}
HD fr_t fr_sub(const fr_t &a, const fr_t &b) {
    fr_t r; u128 t; uint64_t bw = 0;
    for (int i = 0; i < 4; i++) { t = (u128)a.l[i] - b.l[i] - bw; r.l[i] = (uint64_t)t; bw = (uint64_t)(t >> 64) & 1; }
    if (bw) { uint64_t c = 0; for (int i = 0; i < 4; i++) { t = (u128)r.l[i] + fr_mod_limb(i) + c; r.l[i] = (uint64_t)t; c = (uint64_t)(t >> 64); } }
    return r;
}
HD fr_t fr_neg(const fr_t &a) { return fr_sub(fr_zero(), a); }
// Montgomery product a*b*R^-1 mod r with R = 2^261, on nine 29-bit limbs.
// Product scanning: a column is at most 18 products < 2^58 plus a carry, so it fits a 64-bit accumulator with NO
// carry flags at all: every step is one v_mad_u64_u32 into a 64-bit register (full rate on gfx950), columns are
// closed with a shift.  ~330 VALU ops and wide ILP, against ~770 for 4x64-bit-limb CIOS (carry chains through VCC,
// register-pair shuffling, hazard nops).  Inputs/outputs are canonical 4x64-bit limbs (cells are stored that way).
// r in radix 2^29 (derived from H2W_FR_M*, checked by tests): 
#define H2W_R29_0 0x10000001u
#define H2W_R29_1 0x1f0fac9fu
#define H2W_R29_2 0x0e5c2450u
#define H2W_R29_3 0x07d090f3u
#define H2W_R29_4 0x1585d283u
#define H2W_R29_5 0x02db40c0u
#define H2W_R29_6 0x00a6e141u
#define H2W_R29_7 0x0e5c2634u
#define H2W_R29_8 0x0030644eu
constexpr int FR_MONT_BITS = 261;
HD fr_t fr_mont_mul(const fr_t &A, const fr_t &B, uint64_t ninv) {
    const uint32_t MASK = (1u << 29) - 1;
    const uint32_t N[9] = {H2W_R29_0, H2W_R29_1, H2W_R29_2, H2W_R29_3, H2W_R29_4, H2W_R29_5, H2W_R29_6, H2W_R29_7, H2W_R29_8};
    const uint32_t ninv29 = (uint32_t)ninv & MASK;            // -r^-1 mod 2^29
    uint32_t a[9], b[9], m[9], t[9];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < 9; j++) {
        const int lo = 29 * j, w = lo >> 6, sh = lo & 63;
        uint64_t xa = A.l[w] >> sh, xb = B.l[w] >> sh;
        if (sh > 35 && w < 3) { xa |= A.l[w + 1] << (64 - sh); xb |= B.l[w + 1] << (64 - sh); }
        a[j] = (uint32_t)xa & MASK; b[j] = (uint32_t)xb & MASK;
    }
    uint64_t acc = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int k = 0; k < 9; k++) {
        uint64_t e = 0;                                        // second accumulator: halves the dependent-mad chain
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int i = 0; i <= k; i++) { if (i & 1) e += (uint64_t)a[i] * b[k - i]; else acc += (uint64_t)a[i] * b[k - i]; }
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int i = 0; i < k; i++) { if (i & 1) acc += (uint64_t)m[i] * N[k - i]; else e += (uint64_t)m[i] * N[k - i]; }
        acc += e;
        m[k] = ((uint32_t)acc * ninv29) & MASK;
        acc += (uint64_t)m[k] * N[0];
        acc >>= 29;
    }
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int k = 9; k < 17; k++) {
        uint64_t e = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int i = k - 8; i < 9; i++) { if (i & 1) e += (uint64_t)a[i] * b[k - i]; else acc += (uint64_t)a[i] * b[k - i]; }
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int i = k - 8; i < 9; i++) { if (i & 1) acc += (uint64_t)m[i] * N[k - i]; else e += (uint64_t)m[i] * N[k - i]; }
        acc += e;
        t[k - 9] = (uint32_t)acc & MASK; acc >>= 29;
    }
    t[8] = (uint32_t)acc;
    fr_t r;
    r.l[0] = (uint64_t)t[0] | ((uint64_t)t[1] << 29) | ((uint64_t)t[2] << 58);
    r.l[1] = ((uint64_t)t[2] >> 6) | ((uint64_t)t[3] << 23) | ((uint64_t)t[4] << 52);
    r.l[2] = ((uint64_t)t[4] >> 12) | ((uint64_t)t[5] << 17) | ((uint64_t)t[6] << 46);
    r.l[3] = ((uint64_t)t[6] >> 18) | ((uint64_t)t[7] << 11) | ((uint64_t)t[8] << 40);
    if (fr_geq_mod(r)) r = fr_sub_mod_raw(r);
    return r;
}
// The same product on values that STAY in nine 29-bit limbs and are never reduced below r ("lazy" form; the values pass of the PoseidonBN254
// Merkle paths, coop.h bn_values): no unpacking, no packing, no conditional subtraction.  R = 2^261 leaves 7 bits above r (2^253.6):
//   fr9_mont(a, b) = (a b + m r) / R < a b / R + r,
// so the product of a value below 62 r (the worst the values pass forms: a state element after 56 lazy column updates) with a table constant
// (< r) is below 1.4 r, and squares of anything below 6.2 r are below 1.3 r: every value stays below 2^261 without a single reduction
// (the bounds are walked through at bn_values).  Limbs: t[0..7] < 2^29 after fr9_norm or fr9_mont ("normalised"), t[8] takes the rest.
// A limb-wise sum of k normalised values has limbs below k 2^29: fr9_mont takes ONE operand with limbs below 2^30 (a sum of two) when the
// other is normalised (column sums: 9 (2^30 2^29) + 9 2^58 + carry < 2^63); anything wider goes through fr9_norm first.
struct fr9_t { uint32_t t[9]; };
HD fr9_t fr9_from(const fr_t &A) {
    const uint32_t MASK = (1u << 29) - 1; fr9_t a;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < 9; j++) {
        const int lo = 29 * j, w = lo >> 6, sh = lo & 63;
        uint64_t xa = A.l[w] >> sh;
        if (sh > 35 && w < 3) xa |= A.l[w + 1] << (64 - sh);
        a.t[j] = j < 8 ? ((uint32_t)xa & MASK) : (uint32_t)xa;
    }
    return a;
}
// normalised limbs of a value below 2^256 -> four 64-bit words (no reduction)
HD fr_t fr9_pack(const fr9_t &a) {
    const uint32_t *t = a.t; fr_t r;
    r.l[0] = (uint64_t)t[0] | ((uint64_t)t[1] << 29) | ((uint64_t)t[2] << 58);
    r.l[1] = ((uint64_t)t[2] >> 6) | ((uint64_t)t[3] << 23) | ((uint64_t)t[4] << 52);
    r.l[2] = ((uint64_t)t[4] >> 12) | ((uint64_t)t[5] << 17) | ((uint64_t)t[6] << 46);
    r.l[3] = ((uint64_t)t[6] >> 18) | ((uint64_t)t[7] << 11) | ((uint64_t)t[8] << 40);
    return r;
}
HD fr9_t fr9_add(const fr9_t &a, const fr9_t &b) {
    fr9_t r;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = 0; i < 9; i++) r.t[i] = a.t[i] + b.t[i];
    return r;
}
HD fr9_t fr9_norm(const fr9_t &a) {
    const uint32_t MASK = (1u << 29) - 1; fr9_t r; uint32_t c = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = 0; i < 8; i++) { const uint32_t v = a.t[i] + c; r.t[i] = v & MASK; c = v >> 29; }
    r.t[8] = a.t[8] + c;
    return r;
}
HD fr9_t fr9_mont(const fr9_t &A, const fr9_t &B, uint32_t ninv29) {
    const uint32_t MASK = (1u << 29) - 1;
    const uint32_t N[9] = {H2W_R29_0, H2W_R29_1, H2W_R29_2, H2W_R29_3, H2W_R29_4, H2W_R29_5, H2W_R29_6, H2W_R29_7, H2W_R29_8};
    const uint32_t *a = A.t, *b = B.t; uint32_t m[9]; fr9_t r;
    uint64_t acc = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int k = 0; k < 9; k++) {
        uint64_t e = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int i = 0; i <= k; i++) { if (i & 1) e += (uint64_t)a[i] * b[k - i]; else acc += (uint64_t)a[i] * b[k - i]; }
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int i = 0; i < k; i++) { if (i & 1) acc += (uint64_t)m[i] * N[k - i]; else e += (uint64_t)m[i] * N[k - i]; }
        acc += e;
        m[k] = ((uint32_t)acc * ninv29) & MASK;
        acc += (uint64_t)m[k] * N[0];
        acc >>= 29;
    }
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int k = 9; k < 17; k++) {
        uint64_t e = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int i = k - 8; i < 9; i++) { if (i & 1) e += (uint64_t)a[i] * b[k - i]; else acc += (uint64_t)a[i] * b[k - i]; }
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int i = k - 8; i < 9; i++) { if (i & 1) acc += (uint64_t)m[i] * N[k - i]; else e += (uint64_t)m[i] * N[k - i]; }
        acc += e;
        r.t[k - 9] = (uint32_t)acc & MASK; acc >>= 29;
    }
    r.t[8] = (uint32_t)acc;
    return r;
}
// canonical a*b mod r (two Montgomery products)
HD fr_t fr_mul(const fr_t &a, const fr_t &b, const FrParams &P) {
    if ((a.l[1] | a.l[2] | a.l[3] | b.l[1] | b.l[2] | b.l[3]) == 0) return fr_from_u128((u128)a.l[0] * b.l[0]);
    fr_t t = fr_mont_mul(a, b, P.ninv);
    return fr_mont_mul(t, P.r2, P.ninv);
}
inline FrParams fr_params_init() {
    FrParams P; uint64_t inv = 1;
    for (int i = 0; i < 6; i++) inv *= 2 - H2W_FR_M0 * inv;
    P.ninv = (uint64_t)0 - inv;
    fr_t x = fr_from_u64(1);
    for (int i = 0; i < 2 * FR_MONT_BITS; i++) x = fr_add(x, x);   // R^2 mod r, R = 2^261
    P.r2 = x;
    return P;
}
HDN inline fr_t fr_pow(const fr_t &a, const fr_t &e, const FrParams &P) {
    fr_t acc = fr_from_u64(1), base = a;
    for (int i = 0; i < 256; i++) {
        if ((e.l[i >> 6] >> (i & 63)) & 1) acc = fr_mul(acc, base, P);
        base = fr_mul(base, base, P);
    }
    return acc;
}
HDN inline fr_t fr_inv(const fr_t &a, const FrParams &P) {
    fr_t e; e.l[0] = H2W_FR_M0 - 2; e.l[1] = H2W_FR_M1; e.l[2] = H2W_FR_M2; e.l[3] = H2W_FR_M3;
    return fr_pow(a, e, P);
}
HD fr_t fr_pow2(int k) { fr_t r = fr_zero(); r.l[k >> 6] = 1ULL << (k & 63); return r; }
HD void g_store_fr(fr_t *p, const fr_t &v) {
    unsigned long long *q = reinterpret_cast<unsigned long long *>(p);
    H2W_GSTORE64(q, v.l[0]); H2W_GSTORE64(q + 1, v.l[1]); H2W_GSTORE64(q + 2, v.l[2]); H2W_GSTORE64(q + 3, v.l[3]);
}
HD fr_t g_load_fr(const fr_t *p) {
    const unsigned long long *q = reinterpret_cast<const unsigned long long *>(p);
    fr_t v; v.l[0] = H2W_GLOAD64(q); v.l[1] = H2W_GLOAD64(q + 1); v.l[2] = H2W_GLOAD64(q + 2); v.l[3] = H2W_GLOAD64(q + 3); return v;
}
HD uint64_t g_load_u64(const uint64_t *p) { return H2W_GLOAD64(p); }
HD uint64_t fr_bits(const fr_t &v, int lo, int width) {
    if (lo >= 256) return 0;
    int w = lo >> 6, sh = lo & 63;
    uint64_t out = v.l[w] >> sh;
    if (sh && w + 1 < 4) out |= v.l[w + 1] << (64 - sh);
    return width >= 64 ? out : out & ((1ULL << width) - 1);
}

}  // namespace h2w
