// host_fr.h — BN254 Fr product for the HOST side of the eager boundary (eager.cpp: the values of level-1 calls, field/native.rs:48-92, are computed at
// call time because the reference reads AssignedValue::value()).  field.h's product is shaped for the GPU (nine 29-bit limbs, no carry flags: ~60 ns
// per Montgomery product on a CPU core); this one is plain CIOS on four 64-bit limbs with 128-bit intermediates (R = 2^256, ~15 ns), which is what
// bounds the eager level: a PoseidonBN254 permutation is 784 such calls.
#pragma once
#include "field.h"

namespace h2w {

struct HostFr { fr_t r2; uint64_t ninv; };      // R^2 mod r for R = 2^256; -r^-1 mod 2^64
inline fr_t host_mont64(const fr_t &a, const fr_t &b, uint64_t ninv) {
    typedef unsigned __int128 w128;
    const uint64_t n0 = H2W_FR_M0, n1 = H2W_FR_M1, n2 = H2W_FR_M2, n3 = H2W_FR_M3;
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
#define H2W_CIOS_ROUND(bi) { \
    w128 c = (w128)a.l[0] * bi + t0; const uint64_t lo0 = (uint64_t)c; c >>= 64; \
    c += (w128)a.l[1] * bi + t1; const uint64_t lo1 = (uint64_t)c; c >>= 64; \
    c += (w128)a.l[2] * bi + t2; const uint64_t lo2 = (uint64_t)c; c >>= 64; \
    c += (w128)a.l[3] * bi + t3; const uint64_t lo3 = (uint64_t)c; c >>= 64; \
    const uint64_t lo4 = t4 + (uint64_t)c; \
    const uint64_t m = lo0 * ninv; \
    w128 d = (w128)m * n0 + lo0; d >>= 64; \
    d += (w128)m * n1 + lo1; t0 = (uint64_t)d; d >>= 64; \
    d += (w128)m * n2 + lo2; t1 = (uint64_t)d; d >>= 64; \
    d += (w128)m * n3 + lo3; t2 = (uint64_t)d; d >>= 64; \
    d += lo4; t3 = (uint64_t)d; t4 = (uint64_t)(d >> 64); }
    H2W_CIOS_ROUND(b.l[0]) H2W_CIOS_ROUND(b.l[1]) H2W_CIOS_ROUND(b.l[2]) H2W_CIOS_ROUND(b.l[3])
#undef H2W_CIOS_ROUND
    fr_t r; r.l[0] = t0; r.l[1] = t1; r.l[2] = t2; r.l[3] = t3;
    if (t4 || fr_geq_mod(r)) r = fr_sub_mod_raw(r);
    return r;
}
inline HostFr host_fr_init() {
    HostFr h; uint64_t inv = 1;
    for (int i = 0; i < 6; i++) inv *= 2 - H2W_FR_M0 * inv;
    h.ninv = (uint64_t)0 - inv;
    fr_t x = fr_from_u64(1);
    for (int i = 0; i < 512; i++) x = fr_add(x, x);      // 2^512 mod r
    h.r2 = x;
    return h;
}
// canonical a * b mod r
inline fr_t host_fr_mul(const fr_t &a, const fr_t &b, const HostFr &h) {
    if ((a.l[1] | a.l[2] | a.l[3] | b.l[1] | b.l[2] | b.l[3]) == 0) return fr_from_u128((u128)a.l[0] * b.l[0]);
    return host_mont64(host_mont64(a, b, h.ninv), h.r2, h.ninv);
}

}  // namespace h2w
