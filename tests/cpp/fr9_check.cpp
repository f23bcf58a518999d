// Host check of the lazy nine-limb Montgomery arithmetic (csrc/field.h fr9_t) that the values pass of the PoseidonBN254 Merkle paths runs on:
// products of un-reduced representatives (a + k r, up to the bounds written at coop.h bn_values) agree with the canonical product, stay below
// a b / R + r < 2^261, and come out with normalised limbs; a sum of two normalised values is a valid operand against a normalised one.
// Built and run by tests/test_field_lazy.py (g++; field.h compiles as plain C++ on the host).
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include "field.h"
using namespace h2w;

struct Big { uint64_t w[5]; };                      // 320-bit
static Big big_from9(const fr9_t &a) {               // any limbs (un-normalised too)
    Big r = {{0, 0, 0, 0, 0}};
    for (int i = 8; i >= 0; i--) {
        // r = r * 2^29 + a.t[i]
        uint64_t carry = a.t[i];
        for (int j = 0; j < 5; j++) { const unsigned __int128 t = ((unsigned __int128)r.w[j] << 29) + carry; r.w[j] = (uint64_t)t; carry = (uint64_t)(t >> 64); }
    }
    return r;
}
static Big big_r() { Big r = {{H2W_FR_M0, H2W_FR_M1, H2W_FR_M2, H2W_FR_M3, 0}}; return r; }
static Big big_shl(const Big &a, int k) { Big r = {{0, 0, 0, 0, 0}}; for (int j = 4; j >= 0; j--) { r.w[j] = a.w[j] << k; if (j > 0 && k) r.w[j] |= a.w[j - 1] >> (64 - k); } return r; }
static bool big_geq(const Big &a, const Big &b) { for (int j = 4; j >= 0; j--) if (a.w[j] != b.w[j]) return a.w[j] > b.w[j]; return true; }
static Big big_sub(const Big &a, const Big &b) { Big r; unsigned __int128 bw = 0; for (int j = 0; j < 5; j++) { const unsigned __int128 t = (unsigned __int128)a.w[j] - b.w[j] - (uint64_t)bw; r.w[j] = (uint64_t)t; bw = (t >> 64) & 1; } return r; }
static Big big_add(const Big &a, const Big &b) { Big r; unsigned __int128 c = 0; for (int j = 0; j < 5; j++) { const unsigned __int128 t = (unsigned __int128)a.w[j] + b.w[j] + (uint64_t)c; r.w[j] = (uint64_t)t; c = t >> 64; } return r; }
static fr_t big_mod_r(Big a) { for (int k = 12; k >= 0; k--) { const Big m = big_shl(big_r(), k); if (big_geq(a, m)) a = big_sub(a, m); } fr_t r; for (int j = 0; j < 4; j++) r.l[j] = a.w[j]; if (a.w[4]) { printf("reduction left a fifth word\n"); exit(1); } return r; }
static int big_bits(const Big &a) { for (int j = 4; j >= 0; j--) if (a.w[j]) return 64 * j + 64 - __builtin_clzll(a.w[j]); return 0; }

static uint64_t rng_s = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_s ^= rng_s << 7; rng_s ^= rng_s >> 9; return rng_s * 0x2545F4914F6CDD1Dull; }
static fr_t rnd_fr() { Big b = {{rnd(), rnd(), rnd(), rnd() >> 2, 0}}; return big_mod_r(b); }
// a + k r as normalised limbs
static fr9_t lazy_rep(const fr_t &a, unsigned k) {
    fr9_t x = fr9_from(a); fr_t rr; rr.l[0] = H2W_FR_M0; rr.l[1] = H2W_FR_M1; rr.l[2] = H2W_FR_M2; rr.l[3] = H2W_FR_M3; const fr9_t r9 = fr9_from(rr);
    for (unsigned i = 0; i < k; i++) x = fr9_norm(fr9_add(x, r9));
    return x;
}
static bool normalised(const fr9_t &a) { for (int i = 0; i < 8; i++) if (a.t[i] >> 29) return false; return true; }
static bool eq(const fr_t &a, const fr_t &b) { return a.l[0] == b.l[0] && a.l[1] == b.l[1] && a.l[2] == b.l[2] && a.l[3] == b.l[3]; }

int main() {
    const FrParams P = fr_params_init(); const uint32_t ninv29 = (uint32_t)P.ninv & ((1u << 29) - 1);
    int fails = 0; int max_bits = 0;
    // pack / unpack
    for (int it = 0; it < 1000; it++) { const fr_t a = rnd_fr(); if (!eq(fr9_pack(fr9_from(a)), a)) { printf("pack(from(a)) != a\n"); fails++; } }
    // products of un-reduced representatives: (multiple of r on a, on b) as far as the values pass goes (coop.h bn_values) and beyond
    const unsigned ks[][2] = {{0, 0}, {1, 1}, {6, 6}, {7, 0}, {63, 0}, {63, 1}, {25, 5}, {63, 63}, {100, 1}};
    for (auto &k : ks)
        for (int it = 0; it < 300; it++) {
            const fr_t a = rnd_fr(), b = rnd_fr();
            const fr9_t A = lazy_rep(a, k[0]), B = lazy_rep(b, k[1]);
            const fr9_t Pr = fr9_mont(A, B, ninv29);
            const Big pv = big_from9(Pr);
            if (!normalised(Pr)) { printf("product limbs not normalised (k = %u, %u)\n", k[0], k[1]); fails++; }
            if (!eq(big_mod_r(pv), fr_mont_mul(a, b, P.ninv))) { printf("product differs mod r (k = %u, %u)\n", k[0], k[1]); fails++; }
            // < a b / R + r: with a < (k0 + 1) r, b < (k1 + 1) r and r / R < 2^-7.4 the product is below ((k0 + 1)(k1 + 1) 2^-7.4 + 1) r
            const double bound = ((double)(k[0] + 1) * (k[1] + 1) / 168.9 + 1.0);
            Big lim = big_r(); Big acc = {{0, 0, 0, 0, 0}}; for (int m = 0; m < (int)bound + 1; m++) acc = big_add(acc, lim);
            if (big_geq(pv, acc)) { printf("product above its bound (k = %u, %u)\n", k[0], k[1]); fails++; }
            const int nb = big_bits(pv); if (nb > max_bits) max_bits = nb;
        }
    // a limb-wise sum of two normalised values (limbs < 2^30) against a normalised operand, un-normalised
    for (int it = 0; it < 2000; it++) {
        const fr_t a = rnd_fr(), c = rnd_fr(), b = rnd_fr();
        const fr9_t S = fr9_add(lazy_rep(a, it % 64), fr9_from(c));            // not normalised
        const fr9_t Pr = fr9_mont(S, fr9_from(b), ninv29);
        if (!normalised(Pr) || !eq(big_mod_r(big_from9(Pr)), fr_mont_mul(fr_add(a, c), b, P.ninv))) { printf("sum-of-two operand: wrong product\n"); fails++; }
        if (!eq(big_mod_r(big_from9(fr9_norm(S))), fr_add(a, c)) || !normalised(fr9_norm(S))) { printf("norm changed the value\n"); fails++; }
    }
    // sums of five product outputs (a partial round's new s0) normalise and square correctly
    for (int it = 0; it < 500; it++) {
        fr_t acc = fr_zero(); fr9_t s = fr9_from(acc);
        for (int j = 0; j < 5; j++) { const fr_t a = rnd_fr(), b = rnd_fr(); s = fr9_add(s, fr9_mont(lazy_rep(a, 7 * j), fr9_from(b), ninv29)); acc = fr_add(acc, fr_mont_mul(a, b, P.ninv)); }
        s = fr9_norm(s);
        const fr9_t sq = fr9_mont(s, s, ninv29);
        if (!eq(big_mod_r(big_from9(sq)), fr_mont_mul(acc, acc, P.ninv))) { printf("square of a five-term sum differs\n"); fails++; }
    }
    // back to canonical: x / R < r + 1 for x < 2^261
    for (int it = 0; it < 1000; it++) {
        const fr_t a = rnd_fr(); const fr9_t A = lazy_rep(a, it % 100); fr9_t one9; for (int i = 0; i < 9; i++) one9.t[i] = i == 0;
        fr_t s = fr9_pack(fr9_mont(A, one9, ninv29)); if (fr_geq_mod(s)) s = fr_sub_mod_raw(s);
        if (fr_geq_mod(s) || !eq(s, fr_mont_mul(a, fr_from_u64(1), P.ninv))) { printf("canonicalisation differs\n"); fails++; }
    }
    printf("widest product: %d bits\n", max_bits);
    if (fails) { printf("%d failures\n", fails); return 1; }
    printf("OK\n");
    return 0;
}
