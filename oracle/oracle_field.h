/*
 * oracle_field.h — value-domain arithmetic for the oracle (TEST INFRASTRUCTURE ONLY, see oracle.h).
 *
 *  - BN254 scalar field Fr (halo2curves bn256::Fr): canonical 4x64 limbs.
 *  - Goldilocks field p = 2^64 - 2^32 + 1 (plonky2 GoldilocksField) and its
 *    quadratic extension GL[x]/(x^2 - 7) (field/goldilocks/extension.rs:197).
 *
 * Montgomery constants are DERIVED at init (never typed from memory).
 */
#ifndef H2W_ORACLE_FIELD_H
#define H2W_ORACLE_FIELD_H
#include "oracle.h"
#include <string.h>

typedef unsigned __int128 u128;

/* r = 21888242871839275222246405745257275088548364400416034343698204186575808495617 */
static const ofr_t FR_MOD = {{0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}};
static ofr_t FR_R2;          /* 2^512 mod r */
static uint64_t FR_NINV;     /* -r^{-1} mod 2^64 */
static int FR_INIT_DONE = 0;

static inline int fr_is_zero(const ofr_t *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fr_eq(const ofr_t *a, const ofr_t *b) { return memcmp(a, b, sizeof(ofr_t)) == 0; }
static inline int fr_geq(const ofr_t *a, const ofr_t *b) {
    for (int i = 3; i >= 0; i--) { if (a->l[i] != b->l[i]) return a->l[i] > b->l[i]; }
    return 1;
}
static inline ofr_t fr_from_u64(uint64_t x) { ofr_t r = {{x, 0, 0, 0}}; return r; }
static inline ofr_t fr_from_u128(u128 x) { ofr_t r = {{(uint64_t)x, (uint64_t)(x >> 64), 0, 0}}; return r; }
static inline int fr_fits_u128(const ofr_t *a) { return (a->l[2] | a->l[3]) == 0; }
static inline u128 fr_lo128(const ofr_t *a) { return ((u128)a->l[1] << 64) | a->l[0]; }

static inline uint64_t adc(uint64_t a, uint64_t b, uint64_t *c) { u128 t = (u128)a + b + *c; *c = (uint64_t)(t >> 64); return (uint64_t)t; }
static inline uint64_t sbb(uint64_t a, uint64_t b, uint64_t *bw) { u128 t = (u128)a - b - *bw; *bw = (uint64_t)(t >> 64) & 1; return (uint64_t)t; }

static inline ofr_t fr_sub_raw(const ofr_t *a, const ofr_t *b, uint64_t *borrow) {
    ofr_t r; uint64_t bw = 0;
    for (int i = 0; i < 4; i++) r.l[i] = sbb(a->l[i], b->l[i], &bw);
    *borrow = bw; return r;
}
static inline ofr_t fr_add(const ofr_t *a, const ofr_t *b) {
    ofr_t r; uint64_t c = 0;
    for (int i = 0; i < 4; i++) r.l[i] = adc(a->l[i], b->l[i], &c);
    /* r < 2r < 2^255: no carry out; conditional subtract */
    if (fr_geq(&r, &FR_MOD)) { uint64_t bw; r = fr_sub_raw(&r, &FR_MOD, &bw); }
    return r;
}
static inline ofr_t fr_sub(const ofr_t *a, const ofr_t *b) {
    uint64_t bw; ofr_t r = fr_sub_raw(a, b, &bw);
    if (bw) { uint64_t c = 0; for (int i = 0; i < 4; i++) r.l[i] = adc(r.l[i], FR_MOD.l[i], &c); }
    return r;
}
static inline ofr_t fr_neg(const ofr_t *a) { ofr_t z = {{0, 0, 0, 0}}; return fr_sub(&z, a); }

/* Montgomery product a*b*2^-256 mod r (CIOS) */
static inline ofr_t fr_mont_mul(const ofr_t *a, const ofr_t *b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        uint64_t c = 0;
        for (int j = 0; j < 4; j++) { u128 s = (u128)a->l[j] * b->l[i] + t[j] + c; t[j] = (uint64_t)s; c = (uint64_t)(s >> 64); }
        u128 s = (u128)t[4] + c; t[4] = (uint64_t)s; t[5] = (uint64_t)(s >> 64);
        uint64_t m = t[0] * FR_NINV;
        s = (u128)m * FR_MOD.l[0] + t[0]; c = (uint64_t)(s >> 64);
        for (int j = 1; j < 4; j++) { s = (u128)m * FR_MOD.l[j] + t[j] + c; t[j - 1] = (uint64_t)s; c = (uint64_t)(s >> 64); }
        s = (u128)t[4] + c; t[3] = (uint64_t)s; t[4] = t[5] + (uint64_t)(s >> 64);
    }
    ofr_t r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || fr_geq(&r, &FR_MOD)) { uint64_t bw; r = fr_sub_raw(&r, &FR_MOD, &bw); }
    return r;
}
static void fr_init(void) {
    if (FR_INIT_DONE) return;
    /* -r^{-1} mod 2^64 by Newton iteration */
    uint64_t inv = 1;
    for (int i = 0; i < 6; i++) inv *= 2 - FR_MOD.l[0] * inv;
    FR_NINV = (uint64_t)0 - inv;
    /* 2^512 mod r by 512 modular doublings of 1 */
    ofr_t x = fr_from_u64(1);
    for (int i = 0; i < 512; i++) x = fr_add(&x, &x);
    FR_R2 = x;
    FR_INIT_DONE = 1;
}
/* canonical product a*b mod r */
static inline ofr_t fr_mul(const ofr_t *a, const ofr_t *b) {
    if (fr_fits_u128(a) && fr_fits_u128(b) && a->l[1] == 0 && b->l[1] == 0) return fr_from_u128((u128)a->l[0] * b->l[0]);
    ofr_t t = fr_mont_mul(a, b);          /* a*b/R */
    return fr_mont_mul(&t, &FR_R2);       /* *R */
}
static inline ofr_t fr_pow(const ofr_t *a, const ofr_t *e) {
    ofr_t acc = fr_from_u64(1), base = *a;
    for (int i = 0; i < 256; i++) {
        if ((e->l[i >> 6] >> (i & 63)) & 1) acc = fr_mul(&acc, &base);
        base = fr_mul(&base, &base);
    }
    return acc;
}
static inline ofr_t fr_inv(const ofr_t *a) {
    ofr_t two = fr_from_u64(2); uint64_t bw; ofr_t e = fr_sub_raw(&FR_MOD, &two, &bw);
    return fr_pow(a, &e);
}

/* ------------------------------------------------------------------ Goldilocks */
#define GL_P 0xFFFFFFFF00000001ULL
#define GL_NEG_ONE (GL_P - 1)
#define GL_EPS 0xFFFFFFFFULL
static inline uint64_t glf_reduce128(u128 x) {   /* 2^64 = 2^32 - 1, 2^96 = -1 (mod p) */
    uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64), hh = hi >> 32, hl = hi & GL_EPS;
    uint64_t t0 = lo - hh; if (lo < hh) t0 -= GL_EPS;
    uint64_t t1 = hl * GL_EPS, r = t0 + t1; if (r < t1) r += GL_EPS;
    if (r >= GL_P) r -= GL_P;
    return r;
}
static inline uint64_t glf_add(uint64_t a, uint64_t b) { return glf_reduce128((u128)a + b); }
static inline uint64_t glf_sub(uint64_t a, uint64_t b) { return glf_reduce128((u128)a + GL_P - b); }
static inline uint64_t glf_mul(uint64_t a, uint64_t b) { return glf_reduce128((u128)a * b); }
static inline uint64_t glf_exp(uint64_t a, uint64_t e) {
    uint64_t acc = 1;
    while (e) { if (e & 1) acc = glf_mul(acc, a); a = glf_mul(a, a); e >>= 1; }
    return acc;
}
static inline uint64_t glf_inv(uint64_t a) { return glf_exp(a, GL_P - 2); }
/* plonky2 GoldilocksField: MULTIPLICATIVE_GROUP_GENERATOR = 7; POWER_OF_TWO_GENERATOR = 7^((p-1)/2^32)
 * (TWO_ADICITY = 32).  SURVEY App. B quotes 1753635133440165772; tests check the derivation equals it. */
static inline uint64_t glf_power_of_two_generator(void) { return glf_exp(7, (GL_P - 1) >> 32); }
static inline uint64_t glf_primitive_root_of_unity(int n_log) {
    uint64_t g = glf_power_of_two_generator();
    for (int i = 0; i < 32 - n_log; i++) g = glf_mul(g, g);
    return g;
}
typedef struct { uint64_t c[2]; } gle_t;
static inline gle_t gle_mul(gle_t a, gle_t b) {
    gle_t r;
    r.c[0] = glf_add(glf_mul(a.c[0], b.c[0]), glf_mul(7, glf_mul(a.c[1], b.c[1])));
    r.c[1] = glf_add(glf_mul(a.c[0], b.c[1]), glf_mul(a.c[1], b.c[0]));
    return r;
}
static inline gle_t gle_inv(gle_t a) {
    /* 1/(a0 + a1 x) = (a0 - a1 x) / (a0^2 - 7 a1^2) */
    uint64_t n = glf_sub(glf_mul(a.c[0], a.c[0]), glf_mul(7, glf_mul(a.c[1], a.c[1])));
    uint64_t ni = glf_inv(n);
    gle_t r; r.c[0] = glf_mul(a.c[0], ni); r.c[1] = glf_mul(glf_sub(0, a.c[1]), ni);
    return r;
}
#endif
