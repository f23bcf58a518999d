// Host check of the lane-cooperative Montgomery arithmetic (csrc/rowfr.h) and of the PoseidonBN254 permutation built on it (csrc/rowperm.h),
// with the 64 lanes of a wavefront simulated: the same source the values pass k_merkle_bn_values_row runs on the device.
//   * N' = -N^-1 mod 2^261 (rowconst_init);
//   * mont: four independent products per call (one per row) of lazy operands - tight limbs, sums of two, representatives a + k r up to the sizes
//     the pass forms - agree with the canonical product mod r, come out with limbs below 2^29 + 8 on lanes 0..8 and 0 above, stay below
//     a b / R + 1.01 r, and no 64-bit column sum wraps (the simulated lanes flag it);
//   * the carry out of the low half of T + m N is read off one limb (rowfr.h): every product above goes through it;
//   * bn_permute_rows: output state and the 56 x 3 S-box values of the partial rounds equal the reference permutation
//     (chips.h PoseidonBN254PermutationChip on canonical values; hash/poseidon_bn254/permutation.rs:83-203) on the published circomlib tables and on
//     full-width random tables.
// Built and run by tests/test_field_lazy.py (g++).
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include "bntab.h"
static uint32_t g_tab9[h2w::BK9_N * h2w::BK9_W];
#define RF_TAB9 g_tab9
#include "rowperm.h"
#include "poseidon_tables.h"
using namespace h2w;
namespace h2w { namespace rf { bool g_overflow = false; } }

struct Big { uint64_t w[5]; };
static Big big_from_limbs(const uint32_t *t) { Big r = {{0, 0, 0, 0, 0}}; for (int i = 8; i >= 0; i--) { uint64_t carry = t[i]; for (int j = 0; j < 5; j++) { const unsigned __int128 x = ((unsigned __int128)r.w[j] << 29) + carry; r.w[j] = (uint64_t)x; carry = (uint64_t)(x >> 64); } } return r; }
static Big big_r() { Big r = {{H2W_FR_M0, H2W_FR_M1, H2W_FR_M2, H2W_FR_M3, 0}}; return r; }
static Big big_shl(const Big &a, int k) { Big r = {{0, 0, 0, 0, 0}}; for (int j = 4; j >= 0; j--) { r.w[j] = a.w[j] << k; if (j > 0 && k) r.w[j] |= a.w[j - 1] >> (64 - k); } return r; }
static bool big_geq(const Big &a, const Big &b) { for (int j = 4; j >= 0; j--) if (a.w[j] != b.w[j]) return a.w[j] > b.w[j]; return true; }
static Big big_sub(const Big &a, const Big &b) { Big r; unsigned __int128 bw = 0; for (int j = 0; j < 5; j++) { const unsigned __int128 t = (unsigned __int128)a.w[j] - b.w[j] - (uint64_t)bw; r.w[j] = (uint64_t)t; bw = (t >> 64) & 1; } return r; }
static Big big_add(const Big &a, const Big &b) { Big r; unsigned __int128 c = 0; for (int j = 0; j < 5; j++) { const unsigned __int128 t = (unsigned __int128)a.w[j] + b.w[j] + (uint64_t)c; r.w[j] = (uint64_t)t; c = t >> 64; } return r; }
static fr_t big_mod_r(Big a) { for (int k = 40; k >= 0; k--) { const Big m = big_shl(big_r(), k); if (k <= 60 && big_geq(a, m)) a = big_sub(a, m); } fr_t r; for (int j = 0; j < 4; j++) r.l[j] = a.w[j]; if (a.w[4]) { printf("reduction left a fifth word\n"); exit(1); } return r; }
static uint64_t rng_s = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_s ^= rng_s << 7; rng_s ^= rng_s >> 9; return rng_s * 0x2545F4914F6CDD1Dull; }
static fr_t rnd_fr() { Big b = {{rnd(), rnd(), rnd(), rnd() >> 2, 0}}; return big_mod_r(b); }
static bool eq(const fr_t &a, const fr_t &b) { return a.l[0] == b.l[0] && a.l[1] == b.l[1] && a.l[2] == b.l[2] && a.l[3] == b.l[3]; }
// a + k r as tight limbs
static void lazy_limbs(const fr_t &a, unsigned k, uint32_t out[9]) {
    fr9_t x = fr9_from(a); fr_t rr; rr.l[0] = H2W_FR_M0; rr.l[1] = H2W_FR_M1; rr.l[2] = H2W_FR_M2; rr.l[3] = H2W_FR_M3; const fr9_t r9 = fr9_from(rr);
    for (unsigned i = 0; i < k; i++) x = fr9_norm(fr9_add(x, r9));
    for (int i = 0; i < 9; i++) out[i] = x.t[i];
}
// the same value with every limb loosened: limb i gives up to `slack` units of 2^29 to itself from limb i + 1 (a sum of two has such limbs)
static void loosen(uint32_t t[9], unsigned slack) { for (int i = 0; i < 8; i++) { const uint32_t take = (uint32_t)(rnd() % (slack + 1)); const uint32_t can = t[i + 1] < take ? t[i + 1] : take; t[i + 1] -= can; t[i] += can << 29; } }

int main() {
    const FrParams P = fr_params_init();
    rf::RowConst K; rf::rowconst_init(K, P);
    int fails = 0;
    {   // N N' = -1 mod 2^261
        unsigned __int128 acc = 0; bool ok = true;
        for (int k = 0; k < 9; k++) { for (int i = 0; i <= k; i++) acc += (unsigned __int128)K.n[i] * K.np[k - i]; ok = ok && ((uint32_t)acc & rf::M29) == rf::M29; acc >>= 29; }
        if (!ok) { printf("N N' != -1 mod 2^261\n"); fails++; }
    }
    const rf::LaneK L = rf::lane_consts();
    // ---- products
    const unsigned ks[][2] = {{0, 0}, {1, 1}, {6, 6}, {7, 0}, {63, 0}, {63, 1}, {25, 5}, {63, 63}};
    uint32_t widest_limb = 0;
    for (auto &kk : ks)
        for (unsigned slack = 0; slack <= 2; slack++)
            for (int it = 0; it < 60; it++) {
                fr_t a[4], b[4]; rf::A9 A; rf::V bv; uint32_t al[4][9], bl[4][9];
                for (int row = 0; row < 4; row++) {
                    a[row] = rnd_fr(); b[row] = rnd_fr();
                    if (it == 0 && row == 0) { a[row] = fr_zero(); }
                    if (it == 1 && row == 1) { for (int j = 0; j < 4; j++) a[row].l[j] = b[row].l[j] = fr_mod_limb(j); a[row].l[0] -= 1; b[row].l[0] -= 1; }      // r - 1
                    lazy_limbs(a[row], kk[0], al[row]); lazy_limbs(b[row], kk[1], bl[row]);
                    if (slack) { loosen(al[row], slack); loosen(bl[row], slack); }
                    for (int i = 0; i < 9; i++) for (int k = 0; k < 16; k++) A.a[i].l[16 * row + k] = al[row][i];
                    for (int k = 0; k < 16; k++) bv.l[16 * row + k] = k < 9 ? bl[row][k] : 0u;
                }
                rf::g_overflow = false;
                const rf::V res = rf::mont(A, bv, K, L);
                if (rf::g_overflow) { printf("a column sum wrapped (k = %u, %u, slack %u)\n", kk[0], kk[1], slack); fails++; }
                for (int row = 0; row < 4; row++) {
                    uint32_t t[9]; for (int k = 0; k < 9; k++) { t[k] = res.l[16 * row + k]; if (t[k] > widest_limb) widest_limb = t[k]; if (t[k] >= (1u << 29) + 8) { printf("limb too wide\n"); fails++; } }
                    for (int k = 9; k < 16; k++) if (res.l[16 * row + k]) { printf("lane %d of a result is not 0\n", k); fails++; }
                    const Big pv = big_from_limbs(t);
                    if (!eq(big_mod_r(pv), fr_mont_mul(a[row], b[row], P.ninv))) { printf("product differs mod r (k = %u, %u, slack %u, row %d)\n", kk[0], kk[1], slack, row); fails++; }
                    const double bound = ((double)(kk[0] + 1) * (kk[1] + 1) / 168.9 + 1.01);
                    Big acc = {{0, 0, 0, 0, 0}}; for (int m = 0; m < (int)bound + 1; m++) acc = big_add(acc, big_r());
                    if (big_geq(pv, acc)) { printf("product above its bound (k = %u, %u)\n", kk[0], kk[1]); fails++; }
                }
            }
    // ---- the permutation: published tables, then full-width random ones
    for (int tabs = 0; tabs < 2; tabs++) {
        h2w_poseidon_consts_t kc = H2W_POSEIDON_PUBLISHED;
        if (tabs == 1) { for (auto &x : kc.bn_c) x = rnd_fr(); for (auto &x : kc.bn_s) x = rnd_fr(); for (auto &rw : kc.bn_m) for (auto &x : rw) x = rnd_fr(); for (auto &rw : kc.bn_p) for (auto &x : rw) x = rnd_fr(); }
        std::vector<fr_t> tab(BK_ALL); bn_table_build(kc, P, tab.data()); bn_table9_build(tab.data(), g_tab9);
        for (int it = 0; it < 6; it++) {
            fr_t st[4], ref[4];
            for (int i = 0; i < 4; i++) { st[i] = it == 0 ? fr_zero() : rnd_fr(); if (it == 1) { for (int j = 0; j < 4; j++) st[i].l[j] = fr_mod_limb(j); st[i].l[0] -= 1; } ref[i] = st[i]; }
            // reference on canonical values, as chips.h walks it (ark; 4 full [P on the last]; 56 partial; 4 full)
            std::vector<fr_t> want_sbox;
            auto mul = [&](const fr_t &x, const fr_t &y) { return fr_mul(x, y, P); };
            auto exp5 = [&](const fr_t &x, bool rec) { const fr_t x2 = mul(x, x), x4 = mul(x2, x2), x5 = mul(x4, x); if (rec) { want_sbox.push_back(x2); want_sbox.push_back(x4); want_sbox.push_back(x5); } return x5; };
            auto mixm = [&](const h2w_fr_t m[4][4]) { fr_t ns[4]; for (int i = 0; i < 4; i++) { ns[i] = fr_zero(); for (int j = 0; j < 4; j++) ns[i] = fr_add(mul(m[j][i], ref[j]), ns[i]); } for (int i = 0; i < 4; i++) ref[i] = ns[i]; };
            auto ark = [&](int at) { for (int i = 0; i < 4; i++) ref[i] = fr_add(ref[i], kc.bn_c[at + i]); };
            ark(0);
            for (int i = 0; i < 3; i++) { for (int j = 0; j < 4; j++) ref[j] = exp5(ref[j], false); ark((i + 1) * 4); mixm(kc.bn_m); }
            for (int j = 0; j < 4; j++) ref[j] = exp5(ref[j], false); ark(16); mixm(kc.bn_p);
            for (int r = 0; r < 56; r++) {
                ref[0] = exp5(ref[0], true); ref[0] = fr_add(ref[0], kc.bn_c[20 + r]);
                fr_t ns0 = fr_zero(); for (int j = 0; j < 4; j++) ns0 = fr_add(mul(kc.bn_s[7 * r + j], ref[j]), ns0);
                for (int k = 1; k < 4; k++) ref[k] = fr_add(mul(kc.bn_s[7 * r + 4 + k - 1], ref[0]), ref[k]);
                ref[0] = ns0;
            }
            for (int i = 0; i < 3; i++) { for (int j = 0; j < 4; j++) ref[j] = exp5(ref[j], false); ark(20 + 56 + i * 4); mixm(kc.bn_m); }
            for (int j = 0; j < 4; j++) ref[j] = exp5(ref[j], false); mixm(kc.bn_m);
            std::vector<uint32_t> sbx9(56 * 3 * rf::SBX9_W, 0xdeadbeefu);
            rf::g_overflow = false;
            rf::bn_permute_rows(st, K, L, sbx9.data());
            if (rf::g_overflow) { printf("a sum wrapped inside the permutation (tables %d)\n", tabs); fails++; }
            for (int i = 0; i < 4; i++) if (!eq(st[i], ref[i])) { printf("permutation output differs (tables %d, input %d, element %d)\n", tabs, it, i); fails++; }
            for (int v = 0; v < 56 * 3; v++) {
                const uint32_t *t = sbx9.data() + (size_t)v * rf::SBX9_W;
                for (int k = 9; k < rf::SBX9_W; k++) if (t[k]) { printf("S-box value has a limb above the ninth\n"); fails++; }
                // times R, lazy: canonical value = t / R mod r
                uint32_t c = 0; fr9_t n; for (int i = 0; i < 8; i++) { const uint32_t x = t[i] + c; n.t[i] = x & rf::M29; c = x >> 29; } n.t[8] = t[8] + c;
                fr9_t one9; for (int i = 0; i < 9; i++) one9.t[i] = i == 0 ? 1u : 0u;
                fr_t s = fr9_pack(fr9_mont(n, one9, (uint32_t)P.ninv & rf::M29)); if (fr_geq_mod(s)) s = fr_sub_mod_raw(s);
                if (!eq(s, want_sbox[(size_t)v])) { printf("S-box value %d differs (tables %d, input %d)\n", v, tabs, it); fails++; break; }
            }
        }
    }
    if (fails) { printf("FAILED: %d\n", fails); return 1; }
    printf("OK widest limb: %u (2^29 + %d)\n", widest_limb, (int)widest_limb - (1 << 29));
    return 0;
}
