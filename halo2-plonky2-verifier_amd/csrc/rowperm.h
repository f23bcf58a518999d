// rowperm.h — one PoseidonBN254 permutation (hash/poseidon_bn254/permutation.rs:83-203; circomlib's optimised t = 4 form) ON VALUES, by one
// wavefront: the four 16-lane rows hold the four state elements, one 29-bit limb per lane (rowfr.h).  The values pass of the two-pass Merkle
// paths (k_merkle_bn_values_row): one wavefront per (proof, query, tree) walks the path; every permutation is ~31 k wavefront instructions
// instead of the 78 k of the four-lanes-per-path form (coop.h bn_values), and a path is as deep as before: 18 permutations.
//
// Schedule of a partial round (three product slots; every row runs one product per slot, rowfr.h mont):
//   slot 1   row 0  X2 = s0 s0        | rows k  U_k = S'_k s0' of round r - 1   (its column update: s_k += U_k)
//   slot 2   row 0  X4 = X2 X2        | row 1   XA = S_0 s0         | rows 2, 3  W_j = S_j s_j
//   slot 3   row 0  T0 = X4 XA        | row 1   W_1 = S_1 s_1       | rows 2, 3  X5 = X4 s0
//   s0' = X5 + c;   new s0 = T0 + S_0 c + W_1 + W_2 + W_3   (S_0 s0' = X4 (s0 S_0) + S_0 c: the row product does not wait for X5; S_0 c from the table)
// The operand a slot replicates (rowfr.h A9) is a table constant (nine LDS dwords per lane, the same on the 16 lanes of a row) except on the
// rows that multiply two values: there it is the nine limbs of ONE value of the wavefront (s0, X2, X4: nine v_readlane).
// Sizes of the values: as at coop.h bn_values (the algebra is the same); limbs: a product comes out below 2^29 + 8, a sum of two is a valid
// operand, a sum of more goes through rowfr.h tighten.
//
// What it leaves for the emission pass: the unit's output state (canonical, as before) and the S-box values x^2, x^4, x^5 of the 56 partial
// rounds in LIMB form, times R, 12 dwords each (k_sbox_canon turns them into the canonical values the cells show).
// RF_TAB9: the plan's limb-form table (bntab.h bn_table9_build) - LDS on the device, an array in the host check.
#pragma once
#include "rowfr.h"
#include "bntab.h"

namespace h2w {
namespace rf {

#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(1))) uint32_t gu32_t;
RF_FN A9 ld_rep(const V &entry) {                            // the nine limbs of the lane's table entry
    A9 A; const uint32_t *p = RF_TAB9 + entry * BK9_W;
    const uint4 a = *reinterpret_cast<const uint4 *>(p), b = *reinterpret_cast<const uint4 *>(p + 4);
    A.a[0] = a.x; A.a[1] = a.y; A.a[2] = a.z; A.a[3] = a.w; A.a[4] = b.x; A.a[5] = b.y; A.a[6] = b.z; A.a[7] = b.w; A.a[8] = p[8];
    return A;
}
RF_FN V ld_dist(const V &entry, const LaneK &L) { return L.low12 ? RF_TAB9[entry * BK9_W + L.k] : 0u; }      // limb k of the entry on lane k
RF_FN void st_row(uint32_t *base, const V &v, const P &rows, const LaneK &L) { if (rows && L.low12) *(gu32_t *)(base + L.k) = v; }
#else
inline A9 ld_rep(const V &entry) { A9 A; for (int j = 0; j < 9; j++) RF_LANES(A.a[j].l[i] = RF_TAB9[entry.l[i] * BK9_W + j]) return A; }
inline V ld_dist(const V &entry, const LaneK &L) { V r; RF_LANES(r.l[i] = L.low12.l[i] ? RF_TAB9[entry.l[i] * BK9_W + L.k.l[i]] : 0u) return r; }
inline void st_row(uint32_t *base, const V &v, const P &rows, const LaneK &L) { RF_LANES(if (rows.l[i] && L.low12.l[i]) base[L.k.l[i]] = v.l[i]) }
#endif

// a value held uniformly (four 64-bit words, the same on every lane) -> limb k on lane k of every row whose `pick` selects it
RF_FN uint32_t limb_of(const fr_t &x, int j) { return (uint32_t)fr_bits(x, 29 * j, 29); }
// row r of `v` (lazy limbs) -> its canonical value, the same on every lane (nine v_readlane, a scalar carry pass, one conditional subtraction)
RF_FN fr_t gather_canonical(const V &v, int row) {
    uint32_t t[9]; replicate(v, row, t);
    uint32_t c = 0; fr9_t n;
    for (int i = 0; i < 8; i++) { const uint32_t x = t[i] + c; n.t[i] = x & M29; c = x >> 29; }
    n.t[8] = t[8] + c;
    fr_t s = fr9_pack(n);
    if (fr_geq_mod(s)) s = fr_sub_mod_raw(s);
    return s;
}
RF_FN V all_rows_of_row0(const V &v) { V e, o, lo, hi; pair_bcast(v, e, o); half_bcast(e, lo, hi); return lo; }

// st: the state, canonical, the same on every lane; on return the permuted state.  sbx9: [56][3][SBX9_W] dwords of this unit (global memory).
RF_FN void bn_permute_rows(fr_t st[4], const RowConst &K, const LaneK &L, uint32_t *sbx9) {
    const V row = lane_index() >> 4;                         // = the state element this lane works on
    const V km1 = sel(row < V(2u), V(0u), sel(row == V(2u), V(1u), V(2u)));      // k - 1 on rows 1..3 (0 on row 0)
    // the state, one limb per lane, to Montgomery form (times R)
    V S;
    {
        V sc = V(0u);
        for (int j = 0; j < 9; j++) {
            const V lj = sel(L.row0, V(limb_of(st[0], j)), sel(L.row1, V(limb_of(st[1], j)), sel(row == V(2u), V(limb_of(st[2], j)), V(limb_of(st[3], j)))));
            sc = sel(L.k == V((uint32_t)j), lj, sc);
        }
        A9 A; for (int i = 0; i < 9; i++) A.a[i] = V(K.r2[i]);
        S = mont(A, sc, K, L);
    }
    S = tighten(S + ld_dist(V((uint32_t)BK_C) + row, L));                                      // ark(0)
    auto sbox = [&]() {                                                                        // x^5 on every row: three products, each row its own replicated operand
        const V X2 = mont(replicate_rows(S), S, K, L);
        const V X4 = mont(replicate_rows(X2), X2, K, L);
        S = mont(replicate_rows(X4), S, K, L);
    };
    // (an LDS read that is waited for where it stands costs a lone wavefront ~120 cycles: every table entry is read a product ahead of its use)
    struct MixK { A9 m0, m1, m2, m3; };
    auto mix_k = [&](int mb) { MixK k; k.m0 = ld_rep(V((uint32_t)mb) + row); k.m1 = ld_rep(V((uint32_t)mb + 4) + row); k.m2 = ld_rep(V((uint32_t)mb + 8) + row); k.m3 = ld_rep(V((uint32_t)mb + 12) + row); return k; };
    auto mix = [&](const MixK &k) {                                                            // new s_l = sum_j M[j][l] s_j: entry mb + 4 j + l on row l
        V e, o, s0, s1, s2, s3;
        pair_bcast(S, e, o); half_bcast(e, s0, s2); half_bcast(o, s1, s3);                     // s_j on every row
        V acc = mont(k.m0, s0, K, L);
        acc = acc + mont(k.m1, s1, K, L);
        acc = acc + mont(k.m2, s2, K, L);
        acc = acc + mont(k.m3, s3, K, L);
        S = tighten(acc);
    };
    for (int half = 0; half < 2; half++) {
        if (half == 1) {
            V s0_all = all_rows_of_row0(S);                                                    // s0 on every row
            V s0p_all = V(0u);                                                                 // s0' = s0^5 + c of the previous round (rows 1..3); nothing is pending before round 0
            struct RoundK { A9 kp, k2, k1; V kc, kx; };                                        // the table entries of a round, read a round ahead
            auto round_k = [&](int r) {
                const uint32_t ix = (uint32_t)(BK_S + (BN_WIDTH * 2 - 1) * r), ic = (uint32_t)(BK_C + (BN_FULL_ROUNDS / 2 + 1) * BN_WIDTH + r);
                RoundK k;
                k.kp = ld_rep(V(ix - (BN_WIDTH * 2 - 1) + BN_WIDTH) + km1);                    // S'_k of round r - 1 (round 0: s0' = 0, the entry read is in the table)
                k.k2 = ld_rep(V(ix) + sel(L.row1, V(0u), row));                                // row 1: S_0, rows 2, 3: S_2, S_3
                k.k1 = ld_rep(V(ix + 1));                                                      // row 1: S_1
                k.kc = ld_dist(V(ic), L); k.kx = ld_dist(V((uint32_t)(BK9_X + r)), L);
                return k;
            };
            RoundK kr = round_k(0);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
            for (int r = 0; r < BN_PARTIAL_ROUNDS; r++) {
                const RoundK kn = round_k(r + 1 < BN_PARTIAL_ROUNDS ? r + 1 : r);
                uint32_t t[9];
                uint32_t *const sb = sbx9 + (size_t)r * 3 * SBX9_W;
                // slot 1
                A9 A = kr.kp;
                replicate(S, 0, t); put_rows(A, L.row0, t);
                const V P1 = mont(A, sel(L.row0, S, s0p_all), K, L);                           // row 0: X2 | rows k: U_k
                st_row(sb, P1, L.row0, L);
                S = sel(L.row0, S, tighten(S + P1));
                // slot 2
                A = kr.k2;
                replicate(P1, 0, t); put_rows(A, L.row0, t);
                const V P2 = mont(A, sel(L.row0, P1, sel(L.row1, s0_all, S)), K, L);           // row 0: X4 | row 1: XA = S_0 s0 | rows j: W_j = S_j s_j
                st_row(sb + SBX9_W, P2, L.row0, L);
                // slot 3
                A = kr.k1;
                replicate(P2, 0, t); put_rows(A, !L.row1, t);                                  // rows 0, 2, 3: X4
                V e, o; pair_bcast(P2, e, o);                                                  // o: XA (row 1) on row 0
                const V P3 = mont(A, sel(L.row0, o, sel(L.row1, S, s0_all)), K, L);            // row 0: T0 = X4 XA | row 1: W_1 | rows 2, 3: X5 = X4 s0
                st_row(sb + 2 * SBX9_W, P3, row == V(2u), L);
                // s0' = X5 + c on rows 2, 3 -> rows 1, 2, 3
                V lo, hi; half_bcast(P3 + kr.kc, lo, hi); s0p_all = hi;
                // new s0 = (T0 + S_0 c) + W_1 + W_2 + W_3, on every row
                const V tot = sel(L.row0, P3 + kr.kx, sel(L.row1, P3, P2));
                pair_bcast(tot, e, o); half_bcast(e + o, lo, hi);
                s0_all = tighten(lo + hi);
                S = sel(L.row0, s0_all, S);
                kr = kn;
            }
            {   // the last round's column update
                const A9 A = ld_rep(V((uint32_t)(BK_S + (BN_WIDTH * 2 - 1) * (BN_PARTIAL_ROUNDS - 1) + BN_WIDTH)) + km1);
                const V U = mont(A, s0p_all, K, L);
                S = sel(L.row0, S, tighten(S + U));
            }
        }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
        for (int r = 0; r < BN_FULL_ROUNDS / 2; r++) {
            const bool last = r == BN_FULL_ROUNDS / 2 - 1;
            const MixK mk = mix_k(half == 0 && last ? BK_P : BK_M);
            const V ck = ld_dist(V((uint32_t)(BK_C + (half == 0 ? (r + 1) * BN_WIDTH : (BN_FULL_ROUNDS / 2 + 1) * BN_WIDTH + BN_PARTIAL_ROUNDS + r * BN_WIDTH))) + row, L);
            sbox();
            if (!(half == 1 && last)) S = S + ck;
            mix(mk);
        }
    }
    // back to canonical: S / R
    {
        A9 A; for (int i = 0; i < 9; i++) A.a[i] = V(i == 0 ? 1u : 0u);
        S = mont(A, S, K, L);
    }
    for (int i = 0; i < 4; i++) st[i] = gather_canonical(S, i);
}

}  // namespace rf
}  // namespace h2w
