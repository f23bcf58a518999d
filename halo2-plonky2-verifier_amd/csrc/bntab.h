// bntab.h — layout and host-side construction of the PoseidonBN254 tables a plan uploads (hash/poseidon_bn254/permutation.rs:86-109,138-159,163-170:
// C_CONSTANTS, S_CONSTANTS, M_MATRIX, P_MATRIX): canonical and times-R forms for the emitting kernels (coop.h), the limb form of the times-R entries
// for the values passes (coop.h bn_values, rowperm.h).  Plain C++: the host checks of the values passes build the same tables.
#pragma once
#include "chips.h"

namespace h2w {

enum { BK_C = 0, BK_S = 88, BK_M = 88 + 392, BK_P = 88 + 392 + 16, BK_N = 88 + 392 + 32, BK_ZERO = BK_N, BK_ONE = BK_N + 1, BK_T = BK_N + 2 };
constexpr int BN_PERM_CELLS = 4032;                         // cells of one permutation (20 + 1,100 + 2,912), without the Context's one cached load_zero cell
constexpr int BN_NSLOT = 20;                                 // value slots per quad (the widest layer, a full-round mix, stages 4 inputs + 16 partial sums)
constexpr int BN_SLOT_SQ = 16 * 2;                           // 16-byte units per slot row: 16 quads x 32 B
constexpr int BK_XC = 2 * BK_T;                               // behind the two forms: S_0 c of every partial round (first entry of its sparse row times its round
constexpr int BK_X = BK_XC + 56;                              // constant), canonical (the one-pass emitter, bnkc) and times R (the values pass, through s_bn_tab9)
constexpr int BK_ALL = BK_X + 56;
// host: the table a plan uploads (BatchArgs::bn_tab): canonical entries, then the same entries times R, each followed by 0 and 1
inline void bn_table_build(const h2w_poseidon_consts_t &k, const FrParams &P, fr_t *tab /*[BK_ALL]*/) {
    for (int i = 0; i < 88; i++) tab[BK_C + i] = k.bn_c[i];
    for (int i = 0; i < 392; i++) tab[BK_S + i] = k.bn_s[i];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { tab[BK_M + 4 * i + j] = k.bn_m[i][j]; tab[BK_P + 4 * i + j] = k.bn_p[i][j]; }
    tab[BK_ZERO] = fr_zero(); tab[BK_ONE] = fr_from_u64(1);
    for (int i = 0; i < BK_T; i++) tab[BK_T + i] = fr_mont_mul(tab[i], P.r2, P.ninv);
    for (int r = 0; r < 56; r++) {
        tab[BK_X + r] = fr_mont_mul(tab[BK_T + BK_S + 7 * r], tab[BK_T + BK_C + 20 + r], P.ninv);
        tab[BK_XC + r] = fr_mont_mul(tab[BK_X + r], fr_from_u64(1), P.ninv);
    }
}
// The values pass (bn_values) works on nine-limb lazy values (field.h fr9_t): its own table, the times-R entries and the BK_X block in limb
// form, 12 dwords per entry (9 used): 27 KB of LDS instead of s_bn_tab.
constexpr int BK9_N = BK_T + 56, BK9_X = BK_T, BK9_W = 12;
inline void bn_table9_build(const fr_t *tab /*[BK_ALL]: bn_table_build*/, uint32_t *tab9 /*[BK9_N * BK9_W]*/) {
    for (int i = 0; i < BK9_N; i++) {
        const fr9_t v = fr9_from(tab[i < BK_T ? BK_T + i : BK_X + (i - BK_T)]);
        for (int j = 0; j < BK9_W; j++) tab9[i * BK9_W + j] = j < 9 ? v.t[j] : 0u;
    }
}

}  // namespace h2w
