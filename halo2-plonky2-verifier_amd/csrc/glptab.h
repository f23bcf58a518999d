// glptab.h — derived Goldilocks-Poseidon tables of the values phase (plain C++: the plan builds them on the host, tests/cpp/glperm_check.cpp checks the
// identity they rest on).  plonky2's fast partial rounds (hash/poseidon/permutation.rs:245-262):
//     a_k = s0^7 + c_k;   s0 <- m00 a_k + sum_i w_hat[k][i] s_i;   s_i <- s_i + v[k][i] a_k        (k = 0..21, i = 1..11, m00 = circ0 + diag0 as a u64 sum)
// Every s_i is its value at the start plus a combination of the a's so far, and so is every row sum: the device (glperm.h) keeps one accumulator per
// round instead of forming a row sum per round.
#pragma once
#include "chips.h"

namespace h2w {

// Derived tables of the values phase (glperm.h: the partial rounds with their row sums unrolled into per-round accumulators), computed once on the host
// (glp_aux_tables) and kept behind the constants on the device:
//   C[k][j] (XO_C + 22 k + j): what round k's S-box output a_k adds to s0 of round j + 1:  sum_i w_hat[j][i] v[k][i] for k < j, circ0 + diag0 for k = j, 0 for k > j
//   G[r][j] (XO_G + 22 (r - 1) + j): what element r (1..11) of the state BEFORE mds_partial_layer_init adds to round j's row sum:  sum_c w_hat[j][c] init[r-1][c-1]
//   Q[l] (XO_Q + l, l = lane 0..63): what the round constants c_k add to lane l's value over all 22 rounds - sum_k v[k][i] c_k on lane i = 1..11, sum_k C[k][j] c_k on
//            lane 16 + j - so that the rounds multiply by s0^7 alone and the constants are in the sums from the start
constexpr int XO_C = 0, XO_G = N_PARTIAL_ROUNDS * N_PARTIAL_ROUNDS, XO_Q = XO_G + 11 * N_PARTIAL_ROUNDS, GLP_AUX_WORDS = XO_Q + 64;
inline void glp_aux_tables(const h2w_poseidon_consts_t &c, uint64_t *aux) {
    for (int k = 0; k < N_PARTIAL_ROUNDS; k++)
        for (int j = 0; j < N_PARTIAL_ROUNDS; j++) {
            uint64_t a = 0;
            if (k == j) a = (uint64_t)(c.mds_circ[0] + c.mds_diag[0]) % GL_P;      // the u64 sum as the reference forms it (chips.h mds_partial_layer_fast): it wraps for entries no published table has, and so does this
            else if (k < j) for (int i = 0; i < 11; i++) a = gl_muladd(c.fast_partial_round_w_hats[j][i] % GL_P, c.fast_partial_round_vs[k][i] % GL_P, a);
            aux[XO_C + N_PARTIAL_ROUNDS * k + j] = a;
        }
    for (int r = 0; r < 11; r++)
        for (int j = 0; j < N_PARTIAL_ROUNDS; j++) {
            uint64_t a = 0;
            for (int i = 0; i < 11; i++) a = gl_muladd(c.fast_partial_round_w_hats[j][i] % GL_P, c.fast_partial_round_initial_matrix[r][i] % GL_P, a);
            aux[XO_G + N_PARTIAL_ROUNDS * r + j] = a;
        }
    for (int l = 0; l < 64; l++) {
        uint64_t a = 0;
        for (int k = 0; k < N_PARTIAL_ROUNDS; k++) {
            const uint64_t t = l >= 1 && l < SPONGE_WIDTH ? c.fast_partial_round_vs[k][l - 1] % GL_P : l >= 16 && l < 16 + N_PARTIAL_ROUNDS ? aux[XO_C + N_PARTIAL_ROUNDS * k + (l - 16)] : 0;
            a = gl_muladd(t, c.fast_partial_round_constants[k] % GL_P, a);
        }
        aux[XO_Q + l] = a;
    }
}

}      // namespace h2w
