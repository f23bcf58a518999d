// glptab.h — derived Goldilocks-Poseidon tables of the values phase (plain C++: the plan builds them on the host, tests/cpp/glperm_check.cpp checks the
// identity they rest on).  plonky2's fast partial rounds (hash/poseidon/permutation.rs:245-262):
//     a_k = s0^7 + c_k;   s0 <- m00 a_k + sum_i w_hat[k][i] s_i;   s_i <- s_i + v[k][i] a_k        (k = 0..21, i = 1..11, m00 = circ0 + diag0 as a u64 sum)
// Every s_i is its value at the start plus a combination of the a's so far, and so is every row sum: the device (glperm.h) keeps one accumulator per
// round instead of forming a row sum per round.
#pragma once
#include "chips.h"

namespace h2w {

// Derived tables of the values phase (glperm.h), computed once on the host (glp_aux_tables) and kept behind the constants on the device.  The lanes of the
// partial rounds: lane 0 = s0, lanes 1..11 = s_i, lanes 16..37 = the accumulators A_0..A_21; their SLOTS in the tables: 0, 1..11, 12..33, and 34 = every other
// lane (zeros).
//   C[k][j] (XO_C + 22 k + j): what round k's S-box output adds to s0 of round j + 1:  sum_i w_hat[j][i] v[k][i] for k < j, circ0 + diag0 for k = j, 0 for k > j
//   W[t][i][slot] (XO_W + (12 t + i) 35 + slot), W0[slot] (XO_W0 + slot): ONE linear layer from the S-box outputs y_0..y_11 of the fourth full round to the
//       start of the partial rounds - its MDS layer, partial_first_constant_layer, mds_partial_layer_init and the row sums of all 22 rounds over the start
//       values, composed, with the round constants' share of every sum (sum_k v[k][i] c_k, sum_k C[k][j] c_k) in W0: out[slot] = W0 + sum_i W[0][i] y_i.
//       t = 0, 1, 2: the entry times 2^(22 t) - the device multiplies by the three 22-bit limbs of y_i and sums 36 products without reducing in between.
constexpr int GLP_SLOTS = 35, XO_C = 0, XO_W = N_PARTIAL_ROUNDS * N_PARTIAL_ROUNDS, XO_W0 = XO_W + 3 * SPONGE_WIDTH * GLP_SLOTS, GLP_AUX_WORDS = XO_W0 + GLP_SLOTS;
HF int glp_slot_of_lane(int l) { return l < SPONGE_WIDTH ? l : l >= 16 && l < 16 + N_PARTIAL_ROUNDS ? l - 4 : GLP_SLOTS - 1; }
inline void glp_aux_tables(const h2w_poseidon_consts_t &c, uint64_t *aux) {
    for (int k = 0; k < N_PARTIAL_ROUNDS; k++)
        for (int j = 0; j < N_PARTIAL_ROUNDS; j++) {
            uint64_t a = 0;
            if (k == j) a = (uint64_t)(c.mds_circ[0] + c.mds_diag[0]) % GL_P;      // the u64 sum as the reference forms it (chips.h mds_partial_layer_fast): it wraps for entries no published table has, and so does this
            else if (k < j) for (int i = 0; i < 11; i++) a = gl_muladd(c.fast_partial_round_w_hats[j][i] % GL_P, c.fast_partial_round_vs[k][i] % GL_P, a);
            aux[XO_C + N_PARTIAL_ROUNDS * k + j] = a;
        }
    uint64_t M[SPONGE_WIDTH][SPONGE_WIDTH];      // the dense MDS matrix: x'_r = sum_i M[r][i] y_i
    for (int r = 0; r < SPONGE_WIDTH; r++) for (int i = 0; i < SPONGE_WIDTH; i++) M[r][i] = gl_reduce128((u128)c.mds_circ[(i - r + SPONGE_WIDTH) % SPONGE_WIDTH] + (i == r ? c.mds_diag[r] : 0));
    for (int slot = 0; slot < GLP_SLOTS; slot++) {
        uint64_t g[SPONGE_WIDTH] = {0}, q = 0;      // out[slot] = q + sum_{r >= 1} g[r] x'_r   (slot 0: x'_0 itself)
        if (slot >= 1 && slot < SPONGE_WIDTH) {
            for (int r = 1; r < SPONGE_WIDTH; r++) g[r] = c.fast_partial_round_initial_matrix[r - 1][slot - 1] % GL_P;
            for (int k = 0; k < N_PARTIAL_ROUNDS; k++) q = gl_muladd(c.fast_partial_round_vs[k][slot - 1] % GL_P, c.fast_partial_round_constants[k] % GL_P, q);
        } else if (slot >= SPONGE_WIDTH && slot < SPONGE_WIDTH + N_PARTIAL_ROUNDS) {
            const int j = slot - SPONGE_WIDTH;
            for (int r = 1; r < SPONGE_WIDTH; r++) for (int i = 0; i < 11; i++) g[r] = gl_muladd(c.fast_partial_round_w_hats[j][i] % GL_P, c.fast_partial_round_initial_matrix[r - 1][i] % GL_P, g[r]);
            for (int k = 0; k < N_PARTIAL_ROUNDS; k++) q = gl_muladd(aux[XO_C + N_PARTIAL_ROUNDS * k + j], c.fast_partial_round_constants[k] % GL_P, q);
        }
        uint64_t w0 = slot == 0 ? c.fast_partial_first_round_constant[0] % GL_P : q;
        for (int r = 1; r < SPONGE_WIDTH; r++) w0 = gl_muladd(g[r], c.fast_partial_first_round_constant[r] % GL_P, w0);
        aux[XO_W0 + slot] = w0;
        for (int i = 0; i < SPONGE_WIDTH; i++) {
            uint64_t w = slot == 0 ? M[0][i] : 0;
            for (int r = 1; r < SPONGE_WIDTH; r++) w = gl_muladd(g[r], M[r][i], w);
            for (int t = 0; t < 3; t++) { aux[XO_W + (SPONGE_WIDTH * t + i) * GLP_SLOTS + slot] = w; w = gl_mul(w, 1ull << 22); }
        }
    }
}

}      // namespace h2w
