// Host check of csrc/glptab.h: the derived tables reproduce the reference walk from the S-box outputs of the fourth full round to the end of the partial
// rounds (plonky2 hash/poseidon/permutation.rs: mds_layer, partial_first_constant_layer, mds_partial_layer_init :229-243, the 22 fast partial rounds
// :245-262) when that stretch is ONE linear layer followed by one accumulator per round - the form csrc/glperm.h runs on the device.
// g++ -O2 -std=c++17 -I halo2-plonky2-verifier_amd/csrc -I include tests/cpp/glperm_check.cpp
#include <cstdio>
#include <cstring>
#include <vector>
#include "glptab.h"
using namespace h2w;

static uint64_t pow7(uint64_t x) { const uint64_t x2 = gl_mul(x, x), x3 = gl_mul(x2, x), x4 = gl_mul(x2, x2); return gl_mul(x3, x4); }

int main() {
    uint64_t seed = 0x9E3779B97F4A7C15ull; auto rnd = [&] { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return seed % GL_P; };
    int bad = 0;
    for (int trial = 0; trial < 50; trial++) {
        static h2w_poseidon_consts_t k; memset(&k, 0, sizeof k);
        for (int i = 0; i < 12; i++) { k.mds_circ[i] = trial & 1 ? rnd() : 1 + rnd() % 63; k.mds_diag[i] = i == 0 ? (trial & 1 ? rnd() : 8) : trial % 4 == 3 ? rnd() : 0; k.fast_partial_first_round_constant[i] = rnd(); }
        for (int i = 0; i < 22; i++) k.fast_partial_round_constants[i] = trial == 7 ? GL_P - 1 : rnd();
        for (int i = 0; i < 11; i++) for (int j = 0; j < 11; j++) k.fast_partial_round_initial_matrix[i][j] = rnd();
        for (int i = 0; i < 22; i++) for (int j = 0; j < 11; j++) { k.fast_partial_round_w_hats[i][j] = trial == 7 ? GL_P - 1 - j : rnd(); k.fast_partial_round_vs[i][j] = rnd(); }
        std::vector<uint64_t> aux(GLP_AUX_WORDS); glp_aux_tables(k, aux.data());
        for (int kk = 0; kk < N_PARTIAL_ROUNDS; kk++) for (int j = 0; j < kk; j++) if (aux[XO_C + N_PARTIAL_ROUNDS * kk + j] != 0) { printf("C[%d][%d] not zero\n", kk, j); bad++; }
        for (int i = 0; i < 12; i++) for (int slot = 0; slot < GLP_SLOTS; slot++) {
            const uint64_t w = aux[XO_W + i * GLP_SLOTS + slot];
            if (aux[XO_W + (12 + i) * GLP_SLOTS + slot] != gl_mul(w, 1ull << 22) || aux[XO_W + (24 + i) * GLP_SLOTS + slot] != gl_mul(w, 1ull << 44)) { printf("shifted copy of W[%d][%d]\n", i, slot); bad++; }
            if (slot == GLP_SLOTS - 1 && (w != 0 || aux[XO_W0 + slot] != 0)) { printf("idle slot not zero\n"); bad++; }
        }
        if (glp_slot_of_lane(0) != 0 || glp_slot_of_lane(11) != 11 || glp_slot_of_lane(12) != 34 || glp_slot_of_lane(16) != 12 || glp_slot_of_lane(37) != 33 || glp_slot_of_lane(38) != 34 || glp_slot_of_lane(63) != 34) { printf("slot map\n"); bad++; }
        uint64_t y[12]; for (int i = 0; i < 12; i++) y[i] = trial == 9 ? GL_P - 1 - i : rnd();
        // the reference walk: mds_layer, partial_first_constant_layer, mds_partial_layer_init, 22 rounds
        uint64_t x[12], s[12];
        for (int r = 0; r < 12; r++) { uint64_t a = gl_mul(y[r], k.mds_diag[r]); for (int i = 0; i < 12; i++) a = gl_muladd(k.mds_circ[i], y[(i + r) % 12], a); x[r] = gl_add(a, k.fast_partial_first_round_constant[r]); }
        s[0] = x[0];
        for (int c = 1; c < 12; c++) { uint64_t a = 0; for (int r = 1; r < 12; r++) a = gl_muladd(k.fast_partial_round_initial_matrix[r - 1][c - 1], x[r], a); s[c] = a; }
        const uint64_t m00 = (uint64_t)(k.mds_circ[0] + k.mds_diag[0]) % GL_P;      // (a wrapping u64 sum: chips.h mds_partial_layer_fast)
        for (int r = 0; r < N_PARTIAL_ROUNDS; r++) {
            const uint64_t a = gl_add(pow7(s[0]), k.fast_partial_round_constants[r]);
            uint64_t d = gl_mul(m00, a);
            for (int i = 1; i < 12; i++) d = gl_muladd(k.fast_partial_round_w_hats[r][i - 1], s[i], d);
            for (int i = 1; i < 12; i++) s[i] = gl_muladd(k.fast_partial_round_vs[r][i - 1], a, s[i]);
            s[0] = d;
        }
        // the table form: one linear layer onto the 34 slots, then the rounds on s0^7 alone
        uint64_t out[GLP_SLOTS];
        for (int slot = 0; slot < GLP_SLOTS; slot++) { uint64_t a = aux[XO_W0 + slot]; for (int i = 0; i < 12; i++) a = gl_muladd(aux[XO_W + i * GLP_SLOTS + slot], y[i], a); out[slot] = a; }
        uint64_t s0 = out[0], *st = out, *A = out + 12;
        for (int kk = 0; kk < N_PARTIAL_ROUNDS; kk++) {
            const uint64_t a = pow7(s0);      // (no constant: it is in the sums)
            for (int i = 1; i < 12; i++) st[i] = gl_muladd(k.fast_partial_round_vs[kk][i - 1], a, st[i]);
            for (int j = 0; j < N_PARTIAL_ROUNDS; j++) A[j] = gl_muladd(aux[XO_C + N_PARTIAL_ROUNDS * kk + j], a, A[j]);
            s0 = A[kk];
        }
        st[0] = s0;
        for (int i = 0; i < 12; i++) if (st[i] != s[i]) { printf("trial %d element %d differs\n", trial, i); bad++; }
    }
    printf(bad ? "FAILED\n" : "OK\n");
    return bad != 0;
}
