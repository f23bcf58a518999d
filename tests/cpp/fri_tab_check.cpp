// Host check of the per-plan table of FRI-gadget constants (csrc/chips.h FriTab, built by h2w_plan_compile): every entry equals what
// FriChip::interpolate_coset / compute_evaluation / verify_query_round compute per call in the reference (fri/mod.rs:222-322, 379-389) -
// the two-adic subgroup of the arity, its barycentric weights, the inverse generator, the generator of the LDE domain.
// Built and run by tests/test_field_lazy.py (g++).
#include <cstdio>
#include <cstdint>
#include "chips.h"
using namespace h2w;
int main() {
    int fails = 0;
    for (int lde = 5; lde <= 24; lde += 3) {
        FriTab t; fri_tab_build(t, lde);
        if (t.root_lde != gl_primitive_root_of_unity(lde) || t.lde_bits != lde) { printf("root of the LDE domain differs (%d)\n", lde); fails++; }
        // the root has order exactly 2^lde
        uint64_t x = t.root_lde; for (int i = 0; i < lde - 1; i++) x = gl_mul(x, x);
        if (x != GL_P - 1) { printf("root_lde^(2^(lde-1)) != -1 (%d)\n", lde); fails++; }
        for (int ab = 1; ab <= FRI_TAB_BITS; ab++) {
            const int n = 1 << ab; const uint64_t g = gl_primitive_root_of_unity(ab);
            uint64_t dom[MAX_ARITY]; dom[0] = 1; for (int i = 1; i < n; i++) dom[i] = gl_mul(dom[i - 1], g);
            for (int i = 0; i < n; i++) {
                if (t.dom[ab][i] != dom[i]) { printf("dom[%d][%d]\n", ab, i); fails++; }
                uint64_t pr = 1; for (int j = 0; j < n; j++) if (j != i) pr = gl_mul(pr, gl_sub(dom[i], dom[j]));
                if (t.bw[ab][i] != gl_inv(pr) || gl_mul(t.bw[ab][i], pr) != 1) { printf("bw[%d][%d]\n", ab, i); fails++; }
            }
            if (t.g_inv[ab] != gl_exp(g, (uint64_t)n - 1) || gl_mul(t.g_inv[ab], g) != 1) { printf("g_inv[%d]\n", ab); fails++; }
        }
    }
    if (fails) { printf("%d failures\n", fails); return 1; }
    printf("OK\n");
    return 0;
}
