// Micro-benchmark + check of the values-phase Goldilocks Poseidon permutation (csrc/glperm.h) on ONE wavefront: a chain of n permutations against a
// plain host walk of the same fast form over the same (random, canonical) constants, and the cycles one permutation takes (the round-3 form, in the
// history before this file: 31.0 k cycles on the small-entry path, 41.8 k on the dense one: profiles/r04_ubench_glperm.txt).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I halo2-plonky2-verifier_amd/csrc -I include tools/ubench/ubench_glperm.hip -o gpurun_out/ubench_glperm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include "common.h"
#include "batchargs.h"
using namespace h2w;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int V> __global__ void k_perm(const h2w_poseidon_consts_t *k, uint64_t *io, int n, long long *cyc, int small) {
    stage_glp_consts<true>(k, threadIdx.x, 64);
    const int lane = threadIdx.x;
    uint64_t x = (lane & 15) < SPONGE_WIDTH ? io[lane & 15] : 0;      // every 16-lane row alike
    const long long t0 = clock64();
    for (int i = 0; i < n; i++) {
        x = glp_permute_lanes(x, (lds64_t *)s_glp_k, (lds64_t *)s_glp_m, (lds64_t *)s_glp_x, lane, small != 0);
    }
    const long long t1 = clock64();
    if (lane < SPONGE_WIDTH) io[SPONGE_WIDTH + lane] = x;
    if (lane == 0) *cyc = t1 - t0;
}

static uint64_t hmul(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a * b) % GL_P); }
static uint64_t hadd(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a + b) % GL_P); }
static uint64_t hpow7(uint64_t x) { const uint64_t x2 = hmul(x, x), x3 = hmul(x2, x), x4 = hmul(x2, x2); return hmul(x3, x4); }
static void host_perm(uint64_t *s, const uint64_t *K) {      // hash/poseidon/permutation.rs:216-284
    auto full = [&](int rc) {
        uint64_t t[12], o[12];
        for (int i = 0; i < 12; i++) t[i] = hpow7(hadd(s[i], K[KO_ARC + 12 * rc + i]));
        for (int r = 0; r < 12; r++) { uint64_t a = hmul(t[r], K[KO_DIAG + r]); for (int j = 0; j < 12; j++) a = hadd(a, hmul(K[KO_CIRC + (j - r + 12) % 12], t[j])); o[r] = a; }
        memcpy(s, o, sizeof o);
    };
    for (int i = 0; i < 4; i++) full(i);
    for (int i = 0; i < 12; i++) s[i] = hadd(s[i], K[KO_FIRST + i]);
    { uint64_t o[12]; o[0] = s[0]; for (int c = 1; c < 12; c++) { uint64_t a = 0; for (int r = 1; r < 12; r++) a = hadd(a, hmul(K[KO_INIT + (r - 1) * 11 + (c - 1)], s[r])); o[c] = a; } memcpy(s, o, sizeof o); }
    for (int r = 0; r < 22; r++) {
        const uint64_t s0 = hadd(hpow7(s[0]), K[KO_PRC + r]);
        uint64_t d = hmul((uint64_t)(K[KO_CIRC] + K[KO_DIAG]) % GL_P, s0);      // (a wrapping u64 sum, as the reference forms it)
        for (int i = 1; i < 12; i++) d = hadd(d, hmul(K[KO_WHAT + r * 11 + i - 1], s[i]));
        for (int i = 1; i < 12; i++) s[i] = hadd(s[i], hmul(K[KO_VS + r * 11 + i - 1], s0));
        s[0] = d;
    }
    for (int i = 0; i < 4; i++) full(4 + 22 + i);
}

int main() {
    uint64_t seed = 88172645463325252ull; auto rnd = [&] { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return seed; };
    h2w_poseidon_consts_t *dk; uint64_t *io; long long *cyc;
    CK(hipMalloc(&dk, sizeof(h2w_poseidon_consts_t) + GLP_AUX_WORDS * 8)); CK(hipMalloc(&io, 24 * 8)); CK(hipMalloc(&cyc, 8));
    int bad_total = 0;
    for (int cfg = 0; cfg < 12; cfg++) {      // 4..11: full-width entries on the diagonal too | 0: tiny MDS entries (plonky2's are <= 41) | 1: entries up to 2^26 - 1 (the small path's bound, coop.h glp_small_mds) | 2: 64-bit entries (the dense path) | 3: extreme words
        std::vector<uint64_t> K(sizeof(h2w_poseidon_consts_t) / 8, 0);
        for (int i = 0; i < GLP_CONST_WORDS; i++) K[i] = cfg == 3 ? (i % 3 == 0 ? GL_P - 1 : i % 3 == 1 ? GL_P - 1 - (rnd() & 0xFFFF) : rnd() % GL_P) : rnd() % GL_P;
        for (int i = 0; i < 12; i++) {
            const uint64_t m = cfg == 0 ? 63 : cfg == 2 || cfg >= 4 ? ~0ull : (1ull << 26) - 1;
            K[KO_CIRC + i] = cfg == 1 || cfg == 3 ? m - (rnd() & 3) : (rnd() & m) % GL_P; K[KO_DIAG + i] = i == 0 ? (cfg == 0 ? 8 : (rnd() & m) % GL_P) : cfg >= 4 ? (rnd() & m) % GL_P : 0;
        }
        if (cfg == 1 || cfg == 3) K[KO_CIRC] = (1ull << 26) - 1;
        const int small = cfg != 2 && cfg < 4;
        CK(hipMemcpy(dk, K.data(), sizeof(h2w_poseidon_consts_t), hipMemcpyHostToDevice));
        { std::vector<uint64_t> aux(GLP_AUX_WORDS); glp_aux_tables(*reinterpret_cast<const h2w_poseidon_consts_t *>(K.data()), aux.data()); CK(hipMemcpy(dk + 1, aux.data(), GLP_AUX_WORDS * 8, hipMemcpyHostToDevice)); }
        uint64_t st[12]; for (int i = 0; i < 12; i++) st[i] = cfg == 3 ? GL_P - 1 - i : rnd() % GL_P;
        const int n = 64;
        for (int V = 1; V < 2; V++) {
            CK(hipMemcpy(io, st, 12 * 8, hipMemcpyHostToDevice));
            for (int rep = 0; rep < 2; rep++) {
                CK(hipMemcpy(io, st, 12 * 8, hipMemcpyHostToDevice));
                if (V == 0) hipLaunchKernelGGL(k_perm<0>, dim3(1), dim3(64), 0, 0, dk, io, n, cyc, small); else hipLaunchKernelGGL(k_perm<1>, dim3(1), dim3(64), 0, 0, dk, io, n, cyc, small);
                CK(hipDeviceSynchronize());
            }
            long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
            uint64_t out[12]; CK(hipMemcpy(out, io + 12, 12 * 8, hipMemcpyDeviceToHost));
            uint64_t ref[12]; memcpy(ref, st, sizeof ref); for (int i = 0; i < n; i++) host_perm(ref, K.data());
            int bad = 0; for (int i = 0; i < 12; i++) bad += out[i] != ref[i];
            bad_total += bad;
            printf("constants %d (%s), %s form: %.0f cycles per permutation, mismatching elements %d of 12\n", cfg, small ? "small MDS path" : "dense MDS path", "round-4", (double)c / n, bad);
        }
    }
    printf(bad_total ? "FAILED\n" : "OK\n");
    return bad_total != 0;
}
