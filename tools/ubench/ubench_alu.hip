// Micro-benchmark: issue rates of the integer / fp64 ops a 254-bit Montgomery product can be built from (gfx950).
// hipcc -O3 --offload-arch=gfx950 tools/ubench/ubench_alu.hip -o /tmp/ubench_alu && /tmp/ubench_alu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../halo2-plonky2-verifier_amd/csrc/field.h"
using namespace h2w;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 4096;

__global__ void k_mad64(uint64_t *o, uint32_t a, uint32_t b) {   // 8 independent v_mad_u64_u32 chains
    uint64_t x[8]; for (int i = 0; i < 8; i++) x[i] = threadIdx.x + i;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = (uint64_t)(uint32_t)x[i] * (uint32_t)(x[i] >> 32) + x[(i + 1) & 7];   // operands change every iteration
    }
    uint64_t s = 0; for (int i = 0; i < 8; i++) s ^= x[i]; o[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mullo(uint64_t *o, uint32_t a, uint32_t b) {
    uint32_t x[8]; for (int i = 0; i < 8; i++) x[i] = threadIdx.x + i;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = x[i] * (a + i) + b;
    }
    uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= x[i]; o[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_add32(uint64_t *o, uint32_t a, uint32_t b) {
    uint32_t x[8]; for (int i = 0; i < 8; i++) x[i] = threadIdx.x + i;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = (x[i] + a) ^ b;
    }
    uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= x[i]; o[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fma64(uint64_t *o, double a, double b) {
    double x[8]; for (int i = 0; i < 8; i++) x[i] = threadIdx.x + i;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = __builtin_fma(x[i], a, b);
    }
    double s = 0; for (int i = 0; i < 8; i++) s += x[i]; o[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}
__global__ void k_mont(fr_t *io, uint64_t ninv, int iters) {
    fr_t x = io[blockIdx.x * blockDim.x + threadIdx.x], y = io[(blockIdx.x * blockDim.x + threadIdx.x) ^ 1];
    for (int it = 0; it < iters; it++) { x = fr_mont_mul(x, y, ninv); y = fr_mont_mul(y, x, ninv); }
    io[blockIdx.x * blockDim.x + threadIdx.x] = fr_add(x, y);
}
template <class F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    uint64_t *o; CK(hipMalloc(&o, 256 * 32 * 256 * 8));
    fr_t *io; CK(hipMalloc(&io, 256 * 32 * 256 * sizeof(fr_t))); CK(hipMemset(io, 0x5a, 256 * 32 * 256 * sizeof(fr_t)));
    FrParams P = fr_params_init();
    const double ghz = 2.4;
    for (int wpc : {4, 8, 16, 32}) {   // waves per CU (256-thread blocks)
        dim3 g(256 * wpc / 4), b(256);
        double n = (double)ITER * 8;   // ops per lane
        float t1 = timeit([&] { hipLaunchKernelGGL(k_mad64, g, b, 0, 0, o, 3u, 5u); });
        float t2 = timeit([&] { hipLaunchKernelGGL(k_mullo, g, b, 0, 0, o, 3u, 5u); });
        float t3 = timeit([&] { hipLaunchKernelGGL(k_add32, g, b, 0, 0, o, 3u, 5u); });
        float t4 = timeit([&] { hipLaunchKernelGGL(k_fma64, g, b, 0, 0, o, 1.000001, 0.5); });
        // cycles per wave-instruction per SIMD = time * clk / (ops per lane * waves per SIMD)
        double wps = wpc / 4.0;
        printf("waves/CU %2d: cycles per wave-instr per SIMD (at %.1f GHz): mad_u64_u32 %.2f  mul_lo+add %.2f  add+xor(2 ops) %.2f  fma_f64 %.2f\n", wpc, ghz,
               t1 * 1e-3 * ghz * 1e9 / (n * wps), t2 * 1e-3 * ghz * 1e9 / (n * wps), t3 * 1e-3 * ghz * 1e9 / (n * wps), t4 * 1e-3 * ghz * 1e9 / (n * wps));
        int iters = 256;
        float t5 = timeit([&] { hipLaunchKernelGGL(k_mont, g, b, 0, 0, io, P.ninv, iters); });
        printf("             fr_mont_mul: %.0f cycles per wave-product per SIMD; chip %.2f G products/s\n", t5 * 1e-3 * ghz * 1e9 / (2.0 * iters * wps),
               2.0 * iters * g.x * 256 / (t5 * 1e-3) / 1e9);
    }
    {   // latency: one wave
        int iters = 256;
        float t = timeit([&] { hipLaunchKernelGGL(k_mont, dim3(1), dim3(64), 0, 0, io, P.ninv, iters); });
        printf("single wave: fr_mont_mul dependent latency %.0f cycles (%.2f us)\n", t * 1e-3 * ghz * 1e9 / (2.0 * iters), t * 1e3 / (2.0 * iters));
    }
    return 0;
}
