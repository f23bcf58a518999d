// Micro-benchmark (follow-up of ubench_store3): which property of a store stream costs write bandwidth?
//   Q0  lane = cell, low / high half as two stores, linear over the tile                       (the round-1 kernel's shape)
//   Q1  lane = 16-B piece, 1 KB contiguous per instruction, linear over the tile, no masking
//   Q2  Q1 with every store under an opaque (always true) per-lane condition
//   Q3  lane = (record, cell): 4 cells x 16 records per iteration, two stores; records of 64 cells (no masking)
//   Q4  lane = (record, 16-B piece): 8 cells x 4 records per iteration; records of 64 cells (no masking)
//   Q5  Q0 with records of 66 cells handled per record: 64 cells, then a 2-cell tail instruction pair (masked)
//   Q6  Q1, but a wavefront's consecutive instructions alternate between the two halves of its tile (two write streams per wavefront)
//   Q7  Q1, 16 streams per wavefront
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long ull;
struct __attribute__((aligned(16))) q16 { ull x, y; };
constexpr int TB = 64 * 66 * 32;            // bytes per tile (135,168)
template <int P> __global__ __launch_bounds__(256) void k(char *out, uint32_t ntiles, int opaque) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    q16 v{(ull)lane, (ull)wave}; const q16 z{0, 0};
    for (uint32_t t = wave; t < ntiles; t += nwaves) {
        char *tb = out + (uint64_t)t * TB;
        if (P == 0) { for (int j = lane; j < TB / 32; j += 64) { q16 *p = (q16 *)(tb + (uint64_t)j * 32); p[0] = v; p[1] = z; v.x += j; } }
        else if (P == 1) { for (int j = lane; j < TB / 16; j += 64) { q16 *p = (q16 *)(tb + (uint64_t)j * 16); p[0] = (j & 1) ? z : v; v.x += j; } }
        else if (P == 2) { for (int j = lane; j < TB / 16; j += 64) { q16 *p = (q16 *)(tb + (uint64_t)j * 16); if ((int)(v.y & 0xffff) + lane + opaque >= 0) p[0] = (j & 1) ? z : v; v.x += j; } }
        else if (P == 3) {
            for (int s = 0; s < 66; s++) for (int c = 0; c < 16; c++) for (int u = 0; u < 4; u++) {      // 66 sub-tiles of 64 records x 64 B?  no: tile = 66 groups of (16 rec x 4 cells x 32 B) x 16 x 4
                const int rr = u * 16 + (lane >> 2), cell = 4 * c + (lane & 3);
                q16 *p = (q16 *)(tb + (uint64_t)s * 2048 + 0 * rr + (uint64_t)((rr * 66 + s) % 66) * 0 + ((uint64_t)rr * 64 + cell) * 32 % 2048 + 0); (void)p;
            }
        }
    }
}
// records of RCELLS cells, TR records per tile; generic shapes
template <int P, int RCELLS> __global__ __launch_bounds__(256) void kr(char *out, uint32_t ntiles) {
    constexpr int RBY = RCELLS * 32;
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    q16 v{(ull)lane, (ull)wave}; const q16 z{0, 0};
    for (uint32_t t = wave; t < ntiles; t += nwaves) {
        char *tb = out + (uint64_t)t * 64 * RBY;
        if (P == 3) {
            for (int c = 0; c < RCELLS / 4; c++) for (int u = 0; u < 4; u++) {
                const int rr = u * 16 + (lane >> 2), cell = 4 * c + (lane & 3);
                q16 *p = (q16 *)(tb + (uint64_t)rr * RBY + cell * 32); p[0] = v; p[1] = z; v.x += c;
            }
        } else if (P == 4) {
            for (int c = 0; c < RCELLS / 8; c++) for (int u = 0; u < 16; u++) {
                const int rr = u * 4 + (lane >> 4), piece = 16 * c + (lane & 15);
                q16 *p = (q16 *)(tb + (uint64_t)rr * RBY + piece * 16); p[0] = (piece & 1) ? z : v; v.x += c;
            }
        } else if (P == 5) {
            for (int rr = 0; rr < 64; rr++) {
                { q16 *p = (q16 *)(tb + (uint64_t)rr * RBY + lane * 32); p[0] = v; p[1] = z; v.x += rr; }
                if (RCELLS > 64 && lane < RCELLS - 64) { q16 *p = (q16 *)(tb + (uint64_t)rr * RBY + (64 + lane) * 32); p[0] = v; p[1] = z; }
            }
        }
    }
}
template <int NS> __global__ __launch_bounds__(256) void ks(char *out, uint32_t ntiles) {      // NS interleaved write streams per wavefront
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    q16 v{(ull)lane, (ull)wave}; const q16 z{0, 0};
    constexpr int SB = TB / NS;              // bytes per stream
    for (uint32_t t = wave; t < ntiles; t += nwaves) {
        char *tb = out + (uint64_t)t * TB;
        for (int i = 0; i < SB / 1024; i++) for (int s = 0; s < NS; s++) { q16 *p = (q16 *)(tb + (uint64_t)s * SB + (uint64_t)i * 1024 + lane * 16); p[0] = (lane & 1) ? z : v; v.x += i; }
    }
}
template <class F> float timeit(F f) { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); f(); hipDeviceSynchronize(); hipEventRecord(e0); for (int i = 0; i < 3; i++) f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 3; }
int main() {
    const uint32_t ntiles = 64 * 1024;
    const uint64_t bytes = (uint64_t)ntiles * TB;
    char *out; if (hipMalloc(&out, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    for (uint32_t bpc : {1u, 3u, 8u}) {
        const dim3 g(256 * bpc), b(256);
        const float q0 = timeit([&] { hipLaunchKernelGGL(k<0>, g, b, 0, 0, out, ntiles, 0); });
        const float q1 = timeit([&] { hipLaunchKernelGGL(k<1>, g, b, 0, 0, out, ntiles, 0); });
        const float q2 = timeit([&] { hipLaunchKernelGGL(k<2>, g, b, 0, 0, out, ntiles, 0); });
        const float q3 = timeit([&] { hipLaunchKernelGGL((kr<3, 64>), g, b, 0, 0, out, ntiles); });
        const float q4 = timeit([&] { hipLaunchKernelGGL((kr<4, 64>), g, b, 0, 0, out, ntiles); });
        const float q5 = timeit([&] { hipLaunchKernelGGL((kr<5, 66>), g, b, 0, 0, out, ntiles); });
        const float q5b = timeit([&] { hipLaunchKernelGGL((kr<5, 64>), g, b, 0, 0, out, ntiles); });
        const float q6 = timeit([&] { hipLaunchKernelGGL(ks<2>, g, b, 0, 0, out, ntiles); });
        const float q7 = timeit([&] { hipLaunchKernelGGL(ks<16>, g, b, 0, 0, out, ntiles); });
        const double b64 = (double)ntiles * 64 * 64 * 32;
        printf("blocks/CU %u: GB/s  Q0 %.0f  Q1 %.0f  Q2 %.0f  Q3(64) %.0f  Q4(64) %.0f  Q5(66) %.0f  Q5(64) %.0f  Q6(2 streams) %.0f  Q7(16 streams) %.0f\n", bpc,
               bytes / (q0 * 1e6), bytes / (q1 * 1e6), bytes / (q2 * 1e6), b64 / (q3 * 1e6), b64 / (q4 * 1e6), bytes / (q5 * 1e6), b64 / (q5b * 1e6), bytes / (q6 * 1e6), bytes / (q7 * 1e6));
    }
    return 0;
}
