// Micro-benchmark: store-instruction throughput of dwordx4 stores when a wavefront writes 64/L independent regions,
// L adjacent lanes covering L*16 contiguous bytes of one region per instruction (L=1: every lane its own cache line, as a
// lane-per-run cell emitter does; L=4: a quad writes 64 contiguous bytes; L=64: fully coalesced 1 KB).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long ull;
struct __attribute__((aligned(16))) q16 { ull x, y; };
__global__ __launch_bounds__(64) void kG(q16 *out, uint32_t region_bytes, uint32_t L, uint32_t iters) {
    const uint32_t lane = threadIdx.x, grp = lane / L, sub = lane % L, ngrp = 64 / L;
    char *base = (char *)out + ((uint64_t)blockIdx.x * ngrp + grp) * region_bytes + sub * 16;
    q16 v{lane, blockIdx.x};
    for (uint32_t i = 0; i < iters; i++) { *(q16 *)(base + (uint64_t)i * L * 16) = v; v.x += i; }
}
// Q: the BN254 quad emitter's shape: a quad owns a region; lane l writes run l (run_cells cells, 2 x 16 B stores per cell) of each
// 4-run group, either lane-private (coop=0: 64 lines touched per store) or quad-cooperative (coop=1: the quad writes 64 contiguous B)
__global__ __launch_bounds__(64) void kQ(q16 *out, uint32_t region_bytes, uint32_t run_cells, uint32_t groups, int coop) {
    const uint32_t lane = threadIdx.x, quad = lane >> 2, l = lane & 3;
    char *base = (char *)out + ((uint64_t)blockIdx.x * 16 + quad) * region_bytes;
    q16 v{lane, blockIdx.x};
    for (uint32_t g = 0; g < groups; g++) {
        char *gb = base + (uint64_t)g * run_cells * 4 * 32;
        if (!coop) { for (uint32_t c = 0; c < run_cells; c++) { q16 *p = (q16 *)(gb + (l * run_cells + c) * 32); p[0] = v; p[1] = v; v.x += c; } }
        else { for (uint32_t c = 0; c < run_cells * 2; c++) { q16 *p = (q16 *)(gb + c * 64 + l * 16); p[0] = v; v.x += c; } }
    }
}
// R: one run per quad (the partial rounds' lane-0-only S-box cells): mode 0 lane 0 writes it alone (2 x 16 B per cell, 16 active lanes),
// mode 1 lanes 0-1 write 32 contiguous B per instruction, mode 2 all four lanes write 64 contiguous B per instruction
__global__ __launch_bounds__(64) void kR(q16 *out, uint32_t region_bytes, uint32_t cells, int mode) {
    const uint32_t lane = threadIdx.x, quad = lane >> 2, l = lane & 3;
    char *base = (char *)out + ((uint64_t)blockIdx.x * 16 + quad) * region_bytes;
    q16 v{lane, blockIdx.x};
    if (mode == 0) { if (l == 0) for (uint32_t c = 0; c < cells; c++) { q16 *p = (q16 *)(base + c * 32); p[0] = v; p[1] = v; v.x += c; } }
    else if (mode == 1) { if (l < 2) for (uint32_t c = 0; c < cells; c++) { q16 *p = (q16 *)(base + c * 32 + l * 16); p[0] = v; v.x += c; } }
    else { for (uint32_t c = 0; c < cells / 2; c++) { q16 *p = (q16 *)(base + c * 64 + l * 16); p[0] = v; v.x += c; } }
}
template <class F> float timeit(F f) { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); f(); hipDeviceSynchronize(); hipEventRecord(e0); for (int i = 0; i < 3; i++) f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 3; }
int main() {
    const uint64_t bytes = 8ull << 30;
    q16 *out; if (hipMalloc(&out, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    for (uint32_t waves : {1024u, 2048u, 4096u}) for (uint32_t L : {1u, 2u, 4u, 8u, 16u, 64u}) {
        const uint32_t ngrp = 64 / L; const uint64_t regions = (uint64_t)waves * ngrp; const uint32_t region_bytes = (((uint32_t)(bytes / regions) & ~127u) - 128u * 37u) | 128u;   // odd multiple of 128 B: no channel aliasing
        const uint32_t iters = region_bytes / (L * 16);
        float ms = timeit([&] { hipLaunchKernelGGL(kG, dim3(waves), dim3(64), 0, 0, out, region_bytes, L, iters); });
        const double total = (double)regions * iters * L * 16;
        printf("waves %4u  L %2u (%4u B contiguous per group): %.0f GB/s, %.1f ns per wave-store, %.2f cycles/store/CU@2.4GHz\n", waves, L, L * 16, total / (ms * 1e6), ms * 1e6 / iters, ms * 1e6 * 2.4 / (iters * (waves / 256.0)));
    }
    for (uint32_t waves : {1024u, 2048u, 4096u}) for (int coop : {0, 1}) for (uint32_t run : {5u, 12u, 16u}) {
        const uint64_t regions = (uint64_t)waves * 16; const uint32_t region_bytes = (((uint32_t)(bytes / regions) & ~127u) - 128u * 37u) | 128u;
        const uint32_t groups = region_bytes / (run * 4 * 32);
        float ms = timeit([&] { hipLaunchKernelGGL(kQ, dim3(waves), dim3(64), 0, 0, out, region_bytes, run, groups, coop); });
        printf("Q waves %4u coop %d run %2u cells: %.0f GB/s\n", waves, coop, run, (double)regions * groups * run * 4 * 32 / (ms * 1e6));
    }
    for (uint32_t waves : {1024u, 4096u}) for (int mode : {0, 1, 2}) {
        const uint64_t regions = (uint64_t)waves * 16; const uint32_t region_bytes = (((uint32_t)(bytes / regions) & ~127u) - 128u * 37u) | 128u;
        const uint32_t cells = region_bytes / 32 & ~1u;
        float ms = timeit([&] { hipLaunchKernelGGL(kR, dim3(waves), dim3(64), 0, 0, out, region_bytes, cells, mode); });
        const double ninstr = mode == 0 ? cells * 2.0 : mode == 1 ? cells : cells / 2.0;
        printf("R waves %4u mode %d: %.0f GB/s, %.1f cycles/store/CU@2.4GHz\n", waves, mode, (double)regions * cells * 32 / (ms * 1e6), ms * 1e6 * 2.4 / (ninstr * (waves / 256.0)));
    }
    return 0;
}
