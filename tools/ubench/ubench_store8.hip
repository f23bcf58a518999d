// Micro-benchmark (round 3, second): why 16 streams x 2,016 B per visit store at 4.3 TB/s (ubench_store7).  One kernel, templated on the bytes a
// wavefront writes to each of its 16 streams per visit (VB) and on the stream's stride per visit (SB >= VB: SB = VB packs the visits, SB > VB leaves
// a gap so that every visit starts on a 128-byte line).  172,032 streams ("units") of NV visits each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long ull;
struct __attribute__((aligned(16))) q16 { ull x, y; };
template <int VB, int SB, int NV> __global__ __launch_bounds__(256) void k_visit(char *out, uint32_t nblocks_q, uint32_t nitems) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t wpi = nblocks_q / 16, total = wpi * nitems;
    constexpr uint64_t UNIT_B = (uint64_t)SB * NV;
    q16 v{(ull)lane, (ull)wave};
    for (uint32_t w = wave; w < total; w += nwaves) {
        const uint32_t item = w / wpi, ui0 = (w % wpi) * 16;
        for (int r = 0; r < NV; r++) {
#pragma unroll 2
            for (int qq = 0; qq < 16; qq++) {
                char *b = out + ((uint64_t)(ui0 + qq) * nitems + item) * UNIT_B + (uint64_t)r * SB;
#pragma unroll
                for (int i = 0; i < (VB + 1023) / 1024; i++) if (i * 1024 + lane * 16 < VB) *(q16 *)(b + i * 1024 + lane * 16) = v;
            }
            v.x += r;
        }
    }
}
template <class F> float timeit(F f) { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); f(); hipDeviceSynchronize(); hipEventRecord(e0); for (int i = 0; i < 3; i++) f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 3; }
template <int VB, int SB, int NV> void run(char *out, const char *what) {
    const uint32_t nblocks_q = 64 * 28, nitems = 96;
    const double bytes = (double)nblocks_q * nitems * VB * NV;
    const float t = timeit([&] { hipLaunchKernelGGL((k_visit<VB, SB, NV>), dim3(512), dim3(256), 0, 0, out, nblocks_q, nitems); });
    printf("  visit %5d B, stride %5d B, %3d visits: %.2f ms = %.0f GB/s   %s\n", VB, SB, NV, t, bytes / (t * 1e6), what);
}
int main() {
    char *out; if (hipMalloc(&out, 24ull << 30) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(out, 0, 24ull << 30);
    printf("16 streams per wavefront, 2 wavefronts per SIMD, 172,032 streams:\n");
    run<2016, 2016, 64>(out, "the emission kernel's layers: 63 cells, packed");
    run<2048, 2048, 64>(out, "64 cells: every visit is whole 128-byte lines");
    run<2016, 2048, 64>(out, "63 cells on a 2 KB pitch: aligned starts, ragged ends");
    run<1920, 1920, 64>(out, "60 cells = 15 lines: aligned by size");
    run<1984, 1984, 64>(out, "62 cells: 64-byte aligned");
    run<1024, 1024, 128>(out, "");
    run<1008, 1008, 128>(out, "");
    run<4096, 4096, 32>(out, "");
    run<4032, 4032, 32>(out, "");
    run<8192, 8192, 16>(out, "");
    run<8064, 8064, 16>(out, "");
    run<16384, 16384, 8>(out, "");
    return 0;
}
