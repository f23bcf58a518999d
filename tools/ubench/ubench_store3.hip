// Micro-benchmark: write-path rate of the expansion kernel's store shapes.  The advice stream is modelled as back-to-back records of
// 66 cells (2,112 B); a wavefront owns tiles of 64 consecutive records.  No LDS, trivial values: only the stores differ.
//   P0  lane = cell: 64 consecutive cells per iteration, low and high half as two 16-B stores (32-B lane stride)      [round-1 kernel]
//   P1  lane = (record, cell): 4 cells x 16 records per iteration, low / high half as two stores
//   P2  lane = (record, 16-B piece): 8 cells (256 B) x 4 records per iteration, chunk-major (all 64 records, then the next chunk)
//   P3  as P2, record-major: the 9 chunks of 4 records back to back, then the next 4 records
//   P4  lane = 16-B piece of ONE record: 1 KB contiguous per instruction (3 instructions per record, the last one partial)
//   P5  lane = 16-B piece of TWO records: 512 B contiguous each
//   P6  P4 with the tile's records interleaved over time like P2 (chunk-major, 1 KB chunks)
// hipcc -O3 --offload-arch=gfx950 tools/ubench/ubench_store3.hip -o /tmp/ubench_store3 && /tmp/ubench_store3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long ull;
struct __attribute__((aligned(16))) q16 { ull x, y; };
constexpr int RC = 66;                      // cells per record
constexpr int RB = RC * 32;                 // bytes per record
template <int P> __global__ __launch_bounds__(256) void k(char *out, uint32_t ntiles) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    q16 v{(ull)lane, (ull)wave}; const q16 z{0, 0};
    for (uint32_t t = wave; t < ntiles; t += nwaves) {
        char *tb = out + (uint64_t)t * 64 * RB;
        if (P == 0) {
            for (int j = lane; j < 64 * RC; j += 64) { q16 *p = (q16 *)(tb + (uint64_t)j * 32); p[0] = v; p[1] = z; v.x += j; }
        } else if (P == 1) {
            for (int c = 0; c < (RC + 3) / 4; c++) for (int u = 0; u < 4; u++) {
                const int rr = u * 16 + (lane >> 2), cell = 4 * c + (lane & 3);
                if (cell < RC) { q16 *p = (q16 *)(tb + (uint64_t)rr * RB + cell * 32); p[0] = v; p[1] = z; v.x += c; }
            }
        } else if (P == 2) {
            for (int c = 0; c < (RC + 7) / 8; c++) for (int u = 0; u < 16; u++) {
                const int rr = u * 4 + (lane >> 4), piece = 16 * c + (lane & 15);
                if (piece < 2 * RC) { q16 *p = (q16 *)(tb + (uint64_t)rr * RB + piece * 16); p[0] = (piece & 1) ? z : v; v.x += c; }
            }
        } else if (P == 3) {
            for (int u = 0; u < 16; u++) for (int c = 0; c < (RC + 7) / 8; c++) {
                const int rr = u * 4 + (lane >> 4), piece = 16 * c + (lane & 15);
                if (piece < 2 * RC) { q16 *p = (q16 *)(tb + (uint64_t)rr * RB + piece * 16); p[0] = (piece & 1) ? z : v; v.x += c; }
            }
        } else if (P == 4) {
            for (int rr = 0; rr < 64; rr++) for (int c = 0; c < (2 * RC + 63) / 64; c++) {
                const int piece = 64 * c + lane;
                if (piece < 2 * RC) { q16 *p = (q16 *)(tb + (uint64_t)rr * RB + piece * 16); p[0] = (piece & 1) ? z : v; v.x += c; }
            }
        } else if (P == 5) {
            for (int r2 = 0; r2 < 32; r2++) for (int c = 0; c < (2 * RC + 31) / 32; c++) {
                const int rr = r2 * 2 + (lane >> 5), piece = 32 * c + (lane & 31);
                if (piece < 2 * RC) { q16 *p = (q16 *)(tb + (uint64_t)rr * RB + piece * 16); p[0] = (piece & 1) ? z : v; v.x += c; }
            }
        } else if (P == 6) {
            for (int c = 0; c < (2 * RC + 63) / 64; c++) for (int rr = 0; rr < 64; rr++) {
                const int piece = 64 * c + lane;
                if (piece < 2 * RC) { q16 *p = (q16 *)(tb + (uint64_t)rr * RB + piece * 16); p[0] = (piece & 1) ? z : v; v.x += c; }
            }
        }
    }
}
template <class F> float timeit(F f) { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); f(); hipDeviceSynchronize(); hipEventRecord(e0); for (int i = 0; i < 3; i++) f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 3; }
int main() {
    const uint32_t ntiles = 64 * 1024;                       // 64 Ki tiles x 135 KB = 8.9 GB
    const uint64_t bytes = (uint64_t)ntiles * 64 * RB;
    char *out; if (hipMalloc(&out, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    for (uint32_t bpc : {1u, 2u, 3u, 4u, 8u}) {              // 256-thread blocks per CU
        const dim3 g(256 * bpc), b(256);
        float t[7];
        t[0] = timeit([&] { hipLaunchKernelGGL(k<0>, g, b, 0, 0, out, ntiles); });
        t[1] = timeit([&] { hipLaunchKernelGGL(k<1>, g, b, 0, 0, out, ntiles); });
        t[2] = timeit([&] { hipLaunchKernelGGL(k<2>, g, b, 0, 0, out, ntiles); });
        t[3] = timeit([&] { hipLaunchKernelGGL(k<3>, g, b, 0, 0, out, ntiles); });
        t[4] = timeit([&] { hipLaunchKernelGGL(k<4>, g, b, 0, 0, out, ntiles); });
        t[5] = timeit([&] { hipLaunchKernelGGL(k<5>, g, b, 0, 0, out, ntiles); });
        t[6] = timeit([&] { hipLaunchKernelGGL(k<6>, g, b, 0, 0, out, ntiles); });
        printf("blocks/CU %u (waves/CU %2u): GB/s", bpc, bpc * 4);
        for (int i = 0; i < 7; i++) printf("  P%d %.0f", i, bytes / (t[i] * 1e6));
        printf("\n");
    }
    return 0;
}
