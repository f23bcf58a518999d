// Micro-benchmark (fourth round): 16-B pieces, NR records per store instruction (64/NR lanes = (64/NR) x 16 contiguous bytes per record),
// chunk-major (all 64 records of the tile get their chunk c, then chunk c+1) or record-major (a group of NR records is finished first).
// Records of 66 cells x 32 B back to back; 64 records per tile.  No selects on the stored value (single store per iteration).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long ull;
struct __attribute__((aligned(16))) q16 { ull x, y; };
constexpr int RC = 66, RBY = RC * 32, TB = 64 * RBY, NP = 2 * RC;
template <int NR, bool RECMAJOR> __global__ __launch_bounds__(256) void ks(char *out, uint32_t ntiles) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    q16 v{(ull)lane, (ull)wave};
    constexpr int LPR = 64 / NR, NCH = (NP + LPR - 1) / LPR, NG = 64 / NR;
    for (uint32_t t = wave; t < ntiles; t += nwaves) {
        char *tb = out + (uint64_t)t * TB;
        if (RECMAJOR) {
            for (int u = 0; u < NG; u++) for (int c = 0; c < NCH; c++) {
                const int rr = u * NR + lane / LPR, piece = LPR * c + lane % LPR;
                if (piece < NP) { q16 *p = (q16 *)(tb + (uint64_t)rr * RBY + piece * 16); p[0] = v; }
                v.x += c;
            }
        } else {
            for (int c = 0; c < NCH; c++) for (int u = 0; u < NG; u++) {
                const int rr = u * NR + lane / LPR, piece = LPR * c + lane % LPR;
                if (piece < NP) { q16 *p = (q16 *)(tb + (uint64_t)rr * RBY + piece * 16); p[0] = v; }
                v.x += c;
            }
        }
    }
}
__global__ __launch_bounds__(256) void klin(char *out, uint32_t ntiles) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    q16 v{(ull)lane, (ull)wave};
    for (uint32_t t = wave; t < ntiles; t += nwaves) {
        char *tb = out + (uint64_t)t * TB;
        for (int i = 0; i < TB / 1024; i++) { q16 *p = (q16 *)(tb + (uint64_t)i * 1024 + lane * 16); p[0] = v; v.x += i; }
    }
}
template <class F> float timeit(F f) { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); f(); hipDeviceSynchronize(); hipEventRecord(e0); for (int i = 0; i < 3; i++) f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 3; }
int main() {
    const uint32_t ntiles = 64 * 1024;
    const uint64_t bytes = (uint64_t)ntiles * TB;
    char *out; if (hipMalloc(&out, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    for (uint32_t bpc : {1u, 3u, 8u}) {
        const dim3 g(256 * bpc), b(256);
        float t[12];
        t[0] = timeit([&] { hipLaunchKernelGGL(klin, g, b, 0, 0, out, ntiles); });
        t[1] = timeit([&] { hipLaunchKernelGGL((ks<1, true>), g, b, 0, 0, out, ntiles); });
        t[2] = timeit([&] { hipLaunchKernelGGL((ks<2, true>), g, b, 0, 0, out, ntiles); });
        t[3] = timeit([&] { hipLaunchKernelGGL((ks<4, true>), g, b, 0, 0, out, ntiles); });
        t[4] = timeit([&] { hipLaunchKernelGGL((ks<8, true>), g, b, 0, 0, out, ntiles); });
        t[5] = timeit([&] { hipLaunchKernelGGL((ks<16, true>), g, b, 0, 0, out, ntiles); });
        t[6] = timeit([&] { hipLaunchKernelGGL((ks<1, false>), g, b, 0, 0, out, ntiles); });
        t[7] = timeit([&] { hipLaunchKernelGGL((ks<2, false>), g, b, 0, 0, out, ntiles); });
        t[8] = timeit([&] { hipLaunchKernelGGL((ks<4, false>), g, b, 0, 0, out, ntiles); });
        t[9] = timeit([&] { hipLaunchKernelGGL((ks<8, false>), g, b, 0, 0, out, ntiles); });
        t[10] = timeit([&] { hipLaunchKernelGGL((ks<16, false>), g, b, 0, 0, out, ntiles); });
        const char *nm[11] = {"lin", "rec1", "rec2", "rec4", "rec8", "rec16", "chk1", "chk2", "chk4", "chk8", "chk16"};
        printf("blocks/CU %u: GB/s", bpc);
        for (int i = 0; i < 11; i++) printf("  %s %.0f", nm[i], bytes / (t[i] * 1e6));
        printf("\n");
    }
    return 0;
}
