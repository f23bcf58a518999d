// Micro-benchmark (third round): write rate vs. how one wavefront's store instructions are laid over the stream.
// Stream = back-to-back records of 66 cells x 32 B; a wavefront owns tiles of 64 records (135,168 B, not a power of two).
//   R<k>  lane = cell (two 16-B stores, low then high half), one instruction pair covers 64/k records x k cells... (k cells per range, 64/k ranges)
//         k = 64: one 2-KB span per pair (the round-1 kernel); k = 32, 16, 8, 4: chunk-major over the tile
//   N0    k = 64 with nontemporal stores;  N1: 16-B pieces, 1 KB contiguous, nontemporal
//   W0    16-B pieces: instruction A writes the even 64-B sectors of a 2-KB span, instruction B the odd ones
//   W1    16-B pieces at 64-B lane stride (4 KB span per instruction, four instructions fill it)
//   W2    16-B pieces, 1 KB contiguous per instruction (reference "fully coalesced")
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long ull;
struct __attribute__((aligned(16))) q16 { ull x, y; };
constexpr int RC = 66, RBY = RC * 32, TB = 64 * RBY;
template <int K, bool NT> __global__ __launch_bounds__(256) void kr(char *out, uint32_t ntiles) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    q16 v{(ull)lane, (ull)wave}; const q16 z{0, 0};
    constexpr int NR = 64 / K;                       // records per instruction pair
    for (uint32_t t = wave; t < ntiles; t += nwaves) {
        char *tb = out + (uint64_t)t * TB;
        if (K == 64) {
            for (int j = lane; j < TB / 32; j += 64) {
                q16 *p = (q16 *)(tb + (uint64_t)j * 32);
                if (NT) { __builtin_nontemporal_store(v.x, &p[0].x); __builtin_nontemporal_store(v.y, &p[0].y); __builtin_nontemporal_store(0ull, &p[1].x); __builtin_nontemporal_store(0ull, &p[1].y); }
                else { p[0] = v; p[1] = z; }
                v.x += j;
            }
        } else {
            for (int c = 0; c < (RC + K - 1) / K; c++) for (int u = 0; u < 64 / NR; u++) {
                const int rr = u * NR + lane / K, cell = K * c + lane % K;
                if (cell < RC) { q16 *p = (q16 *)(tb + (uint64_t)rr * RBY + cell * 32); p[0] = v; p[1] = z; v.x += c; }
            }
        }
    }
}
template <int W, bool NT> __global__ __launch_bounds__(256) void kw(char *out, uint32_t ntiles) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    q16 v{(ull)lane, (ull)wave};
    for (uint32_t t = wave; t < ntiles; t += nwaves) {
        char *tb = out + (uint64_t)t * TB;
        if (W == 0) {
            for (int i = 0; i < TB / 2048; i++) for (int h = 0; h < 2; h++) { q16 *p = (q16 *)(tb + (uint64_t)i * 2048 + (lane >> 2) * 128 + h * 64 + (lane & 3) * 16); p[0] = v; v.x += i; }
        } else if (W == 1) {
            for (int i = 0; i < TB / 4096; i++) for (int h = 0; h < 4; h++) { q16 *p = (q16 *)(tb + (uint64_t)i * 4096 + lane * 64 + h * 16); p[0] = v; v.x += i; }
        } else {
            for (int i = 0; i < TB / 1024; i++) { q16 *p = (q16 *)(tb + (uint64_t)i * 1024 + lane * 16); if (NT) { __builtin_nontemporal_store(v.x, &p[0].x); __builtin_nontemporal_store(v.y, &p[0].y); } else p[0] = v; v.x += i; }
        }
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
        t[0] = timeit([&] { hipLaunchKernelGGL((kr<64, false>), g, b, 0, 0, out, ntiles); });
        t[1] = timeit([&] { hipLaunchKernelGGL((kr<32, false>), g, b, 0, 0, out, ntiles); });
        t[2] = timeit([&] { hipLaunchKernelGGL((kr<16, false>), g, b, 0, 0, out, ntiles); });
        t[3] = timeit([&] { hipLaunchKernelGGL((kr<8, false>), g, b, 0, 0, out, ntiles); });
        t[4] = timeit([&] { hipLaunchKernelGGL((kr<4, false>), g, b, 0, 0, out, ntiles); });
        t[5] = timeit([&] { hipLaunchKernelGGL((kr<64, true>), g, b, 0, 0, out, ntiles); });
        t[6] = timeit([&] { hipLaunchKernelGGL((kw<2, true>), g, b, 0, 0, out, ntiles); });
        t[7] = timeit([&] { hipLaunchKernelGGL((kw<0, false>), g, b, 0, 0, out, ntiles); });
        t[8] = timeit([&] { hipLaunchKernelGGL((kw<1, false>), g, b, 0, 0, out, ntiles); });
        t[9] = timeit([&] { hipLaunchKernelGGL((kw<2, false>), g, b, 0, 0, out, ntiles); });
        const char *nm[10] = {"R64", "R32", "R16", "R8", "R4", "N0", "N1", "W0", "W1", "W2"};
        printf("blocks/CU %u: GB/s", bpc);
        for (int i = 0; i < 10; i++) printf("  %s %.0f", nm[i], bytes / (t[i] * 1e6));
        printf("\n");
    }
    return 0;
}
