// Micro-benchmark (round 3): the store pattern of k_merkle_bn_emit without its arithmetic.
// 172,032 permutation units of 4,032 cells x 32 B (the 64-proof cfg-3 launch: 22.2 GB); a unit's cells are contiguous, the units of one
// wavefront (its 16 quads) lie in 16 different query blocks.  Per round (64 of them) a wavefront stores one layer (63 cells = 2,016 B) of each
// of its units, 1 KB per store instruction:
//   streams16   16 units per wavefront, round by round (what the emission kernel does)
//   streams16 + alu N   the same with N dependent multiply-adds per round and lane between the layers (arithmetic to overlap)
//   seq         one unit per wavefront at a time: its 129 KB written front to back (what a round-parallel emission would do)
//   lin         the whole buffer front to back, 1 KB per instruction
// at 1, 2, 4 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long ull;
struct __attribute__((aligned(16))) q16 { ull x, y; };
constexpr int UNIT_CELLS = 4032, ROUNDS = 64, LAYER_B = UNIT_CELLS * 32 / ROUNDS;      // 2,016 B per layer
constexpr uint64_t UNIT_B = (uint64_t)UNIT_CELLS * 32;
// unit u of the launch -> byte offset: units of a (proof, query) block are contiguous (126 of them: 7 trees x 18 levels), consecutive u of one
// wavefront differ in (proof, query): u = item * upad + ui, ui -> block, item -> unit inside the block
__device__ __forceinline__ uint64_t unit_base(uint32_t item, uint32_t ui) { return ((uint64_t)ui * 96 + item) * UNIT_B; }

template <int ALU> __global__ __launch_bounds__(256) void k_streams(char *out, uint32_t nblocks_q, uint32_t nitems, ull *sink) {
    const int lane = threadIdx.x & 63, q = lane >> 2, l4 = lane & 3;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t wpi = nblocks_q / 16, total = wpi * nitems;
    q16 v{(ull)lane, (ull)wave}; ull acc = lane;
    for (uint32_t w = wave; w < total; w += nwaves) {
        const uint32_t item = w / wpi, ui0 = (w % wpi) * 16;
        for (int r = 0; r < ROUNDS; r++) {
            if (ALU) { for (int i = 0; i < ALU; i++) acc = acc * 0x9e3779b97f4a7c15ull + v.y; v.x ^= acc; }
            // wavefront-wide flush: quad after quad, 1 KB contiguous per instruction (2,016 B per quad layer = 126 pieces of 16 B: two instructions, the second ragged)
#pragma unroll 4
            for (int qq = 0; qq < 16; qq++) {
                char *b = out + unit_base(item, ui0 + qq) + (uint64_t)r * LAYER_B;
                *(q16 *)(b + lane * 16) = v;
                if (lane < 62) *(q16 *)(b + 1024 + lane * 16) = v;
            }
        }
    }
    if (acc == 0x1234567) sink[0] = acc + q + l4;
}
__global__ __launch_bounds__(256) void k_seq(char *out, uint32_t nunits) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    q16 v{(ull)lane, (ull)wave};
    for (uint32_t u = wave; u < nunits; u += nwaves) {
        char *b = out + (uint64_t)u * UNIT_B;
        for (int i = 0; i < (int)(UNIT_B / 1024); i++) { *(q16 *)(b + (uint64_t)i * 1024 + lane * 16) = v; v.x += i; }
    }
}
// sixteen units of a wavefront front to back one after the other would be k_seq; this one interleaves 16 units but with 8 KB (4 layers) per visit
__global__ __launch_bounds__(256) void k_streams_4(char *out, uint32_t nblocks_q, uint32_t nitems) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t wpi = nblocks_q / 16, total = wpi * nitems;
    q16 v{(ull)lane, (ull)wave};
    for (uint32_t w = wave; w < total; w += nwaves) {
        const uint32_t item = w / wpi, ui0 = (w % wpi) * 16;
        for (int r = 0; r < ROUNDS; r += 4)
            for (int qq = 0; qq < 16; qq++) {
                char *b = out + unit_base(item, ui0 + qq) + (uint64_t)r * LAYER_B;
#pragma unroll
                for (int i = 0; i < 8; i++) if (i < 7 || lane < 56) *(q16 *)(b + i * 1024 + lane * 16) = v;      // 4 x 2,016 = 8,064 B
            }
    }
}
template <class F> float timeit(F f) { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); f(); hipDeviceSynchronize(); hipEventRecord(e0); for (int i = 0; i < 3; i++) f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 3; }
int main() {
    const uint32_t nblocks_q = 64 * 28, nitems = 96;       // 1,792 (proof, query) blocks x 96 units each = 172,032 units
    const uint32_t nunits = nblocks_q * nitems;
    const uint64_t bytes = (uint64_t)nunits * UNIT_B;
    char *out; ull *sink; if (hipMalloc(&out, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(out, 0, bytes);
    printf("%u units, %.2f GB\n", nunits, bytes / 1e9);
    for (uint32_t wps : {1u, 2u, 4u}) {
        const dim3 g(256 * wps), b(256);
        float t[8];
        t[0] = timeit([&] { hipLaunchKernelGGL(k_seq, g, b, 0, 0, out, nunits); });
        t[1] = timeit([&] { hipLaunchKernelGGL((k_streams<0>), g, b, 0, 0, out, nblocks_q, nitems, sink); });
        t[2] = timeit([&] { hipLaunchKernelGGL((k_streams<128>), g, b, 0, 0, out, nblocks_q, nitems, sink); });
        t[3] = timeit([&] { hipLaunchKernelGGL((k_streams<256>), g, b, 0, 0, out, nblocks_q, nitems, sink); });
        t[4] = timeit([&] { hipLaunchKernelGGL((k_streams<512>), g, b, 0, 0, out, nblocks_q, nitems, sink); });
        t[5] = timeit([&] { hipLaunchKernelGGL(k_streams_4, g, b, 0, 0, out, nblocks_q, nitems); });
        const char *nm[6] = {"seq", "streams16", "streams16+alu128", "streams16+alu256", "streams16+alu512", "streams16x8KB"};
        printf("wavefronts/SIMD %u:", wps);
        for (int i = 0; i < 6; i++) printf("  %s %.2f ms = %.0f GB/s", nm[i], t[i], bytes / (t[i] * 1e6));
        printf("\n");
    }
    return 0;
}
