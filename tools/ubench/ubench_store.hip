// Micro-benchmark: ceiling of a write-only 32-byte-cell stream on MI355X for different store shapes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long ull;
struct __attribute__((aligned(16))) q16 { ull x, y; };
// A: lane i writes cell i as two 16-B stores (32-B lane stride)
__global__ void kA(q16 *out, uint64_t ncells) {
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < ncells; c += (uint64_t)gridDim.x * blockDim.x) {
        q16 lo{c, c ^ 0x55}, hi{0, 0}; out[2 * c] = lo; out[2 * c + 1] = hi;
    }
}
// B: lane pairs: even lane low half, odd lane high half -> 1 KB contiguous per store instruction
__global__ void kB(q16 *out, uint64_t ncells) {
    for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < 2 * ncells; h += (uint64_t)gridDim.x * blockDim.x) {
        q16 v{(h & 1) ? 0 : (h >> 1), (h & 1) ? 0 : ((h >> 1) ^ 0x55)}; out[h] = v;
    }
}
// C: like A with nontemporal stores
__global__ void kC(q16 *out, uint64_t ncells) {
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < ncells; c += (uint64_t)gridDim.x * blockDim.x) {
        __builtin_nontemporal_store((ull)c, &out[2 * c].x); __builtin_nontemporal_store((ull)(c ^ 0x55), &out[2 * c].y);
        __builtin_nontemporal_store((ull)0, &out[2 * c + 1].x); __builtin_nontemporal_store((ull)0, &out[2 * c + 1].y);
    }
}
// D: each lane writes 2 consecutive cells (64 B per lane, 4 stores)
__global__ void kD(q16 *out, uint64_t ncells) {
    for (uint64_t c = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; c < ncells; c += (uint64_t)gridDim.x * blockDim.x * 2) {
        q16 lo{c, c ^ 0x55}, hi{0, 0}; out[2 * c] = lo; out[2 * c + 1] = hi; out[2 * c + 2] = lo; out[2 * c + 3] = hi;
    }
}
// E: expand-like: block b writes whole chunks of `chunk` cells (b, b+G, ...), 256 threads stride inside the chunk, barrier per chunk
__global__ void kE(q16 *out, uint64_t ncells, uint32_t chunk, int barrier) {
    const uint64_t nch = ncells / chunk;
    for (uint64_t ch = blockIdx.x; ch < nch; ch += gridDim.x) {
        for (uint32_t j = threadIdx.x; j < chunk; j += blockDim.x) { const uint64_t c = ch * chunk + j; q16 lo{c, c ^ 0x55}, hi{0, 0}; out[2 * c] = lo; out[2 * c + 1] = hi; }
        if (barrier) __syncthreads();
    }
}
// F: unit-kernel-like: each wave owns 64 regions (region_cells apart); per step it writes `chunk` B to each region:
// a store instruction covers (1024/chunk) regions x chunk contiguous bytes
__global__ void kF(q16 *out, uint64_t ncells, uint32_t region_cells, uint32_t chunk) {
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64, lane = threadIdx.x & 63;
    const uint64_t nwaves = (uint64_t)gridDim.x * (blockDim.x / 64);
    const uint32_t lanes_per_region = chunk / 16, regions_per_instr = 64 / lanes_per_region;
    for (uint64_t w = wave; (w + 1) * 64 * region_cells <= ncells; w += nwaves) {
        const uint64_t base = w * 64 * region_cells;   // cells
        for (uint32_t off = 0; off < region_cells * 32; off += chunk) {          // byte offset inside every region
            for (uint32_t it = 0; it < 64 / regions_per_instr; it++) {
                const uint64_t region = it * regions_per_instr + lane / lanes_per_region;
                char *p = (char *)out + (base + region * region_cells) * 32 + off + (lane % lanes_per_region) * 16;
                q16 v{region, off}; *(q16 *)p = v;
            }
        }
    }
}
template <class F> float timeit(F f) { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); f(); hipDeviceSynchronize(); hipEventRecord(e0); for (int i = 0; i < 3; i++) f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 3; }
int main() {
    const uint64_t ncells = 400ull << 20;   // 12.8 GB
    q16 *out; if (hipMalloc(&out, ncells * 32) != hipSuccess) { printf("alloc failed\n"); return 1; }
    for (int blocks : {1024, 2048, 4096, 8192}) {
        float a = timeit([&] { hipLaunchKernelGGL(kA, dim3(blocks), dim3(256), 0, 0, out, ncells); });
        float b = timeit([&] { hipLaunchKernelGGL(kB, dim3(blocks), dim3(256), 0, 0, out, ncells); });
        float c = timeit([&] { hipLaunchKernelGGL(kC, dim3(blocks), dim3(256), 0, 0, out, ncells); });
        float d = timeit([&] { hipLaunchKernelGGL(kD, dim3(blocks), dim3(256), 0, 0, out, ncells); });
        double gb = ncells * 32 / 1e9;
        printf("blocks %5d: A(2x16B/lane) %.0f GB/s  B(lane-pair halves) %.0f GB/s  C(nontemporal) %.0f GB/s  D(64B/lane) %.0f GB/s\n", blocks, gb / (a * 1e-3), gb / (b * 1e-3), gb / (c * 1e-3), gb / (d * 1e-3));
    }
    double gb = ncells * 32 / 1e9;
    for (uint32_t chunk : {256u, 1024u, 2100u, 8192u}) for (int barrier : {0, 1}) for (int blocks : {2048, 8192}) {
        float e = timeit([&] { hipLaunchKernelGGL(kE, dim3(blocks), dim3(256), 0, 0, out, ncells, chunk, barrier); });
        printf("E chunk %5u cells barrier %d blocks %5d: %.0f GB/s\n", chunk, barrier, blocks, gb / (e * 1e-3));
    }
    for (uint32_t chunk : {256u, 512u, 1024u}) for (int blocks : {1345, 4096}) {
        float f = timeit([&] { hipLaunchKernelGGL(kF, dim3(blocks), dim3(64), 0, 0, out, ncells, 4032u, chunk); });
        printf("F region 4032 cells, chunk %4u B, %d one-wave blocks: %.0f GB/s\n", chunk, blocks, gb / (f * 1e-3));
    }
    return 0;
}
