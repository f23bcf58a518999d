// Micro-benchmark: ONE wavefront, dependent chains - what an instruction costs a wavefront that has its SIMD to itself (gfx950), and two forms of
// the lazy Goldilocks product of the values passes.  hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench/ubench_wave1.hip -o gpurun_out/ubench_wave1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned __int128 u128;
constexpr uint64_t GL_EPS = 0xFFFFFFFFull, GL_P = 0xFFFFFFFF00000001ull;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint64_t glz_reduce(u128 x) {
    const uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64), hh = hi >> 32, hl = hi & GL_EPS;
    uint64_t t0 = lo - hh; if (lo < hh) t0 -= GL_EPS;
    const uint64_t t1 = hl * GL_EPS; uint64_t r = t0 + t1; if (r < t1) r += GL_EPS;
    return r;
}
__device__ __forceinline__ uint64_t mul_v0(uint64_t a, uint64_t b) { return glz_reduce((u128)a * b); }
// x = lo + p2 2^64 + p3 2^96 (mod p), any 64-bit representative: u = lo + p2 eps (carry c), v = u - p3 (borrow b), r = v + (c - b) eps
__device__ __forceinline__ uint64_t red_v2(uint64_t lo, uint32_t p2, uint32_t p3) {
    // scratch: u = v[56:57], v = v[58:59], k = v[60:61], mc = v62, mb = v63, c = s[20:21]
    uint64_t r;
    asm("v_mad_u64_u32 v[56:57], s[20:21], %[p2], -1, %[lo]\n\t"
        "v_sub_co_u32_e32 v58, vcc, v56, %[p3]\n\t"
        "s_nop 0\n\t"
        "v_cndmask_b32_e64 v62, 0, -1, s[20:21]\n\t"
        "v_subbrev_co_u32_e32 v59, vcc, 0, v57, vcc\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e64 v63, 0, -1, vcc\n\t"
        "v_cndmask_b32_e64 v61, v63, 0, s[20:21]\n\t"
        "v_sub_u32_e32 v60, v62, v63\n\t"
        "v_lshl_add_u64 %[r], v[58:59], 0, v[60:61]"
        : [r] "=v"(r) : [lo] "v"(lo), [p2] "v"(p2), [p3] "v"(p3) : "vcc", "s20", "s21", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
    return r;
}
__device__ __forceinline__ uint64_t mul_v2(uint64_t a, uint64_t b) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const uint64_t t0 = (uint64_t)a0 * b0;
    const uint64_t t1 = (uint64_t)a0 * b1 + (t0 >> 32);
    const uint64_t t2 = (uint64_t)a1 * b0 + (uint32_t)t1;
    const uint64_t t3 = (uint64_t)a1 * b1 + (t1 >> 32) + (t2 >> 32);
    return red_v2((t2 << 32) | (uint32_t)t0, (uint32_t)t3, (uint32_t)(t3 >> 32));
}
template <int V> __global__ void k_mul(uint64_t *o, int n, long long *cyc) {
    uint64_t x = o[threadIdx.x], y = o[threadIdx.x + 64];
    const long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++) { x = V == 0 ? mul_v0(x, y) : mul_v2(x, y); y = V == 0 ? mul_v0(y, x) : mul_v2(y, x); }
    }
    const long long t1 = clock64();
    o[128 + threadIdx.x] = x; o[192 + threadIdx.x] = y; if (threadIdx.x == 0) *cyc = t1 - t0;
}
// T: 0 dependent v_add_u32 | 1 the same with s_nop 0 behind each | 2 with s_nop 1 | 3 dependent v_lshl_add_u64 | 4 dependent v_mad_u64_u32
// 5 v_add_co/v_addc pairs with the two wait states | 6 v_readlane pair + v_add using the sgprs | 7 dpp add (row_shr:1) | 8 ds_bpermute dependent | 9 v_mov_dpp + add
template <int T> __global__ void k_op(uint64_t *o, int n, long long *cyc) {
    uint32_t x = (uint32_t)o[threadIdx.x], y = (uint32_t)o[threadIdx.x + 64]; uint64_t z = o[threadIdx.x];
    const long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < 64; j++) {      // (64 per trip: the trip itself - scalar bookkeeping and a taken branch - is ~40 ticks)
            if (T == 0) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(x) : "v"(y));
            if (T == 1) asm volatile("v_add_u32_e32 %0, %0, %1\n\ts_nop 0" : "+v"(x) : "v"(y));
            if (T == 2) asm volatile("v_add_u32_e32 %0, %0, %1\n\ts_nop 1" : "+v"(x) : "v"(y));
            if (T == 3) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(z) : "v"(o[0]));
            if (T == 4) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(z) : "v"(x), "v"(y) : "s20", "s21");
            if (T == 5) asm volatile("v_add_co_u32_e32 %0, vcc, %0, %2\n\ts_nop 1\n\tv_addc_co_u32_e32 %1, vcc, %1, %2, vcc" : "+v"(x), "+v"(y) : "v"((uint32_t)z) : "vcc");
            if (T == 6) asm volatile("v_readlane_b32 s20, %0, 3\n\tv_readlane_b32 s21, %1, 3\n\tv_add_u32_e32 %0, s20, %0\n\tv_add_u32_e32 %1, s21, %1" : "+v"(x), "+v"(y) : : "s20", "s21");
            if (T == 7) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x));
            if (T == 8) x = __builtin_amdgcn_ds_bpermute((int)(((threadIdx.x + 5) & 63) * 4), (int)x) + 1;
            if (T == 9) { uint32_t t; asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_u32_e32 %1, %1, %0" : "=&v"(t), "+v"(x)); }
        }
    }
    const long long t1 = clock64();
    o[128 + threadIdx.x] = x + y + z; if (threadIdx.x == 0) *cyc = t1 - t0;
}

// U: four INDEPENDENT chains of one instruction type (what the instruction costs to issue when the wavefront has other work between dependent uses):
// 0 v_add_u32 | 1 v_mad_u64_u32 | 2 v_fma_f64 | 3 v_mul_lo_u32 | 4 v_mul_hi_u32 | 5 v_mad_u32_u24 | 6 v_mov_b32_dpp (row_shr:1) | 7 v_cvt_f64_u32 | 8 v_lshl_add_u64
template <int U> __global__ void k_ilp(uint64_t *o, int n, long long *cyc) {
    uint32_t x0 = (uint32_t)o[threadIdx.x], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, y = (uint32_t)o[threadIdx.x + 64] | 1;
    uint64_t z0 = o[threadIdx.x], z1 = z0 + 1, z2 = z0 + 2, z3 = z0 + 3, zy = o[threadIdx.x + 64];
    double d0 = (double)x0, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, dy = 1.0000001;
    const long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {      // (64 instructions per trip)
            if (U == 0) asm volatile("v_add_u32_e32 %0, %0, %4\n\tv_add_u32_e32 %1, %1, %4\n\tv_add_u32_e32 %2, %2, %4\n\tv_add_u32_e32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y));
            if (U == 1) asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(z0), "+v"(z1), "+v"(z2), "+v"(z3) : "v"(x0), "v"(y) : "vcc");
            if (U == 2) asm volatile("v_fma_f64 %0, %0, %4, %0\n\tv_fma_f64 %1, %1, %4, %1\n\tv_fma_f64 %2, %2, %4, %2\n\tv_fma_f64 %3, %3, %4, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dy));
            if (U == 3) asm volatile("v_mul_lo_u32 %0, %0, %4\n\tv_mul_lo_u32 %1, %1, %4\n\tv_mul_lo_u32 %2, %2, %4\n\tv_mul_lo_u32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y));
            if (U == 4) asm volatile("v_mul_hi_u32 %0, %0, %4\n\tv_mul_hi_u32 %1, %1, %4\n\tv_mul_hi_u32 %2, %2, %4\n\tv_mul_hi_u32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y));
            if (U == 5) asm volatile("v_mad_u32_u24 %0, %0, %4, %0\n\tv_mad_u32_u24 %1, %1, %4, %1\n\tv_mad_u32_u24 %2, %2, %4, %2\n\tv_mad_u32_u24 %3, %3, %4, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y));
            if (U == 6) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mov_b32_dpp %3, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
            if (U == 7) asm volatile("v_cvt_f64_u32_e32 %0, %4\n\tv_cvt_f64_u32_e32 %1, %5\n\tv_cvt_f64_u32_e32 %2, %6\n\tv_cvt_f64_u32_e32 %3, %7" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3));
            if (U == 8) asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n\tv_lshl_add_u64 %1, %1, 0, %4\n\tv_lshl_add_u64 %2, %2, 0, %4\n\tv_lshl_add_u64 %3, %3, 0, %4" : "+v"(z0), "+v"(z1), "+v"(z2), "+v"(z3) : "v"(zy));
        }
    }
    const long long t1 = clock64();
    o[128 + threadIdx.x] = x0 + x1 + x2 + x3 + z0 + z1 + z2 + z3 + (uint64_t)(d0 + d1 + d2 + d3); if (threadIdx.x == 0) *cyc = t1 - t0;
}
static uint64_t h_mul(uint64_t a, uint64_t b) { return (uint64_t)(((u128)(a % GL_P) * (b % GL_P)) % GL_P); }
int main() {
    uint64_t *o; long long *cyc; CK(hipMalloc(&o, 256 * 8)); CK(hipMalloc(&cyc, 8));
    uint64_t h[256];
    uint64_t s = 88172645463325252ull; auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int i = 0; i < 128; i++) h[i] = rnd();
    const uint64_t edge[] = {0, 1, GL_P - 1, GL_P, GL_P + 1, ~0ull, ~0ull - 1, GL_EPS, GL_EPS + 1, 1ull << 63, (1ull << 32), 0xFFFFFFFF00000000ull};
    for (int i = 0; i < 12; i++) { h[i] = edge[i]; h[64 + i] = edge[11 - i]; h[12 + i] = edge[i]; h[64 + 12 + i] = edge[i]; h[24 + i] = ~0ull; h[64 + 24 + i] = edge[i]; }
    const int n = 20000;
    for (int V : {0, 2}) {
        CK(hipMemcpy(o, h, 128 * 8, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; rep++) {
            CK(hipMemcpy(o, h, 128 * 8, hipMemcpyHostToDevice));
            hipEventRecord(e0);
            if (V == 0) hipLaunchKernelGGL(k_mul<0>, dim3(1), dim3(64), 0, 0, o, n, cyc); else hipLaunchKernelGGL(k_mul<2>, dim3(1), dim3(64), 0, 0, o, n, cyc);
            hipEventRecord(e1); CK(hipDeviceSynchronize());
        }
        float ms; hipEventElapsedTime(&ms, e0, e1); long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        uint64_t r[128]; CK(hipMemcpy(r, o + 128, 128 * 8, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; l++) { uint64_t x = h[l] % GL_P, y = h[64 + l] % GL_P; for (int i = 0; i < n * 4; i++) { x = h_mul(x, y); y = h_mul(y, x); } if (r[l] % GL_P != x || r[64 + l] % GL_P != y) bad++; }
        printf("mul_v%d: %.1f ns = %.1f clock64 ticks per product (wall %.3f ms, %lld ticks), mismatching lanes %d\n", V, ms * 1e6 / (8.0 * n), (double)c / (8.0 * n), ms, c, bad);
    }
    const char *names[] = {"v_add_u32", "v_add_u32 + s_nop 0", "v_add_u32 + s_nop 1", "v_lshl_add_u64", "v_mad_u64_u32", "add_co, s_nop 1, addc", "2 readlane + 2 add", "v_add_u32_dpp", "ds_bpermute + add", "v_mov_dpp + add"};
    for (int T = 0; T < 10; T++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            switch (T) {
#define C(i) case i: hipLaunchKernelGGL(k_op<i>, dim3(1), dim3(64), 0, 0, o, n, cyc); break;
                C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9)
            }
            hipEventRecord(e1); CK(hipDeviceSynchronize());
        }
        float ms; hipEventElapsedTime(&ms, e0, e1); long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        printf("%-26s %.2f ns = %.2f ticks per group\n", names[T], ms * 1e6 / (64.0 * n), (double)c / (64.0 * n));
    }
    const char *unames[] = {"v_add_u32", "v_mad_u64_u32", "v_fma_f64", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_mov_b32_dpp", "v_cvt_f64_u32", "v_lshl_add_u64"};
    for (int U = 0; U < 9; U++) {
        for (int rep = 0; rep < 2; rep++) {
            switch (U) {
#define D(i) case i: hipLaunchKernelGGL(k_ilp<i>, dim3(1), dim3(64), 0, 0, o, n, cyc); break;
                D(0) D(1) D(2) D(3) D(4) D(5) D(6) D(7) D(8)
            }
            CK(hipDeviceSynchronize());
        }
        long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        printf("four independent chains of %-16s %.2f ticks per instruction\n", unames[U], (double)c / (64.0 * n));
    }
    return 0;
}
