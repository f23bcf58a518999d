// expand.hip — the bandwidth kernel: expands 32-byte block records into the advice-cell stream.
//
// One lane per output cell; a workgroup walks tiles of TILE_RECS consecutive records: one lane per record
// derives the 11 u128 bases (u64x64 product, Goldilocks divmod by shifts/adds, +2^rb-p) into LDS, then all
// lanes stream the tile's cells: cell -> (record, slot) by a 5-step LDS binary search, slot -> bit-field of a
// base or a 256-bit constant (template + constant tables staged in LDS once per workgroup), coalesced
// 32-byte stores (consecutive lanes -> consecutive cells).  HBM write-bound: 32 B written per cell,
// ~0.6 B read (40-byte record+meta per ~65 cells).  No MFMA: wide-integer bit slicing.
#include <hip/hip_runtime.h>
#include "records.h"
#include "common.h"

namespace h2w {

constexpr int TILE_RECS = 32;
constexpr int EXPAND_THREADS = 256;
typedef unsigned long long ull;
struct __attribute__((aligned(16))) u128s { ull lo, hi; };

__global__ __launch_bounds__(EXPAND_THREADS) void expand_kernel(ExpandArgs A) {
    __shared__ uint32_t s_slots[MAX_SLOTS];
    __shared__ u128s s_consts[MAX_CONSTS * 2];
    __shared__ tmpl_info_t s_info[T_MAX];
    __shared__ u128s s_bases[TILE_RECS][B_COUNT];
    __shared__ uint32_t s_pre[TILE_RECS + 1];
    __shared__ uint32_t s_sbase[TILE_RECS];
    __shared__ ull s_coff[TILE_RECS];
    __shared__ ull s_lit[TILE_RECS];

    const int tid = threadIdx.x;
    for (uint32_t i = tid; i < A.nslots; i += EXPAND_THREADS) s_slots[i] = A.slots[i];
    for (uint32_t i = tid; i < A.nconsts * 2; i += EXPAND_THREADS) s_consts[i] = ((const u128s *)A.consts)[i];
    for (uint32_t i = tid; i < A.ntmpl; i += EXPAND_THREADS) s_info[i] = A.info[i];
    __syncthreads();

    const uint64_t proof = blockIdx.y;
    const rec_t *recs = A.recs + proof * A.rec_stride;
    fr_t *out = A.out + proof * A.cell_stride;
    const uint64_t ntiles = (A.nrec + TILE_RECS - 1) / TILE_RECS;
    const u128 two_rb = (u128)1 << A.rb;

    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t r0 = tile * TILE_RECS;
        const int nr = (int)((A.nrec - r0) < TILE_RECS ? (A.nrec - r0) : TILE_RECS);
        if (tid < TILE_RECS) {
            uint32_t n = 0;
            if (tid < nr) {
                const uint64_t m = A.meta[r0 + tid];
                const rec_t rc = recs[r0 + tid];
                const uint32_t t = meta_tmpl(m);
                const tmpl_info_t ti = s_info[t];
                n = ti.ncells;
                s_sbase[tid] = ti.slot_base;
                s_coff[tid] = meta_off(m);
                s_lit[tid] = ~0ull;
                if (t == T_LITERAL) { n = (uint32_t)rc.b; s_lit[tid] = rc.a; }
                u128 V, X0, X1;
                if (ti.mode == M_WIDEV) V = ((u128)rc.b << 64) | rc.a; else V = (u128)rc.a * rc.b + rc.c;
                if (ti.mode == M_LOADW) { X0 = rc.a; X1 = rc.b; }
                else {
                    // hint of GoldilocksChip::reduce (base.rs:349-352): quotient = (V div p) mod p, remainder = V mod p.
                    // Exact division by p = 2^64-2^32+1 without a divider: (V-r)*p^-1 mod 2^64 with p^-1 = 1+2^32,
                    // plus the one possible carry bit of the quotient (V < 2^128 => V div p < 2^64 + 2^32 + 2).
                    const uint64_t r = gl_reduce128(V);
                    const u128 Dv = V - r;
                    const uint64_t dl = (uint64_t)Dv;
                    const uint64_t qlo = dl + (dl << 32);
                    const uint64_t qhi = ((u128)qlo * GL_P != Dv) ? 1 : 0;
                    X0 = gl_reduce128(((u128)qhi << 64) | qlo); X1 = r;
                }
                u128 b[B_COUNT];
                b[B_A] = rc.a; b[B_B] = rc.b; b[B_C] = rc.c; b[B_D] = rc.d; b[B_V] = V;
                b[B_X0] = X0; b[B_X0P] = X0 + two_rb - GL_P; b[B_X0PP] = X0 + two_rb;
                b[B_X1] = X1; b[B_X1P] = X1 + two_rb - GL_P; b[B_X1PP] = X1 + two_rb;
#pragma unroll
                for (int k = 0; k < B_COUNT; k++) { s_bases[tid][k].lo = (ull)b[k]; s_bases[tid][k].hi = (ull)(b[k] >> 64); }
            }
            // inclusive scan over the 32 record lanes (wave-level shuffles; lanes 0..31 of wave 0)
            uint32_t x = n;
#pragma unroll
            for (int d = 1; d < TILE_RECS; d <<= 1) { uint32_t y = __shfl_up(x, d, 64); if (tid >= d) x += y; }
            s_pre[tid + 1] = x;
            if (tid == 0) s_pre[0] = 0;
        }
        __syncthreads();
        const uint32_t total = s_pre[nr];
        for (uint32_t j = tid; j < total; j += EXPAND_THREADS) {
            int lo = 0, hi = nr;     // find i: pre[i] <= j < pre[i+1]
#pragma unroll
            for (int it = 0; it < 5; it++) { int mid = (lo + hi) >> 1; if (s_pre[mid] <= j) lo = mid; else hi = mid; }
            const int i = lo;
            const uint32_t s = j - s_pre[i];
            u128s vlo, vhi; vhi.lo = 0; vhi.hi = 0;
            const ull lit = s_lit[i];
            if (lit != ~0ull) {
                const u128s *src = (const u128s *)(A.pool + lit + s);
                vlo = src[0]; vhi = src[1];
            } else {
                const uint32_t d = s_slots[s_sbase[i] + s];
                if (d & 0x80000000u) { vlo = s_consts[(d & 0xffffu) * 2]; vhi = s_consts[(d & 0xffffu) * 2 + 1]; }
                else {
                    const u128s bs = s_bases[i][d & 15u];
                    u128 v = ((u128)bs.hi << 64) | bs.lo;
                    const uint32_t sh = (d >> 4) & 127u, w = (d >> 11) & 255u, ls = (d >> 19) & 127u;
                    v >>= sh;
                    if (w < 128) v &= (((u128)1 << w) - 1);
                    v <<= ls;
                    vlo.lo = (ull)v; vlo.hi = (ull)(v >> 64);
                }
            }
            u128s *dst = (u128s *)(out + s_coff[i] + s);
            dst[0] = vlo; dst[1] = vhi;
        }
        __syncthreads();
    }
}

void launch_expand(const ExpandArgs &A, uint64_t nproofs, int grid_x, hipStream_t stream) {
    if (A.nrec == 0 || nproofs == 0) return;
    uint64_t ntiles = (A.nrec + TILE_RECS - 1) / TILE_RECS;
    uint64_t gx = (uint64_t)grid_x; if (gx > ntiles) gx = ntiles; if (gx < 1) gx = 1;
    dim3 grid((unsigned)gx, (unsigned)nproofs);
    hipLaunchKernelGGL(expand_kernel, grid, dim3(EXPAND_THREADS), 0, stream, A);
}

}  // namespace h2w
