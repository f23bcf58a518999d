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
#include <cstdlib>

namespace h2w {

constexpr int EXPAND_THREADS = 256;
typedef unsigned long long ull;
struct __attribute__((aligned(16))) u128s { ull lo, hi; };

// LDS of a block: the template tables are sized by what the plan's lookup_bits needs (dynamic LDS: ~2.2 KB at lookup_bits 21, the
// static MAX_* worst case would be 9 KB) — the expansion kernel shares its CUs with the strand kernels of other batches, whose
// staging regions hold most of the LDS, and every 16 KB block that does not fit is four fewer wavefronts streaming.
extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn_tables[];
template <int ABLATE, int TILE_RECS, int NSTEP, bool COLS = false> __global__ __launch_bounds__(EXPAND_THREADS) void expand_kernel_t(ExpandArgs A) {
    u128s *s_consts = reinterpret_cast<u128s *>(s_dyn_tables);
    uint32_t *s_slots = reinterpret_cast<uint32_t *>(s_consts + A.nconsts * 2);
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

    // Tiles differ in size (1 to 66 cells per record), so a static tile -> block assignment leaves a tail of straggling blocks;
    // with a per-proof work counter every block takes the next tile when it is done (the fetch for the next tile is issued by an
    // otherwise idle wavefront during the record phase and read after the tile's closing barrier).
    __shared__ uint32_t s_next;
    // column-major emission: a record's cells are contiguous and cross at most one column boundary (the records of a tile are NOT
    // contiguous with each other: direct cells and PoseidonBN254 permutations sit between them): cell s of record i goes to
    // s_coff[i] + s (+ s_dd[i] from cell s_split[i] on), with the column shift already folded into s_coff
    __shared__ uint32_t s_split[COLS ? TILE_RECS : 1]; __shared__ ull s_dd[COLS ? TILE_RECS : 1];
    const bool dyn = A.tile_ctr != nullptr;
    uint64_t tile = blockIdx.x;
    if (dyn) { if (tid == 0) s_next = atomicAdd(&A.tile_ctr[proof], 1u); __syncthreads(); tile = s_next; __syncthreads(); }
    for (; tile < ntiles;) {
        if (dyn && tid == 64) s_next = atomicAdd(&A.tile_ctr[proof], 1u);

        const uint64_t r0 = tile * TILE_RECS;
        const int nr = (int)((A.nrec - r0) < TILE_RECS ? (A.nrec - r0) : TILE_RECS);
        if (tid < TILE_RECS) {
            uint32_t n = 0;
            bool mine = tid < nr;
            if (mine && A.shard_world > 1) {              // unit = (proof, query), round-robin; the prologue block is kept by every rank
                const uint64_t r = r0 + tid;
                if (r >= A.q_rec0_first) {
                    const uint64_t q = r < A.q_rec0_rest ? 0 : 1 + (r - A.q_rec0_rest) / A.q_nrec_rest;
                    mine = (proof * A.nq + q) % A.shard_world == A.shard_rank;
                }
            }
            if (mine) {
                const uint64_t m = A.meta[r0 + tid];
                const rec_t rc = recs[r0 + tid];
                const uint32_t t = meta_tmpl(m);
                const tmpl_info_t ti = s_info[t];
                n = ti.ncells;
                s_sbase[tid] = ti.slot_base;
                {
                    ull coff = meta_off(m);
                    if constexpr (COLS) {
                        ColCursor cc; cc.init(A.cm); cc.locate(coff);
                        const ull room = cc.hi - coff, d0 = cc.delta;               // cells of this record before the next column starts
                        ull dd = 0; uint32_t split = 0xffffffffu;
                        if (room < 4096) { split = (uint32_t)room; cc.locate(cc.hi); dd = cc.delta - d0; }
                        s_split[tid] = split; s_dd[tid] = dd; coff += d0;
                    }
                    s_coff[tid] = coff;
                }
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
                b[B_X1] = X1; b[B_X1P] = X1 + two_rb - GL_P; b[B_X1PP] = X1 + two_rb; b[B_W] = X0 * (u128)GL_P + X1;
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
            if (ABLATE == 0) {
#pragma unroll
                for (int it = 0; it < NSTEP; it++) { int mid = (lo + hi) >> 1; if (s_pre[mid] <= j) lo = mid; else hi = mid; }
            } else lo = (int)(j / 66u) < nr ? (int)(j / 66u) : nr - 1;        // timing-only ablation (wrong cells)
            const int i = lo;
            const uint32_t s = ABLATE == 0 ? j - s_pre[i] : j % 60u;
            u128s vlo, vhi; vhi.lo = 0; vhi.hi = 0;
            const ull lit = s_lit[i];
            if (ABLATE == 2) { vlo.lo = j; vlo.hi = lit; }
            else if (lit != ~0ull) {
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
            ull cellidx = s_coff[i] + s;
            if constexpr (COLS) { if (s >= s_split[i]) cellidx += s_dd[i]; }
            u128s *dst = ABLATE == 0 ? (u128s *)(out + cellidx) : (u128s *)(out + s_coff[0] + j);
#ifdef H2W_ABL_EXPAND_NOSTORE      // timing-only ablation: every cell is computed, (almost) none is written
            if (vlo.lo == 0x123456789abcdefull && vhi.hi == 0x0fedcba987654321ull) { dst[0] = vlo; dst[1] = vhi; }
#else
            dst[0] = vlo; dst[1] = vhi;
#endif
        }
        __syncthreads();
        tile = dyn ? (uint64_t)s_next : tile + gridDim.x;
        if (dyn) __syncthreads();          // s_next is rewritten at the top of the next tile
    }
}


// ---- variant 1: wave-level tiles.  Each 64-lane wavefront owns a tile of 64 records: all 64 lanes derive bases (one
// record each), then the wave streams the tile's cells.  No workgroup barrier in the loop (a wave is in lockstep), so
// the 4 waves of a workgroup overlap their record phases and store phases freely.
constexpr int WTILE = 64;
__global__ __launch_bounds__(EXPAND_THREADS) void expand_kernel_w(ExpandArgs A) {
    __shared__ uint32_t s_slots[MAX_SLOTS];
    __shared__ u128s s_consts[MAX_CONSTS * 2];
    __shared__ tmpl_info_t s_info[T_MAX];
    __shared__ u128s s_bases[4][WTILE][B_COUNT];
    __shared__ uint32_t s_pre[4][WTILE + 1];
    __shared__ uint32_t s_sbase[4][WTILE];
    __shared__ ull s_coff[4][WTILE];
    __shared__ ull s_lit[4][WTILE];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (uint32_t i = tid; i < A.nslots; i += EXPAND_THREADS) s_slots[i] = A.slots[i];
    for (uint32_t i = tid; i < A.nconsts * 2; i += EXPAND_THREADS) s_consts[i] = ((const u128s *)A.consts)[i];
    for (uint32_t i = tid; i < A.ntmpl; i += EXPAND_THREADS) s_info[i] = A.info[i];
    __syncthreads();

    const uint64_t proof = blockIdx.y;
    const rec_t *recs = A.recs + proof * A.rec_stride;
    fr_t *out = A.out + proof * A.cell_stride;
    const uint64_t ntiles = (A.nrec + WTILE - 1) / WTILE;
    const u128 two_rb = (u128)1 << A.rb;

    for (uint64_t tile = (uint64_t)blockIdx.x * 4 + wv; tile < ntiles; tile += (uint64_t)gridDim.x * 4) {
        const uint64_t r0 = tile * WTILE;
        const int nr = (int)((A.nrec - r0) < WTILE ? (A.nrec - r0) : WTILE);
        uint32_t n = 0;
        if (lane < nr) {
            const uint64_t m = A.meta[r0 + lane];
            const rec_t rc = recs[r0 + lane];
            const uint32_t t = meta_tmpl(m);
            const tmpl_info_t ti = s_info[t];
            n = ti.ncells;
            s_sbase[wv][lane] = ti.slot_base;
            s_coff[wv][lane] = meta_off(m);
            s_lit[wv][lane] = ~0ull;
            if (t == T_LITERAL) { n = (uint32_t)rc.b; s_lit[wv][lane] = rc.a; }
            u128 V, X0, X1;
            if (ti.mode == M_WIDEV) V = ((u128)rc.b << 64) | rc.a; else V = (u128)rc.a * rc.b + rc.c;
            if (ti.mode == M_LOADW) { X0 = rc.a; X1 = rc.b; }
            else {
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
            b[B_X1] = X1; b[B_X1P] = X1 + two_rb - GL_P; b[B_X1PP] = X1 + two_rb; b[B_W] = X0 * (u128)GL_P + X1;
#pragma unroll
            for (int k = 0; k < B_COUNT; k++) { s_bases[wv][lane][k].lo = (ull)b[k]; s_bases[wv][lane][k].hi = (ull)(b[k] >> 64); }
        }
        uint32_t x = n;
#pragma unroll
        for (int d = 1; d < WTILE; d <<= 1) { uint32_t y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
        s_pre[wv][lane + 1] = x;
        if (lane == 0) s_pre[wv][0] = 0;
        const uint32_t total = __shfl(x, 63, 64);
        __builtin_amdgcn_s_waitcnt(0xc07f);            // this wave's LDS writes are visible to its own later reads (in-order LDS)
        for (uint32_t j = lane; j < total; j += 64) {
            int lo = 0, hi = nr;
#pragma unroll
            for (int it = 0; it < 6; it++) { int mid = (lo + hi) >> 1; if (s_pre[wv][mid] <= j) lo = mid; else hi = mid; }
            const int i = lo;
            const uint32_t s = j - s_pre[wv][i];
            u128s vlo, vhi; vhi.lo = 0; vhi.hi = 0;
            const ull lit = s_lit[wv][i];
            if (lit != ~0ull) { const u128s *src = (const u128s *)(A.pool + lit + s); vlo = src[0]; vhi = src[1]; }
            else {
                const uint32_t d = s_slots[s_sbase[wv][i] + s];
                if (d & 0x80000000u) { vlo = s_consts[(d & 0xffffu) * 2]; vhi = s_consts[(d & 0xffffu) * 2 + 1]; }
                else {
                    const u128s bs = s_bases[wv][i][d & 15u];
                    u128 v = ((u128)bs.hi << 64) | bs.lo;
                    const uint32_t sh = (d >> 4) & 127u, w = (d >> 11) & 255u, ls = (d >> 19) & 127u;
                    v >>= sh;
                    if (w < 128) v &= (((u128)1 << w) - 1);
                    v <<= ls;
                    vlo.lo = (ull)v; vlo.hi = (ull)(v >> 64);
                }
            }
            u128s *dst = (u128s *)(out + s_coff[wv][i] + s);
            dst[0] = vlo; dst[1] = vhi;
        }
    }
}


// ---- variant 2: wave-level tiles, one record per wave iteration (no cell->record search).
// Lane = record derives bases; then for each record r of the tile the 64 lanes write its cells 0..63 (record fields
// are wave-uniform), and the 1-2 leftover cells of the 65/66-cell Goldilocks blocks are written lane-per-record.
__device__ __forceinline__ void eval_cell(uint32_t d, const u128s *bases, const u128s *consts, u128s &vlo, u128s &vhi) {
    vhi.lo = 0; vhi.hi = 0;
    if (d & 0x80000000u) { vlo = consts[(d & 0xffffu) * 2]; vhi = consts[(d & 0xffffu) * 2 + 1]; }
    else {
        const u128s bs = bases[d & 15u];
        u128 v = ((u128)bs.hi << 64) | bs.lo;
        const uint32_t sh = (d >> 4) & 127u, w = (d >> 11) & 255u, ls = (d >> 19) & 127u;
        v >>= sh;
        if (w < 128) v &= (((u128)1 << w) - 1);
        v <<= ls;
        vlo.lo = (ull)v; vlo.hi = (ull)(v >> 64);
    }
}
__global__ __launch_bounds__(EXPAND_THREADS) void expand_kernel_r(ExpandArgs A) {
    __shared__ uint32_t s_slots[MAX_SLOTS];
    __shared__ u128s s_consts[MAX_CONSTS * 2];
    __shared__ tmpl_info_t s_info[T_MAX];
    __shared__ u128s s_bases[4][WTILE][B_COUNT];
    __shared__ uint32_t s_n[4][WTILE];
    __shared__ uint32_t s_sbase[4][WTILE];
    __shared__ ull s_coff[4][WTILE];
    __shared__ ull s_lit[4][WTILE];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (uint32_t i = tid; i < A.nslots; i += EXPAND_THREADS) s_slots[i] = A.slots[i];
    for (uint32_t i = tid; i < A.nconsts * 2; i += EXPAND_THREADS) s_consts[i] = ((const u128s *)A.consts)[i];
    for (uint32_t i = tid; i < A.ntmpl; i += EXPAND_THREADS) s_info[i] = A.info[i];
    __syncthreads();

    const uint64_t proof = blockIdx.y;
    const rec_t *recs = A.recs + proof * A.rec_stride;
    fr_t *out = A.out + proof * A.cell_stride;
    const uint64_t ntiles = (A.nrec + WTILE - 1) / WTILE;
    const u128 two_rb = (u128)1 << A.rb;

    for (uint64_t tile = (uint64_t)blockIdx.x * 4 + wv; tile < ntiles; tile += (uint64_t)gridDim.x * 4) {
        const uint64_t r0 = tile * WTILE;
        const int nr = (int)((A.nrec - r0) < WTILE ? (A.nrec - r0) : WTILE);
        uint32_t n = 0, my_sbase = 0; ull my_coff = 0, my_lit = ~0ull;
        if (lane < nr) {
            const uint64_t m = A.meta[r0 + lane];
            const rec_t rc = recs[r0 + lane];
            const uint32_t t = meta_tmpl(m);
            const tmpl_info_t ti = s_info[t];
            n = ti.ncells; my_sbase = ti.slot_base; my_coff = meta_off(m);
            if (t == T_LITERAL) { n = (uint32_t)rc.b; my_lit = rc.a; }
            u128 V, X0, X1;
            if (ti.mode == M_WIDEV) V = ((u128)rc.b << 64) | rc.a; else V = (u128)rc.a * rc.b + rc.c;
            if (ti.mode == M_LOADW) { X0 = rc.a; X1 = rc.b; }
            else {
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
            b[B_X1] = X1; b[B_X1P] = X1 + two_rb - GL_P; b[B_X1PP] = X1 + two_rb; b[B_W] = X0 * (u128)GL_P + X1;
#pragma unroll
            for (int k = 0; k < B_COUNT; k++) { s_bases[wv][lane][k].lo = (ull)b[k]; s_bases[wv][lane][k].hi = (ull)(b[k] >> 64); }
        }
        s_n[wv][lane] = n; s_sbase[wv][lane] = my_sbase; s_coff[wv][lane] = my_coff; s_lit[wv][lane] = my_lit;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        for (int r = 0; r < nr; r++) {
            const uint32_t nn = s_n[wv][r]; const ull coff = s_coff[wv][r], lit = s_lit[wv][r];
            const bool big = nn > 128 || lit != ~0ull;            // literal runs / long records: full strided loop here
            const uint32_t lim = big ? nn : (nn < 64 ? nn : 64);
            for (uint32_t sidx = lane; sidx < lim; sidx += 64) {
                u128s vlo, vhi;
                if (lit != ~0ull) { const u128s *src = (const u128s *)(A.pool + lit + sidx); vlo = src[0]; vhi = src[1]; }
                else eval_cell(s_slots[s_sbase[wv][r] + sidx], s_bases[wv][r], s_consts, vlo, vhi);
                u128s *dst = (u128s *)(out + coff + sidx);
                dst[0] = vlo; dst[1] = vhi;
            }
        }
        if (my_lit == ~0ull && n > 64 && n <= 128) {                  // leftovers: lane = record
            for (uint32_t sidx = 64; sidx < n; sidx++) {
                u128s vlo, vhi; eval_cell(s_slots[my_sbase + sidx], s_bases[wv][lane], s_consts, vlo, vhi);
                u128s *dst = (u128s *)(out + my_coff + sidx);
                dst[0] = vlo; dst[1] = vhi;
            }
        }
    }
}

// ---- the default kernel for plans without a literal pool (the batched hot path).  Same tiles, same bases, same stores as
// expand_kernel_t, with fewer instructions per cell - overlapped batches are bound by instruction issue, not by HBM (DESIGN.md
// "What bounds the whole job"):
//   * cell -> record without a search: a wavefront's 64 consecutive cells form one aligned 64-cell chunk of the tile; the record lanes
//     leave the record that holds each chunk's first cell in s_hint, so the wavefront reads a wave-uniform [first, last] record range
//     (1-3 records at lookup_bits 21, where a template has <= 66 cells) and each lane counts the record starts at or below its cell;
//   * one 16-byte LDS read per cell for the record's {first cell of the tile, slot base, advice offset};
//   * constants and bases go through the same load / shift / mask sequence (a constant is "the whole 128 bits"), and the usual
//     fields (shift < 64, no left shift) are cut out with 64-bit operations; other descriptors take the generic 128-bit path.
struct __attribute__((aligned(16))) rec_info_t { uint32_t pre, sbase; ull coff; };
static inline size_t hint_words(uint32_t max_cells, int tile_recs) { return ((size_t)tile_recs * max_cells + 63) / 64 + 2; }      // chunk table of a tile
template <bool COLS, int TILE_RECS> __global__ __launch_bounds__(EXPAND_THREADS) void expand_kernel_h(ExpandArgs A) {
    static_assert(TILE_RECS == 32 || TILE_RECS == 64, "the record lanes are one wavefront's");
    u128s *s_consts = reinterpret_cast<u128s *>(s_dyn_tables);
    uint32_t *s_slots = reinterpret_cast<uint32_t *>(s_consts + A.nconsts * 2);
    __shared__ tmpl_info_t s_info[T_MAX];
    __shared__ u128s s_bases[TILE_RECS][B_COUNT];
    __shared__ rec_info_t s_rec[TILE_RECS + 1];
    uint32_t *s_hint = s_slots + A.nslots;                    // (TILE_RECS * max_cells + 63) / 64 + 2 words, dynamic like the tables
    __shared__ uint32_t s_next;
    __shared__ uint32_t s_split[COLS ? TILE_RECS : 1]; __shared__ ull s_dd[COLS ? TILE_RECS : 1];

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
    const bool dyn = A.tile_ctr != nullptr;
    uint64_t tile = blockIdx.x;
    if (dyn) { if (tid == 0) s_next = atomicAdd(&A.tile_ctr[proof], 1u); __syncthreads(); tile = s_next; __syncthreads(); }
    for (; tile < ntiles;) {
        if (dyn && tid == 64) s_next = atomicAdd(&A.tile_ctr[proof], 1u);
        const uint64_t r0 = tile * TILE_RECS;
        const int nr = (int)((A.nrec - r0) < TILE_RECS ? (A.nrec - r0) : TILE_RECS);
        if (tid < TILE_RECS) {
            uint32_t n = 0, sbase = 0; ull coff = 0;
            bool mine = tid < nr;
            if (mine && A.shard_world > 1) {              // unit = (proof, query), round-robin; the prologue block is kept by every rank
                const uint64_t r = r0 + tid;
                if (r >= A.q_rec0_first) {
                    const uint64_t q = r < A.q_rec0_rest ? 0 : 1 + (r - A.q_rec0_rest) / A.q_nrec_rest;
                    mine = (proof * A.nq + q) % A.shard_world == A.shard_rank;
                }
            }
            if (mine) {
                const uint64_t m = A.meta[r0 + tid];
                const rec_t rc = recs[r0 + tid];
                const uint32_t t = meta_tmpl(m);
                const tmpl_info_t ti = s_info[t];
                n = ti.ncells; sbase = ti.slot_base; coff = meta_off(m);
                if constexpr (COLS) {
                    ColCursor cc; cc.init(A.cm); cc.locate(coff);
                    const ull room = cc.hi - coff, d0 = cc.delta;               // cells of this record before the next column starts
                    ull dd = 0; uint32_t split = 0xffffffffu;
                    if (room < 4096) { split = (uint32_t)room; cc.locate(cc.hi); dd = cc.delta - d0; }
                    s_split[tid] = split; s_dd[tid] = dd; coff += d0;
                }
                u128 V, X0, X1;
                if (ti.mode == M_WIDEV) V = ((u128)rc.b << 64) | rc.a; else V = (u128)rc.a * rc.b + rc.c;
                if (ti.mode == M_LOADW) { X0 = rc.a; X1 = rc.b; }
                else {      // hint of GoldilocksChip::reduce (base.rs:349-352), divider-free: see expand_kernel_t
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
                b[B_X1] = X1; b[B_X1P] = X1 + two_rb - GL_P; b[B_X1PP] = X1 + two_rb; b[B_W] = X0 * (u128)GL_P + X1;
#pragma unroll
                for (int k = 0; k < B_COUNT; k++) { s_bases[tid][k].lo = (ull)b[k]; s_bases[tid][k].hi = (ull)(b[k] >> 64); }
            }
            // inclusive scan over the 32 record lanes (wave-level shuffles; lanes 0..31 of wave 0)
            uint32_t x = n;
#pragma unroll
            for (int d = 1; d < TILE_RECS; d <<= 1) { uint32_t y = __shfl_up(x, d, 64); if (tid >= d) x += y; }
            const uint32_t first = x - n;                                  // this record's first cell in the tile
            rec_info_t ri; ri.pre = first; ri.sbase = sbase; ri.coff = coff; s_rec[tid] = ri;
            if (n) for (uint32_t c = (first + 63) >> 6; (c << 6) < x; c++) s_hint[c] = (uint32_t)tid;      // chunks whose first cell is mine (<= 2)
            if (tid == TILE_RECS - 1) { rec_info_t e; e.pre = x; e.sbase = 0; e.coff = 0; s_rec[TILE_RECS] = e; s_hint[(x + 63) >> 6] = (uint32_t)(nr - 1); }
        }
        __syncthreads();
        const uint32_t total = s_rec[TILE_RECS].pre;
        for (uint32_t cb = (uint32_t)(tid & ~63); cb < total; cb += EXPAND_THREADS) {      // cb: first cell of this wavefront's chunk
            const uint32_t j = cb + (uint32_t)(tid & 63);
            const int i0 = __builtin_amdgcn_readfirstlane((int)s_hint[cb >> 6]);
            int ihi = __builtin_amdgcn_readfirstlane((int)s_hint[(cb >> 6) + 1]);
            if (cb + 64 >= total) ihi = nr - 1;
            // records after ihi start at or after the next chunk (or at `total`), so comparing three more starts unconditionally is
            // exact whenever the chunk holds at most four records (templates of 16+ cells); runs of tiny templates take the loop
            int i = i0 + (s_rec[i0 + 1].pre <= j ? 1 : 0) + (s_rec[i0 + 2 <= TILE_RECS ? i0 + 2 : TILE_RECS].pre <= j ? 1 : 0)
                       + (s_rec[i0 + 3 <= TILE_RECS ? i0 + 3 : TILE_RECS].pre <= j ? 1 : 0);
            if (ihi > i0 + 3) for (int k = i0 + 4; k <= ihi; k++) i += (s_rec[k].pre <= j) ? 1 : 0;
            if (j < total) {
                const rec_info_t ri = s_rec[i];
                const uint32_t sidx = j - ri.pre;
                const uint32_t d = s_slots[ri.sbase + sidx];
                const bool is_const = (d >> 31) != 0;
                const u128s *src = is_const ? &s_consts[(d & 0xffffu) * 2] : &s_bases[i][d & 15u];
                const u128s bs = src[0];
                u128s vlo, vhi; vhi.lo = 0; vhi.hi = 0;
                if (is_const) vhi = src[1];
                if (!is_const && (d & ((127u << 19) | (64u << 4)))) {          // left-shifted or far field: generic 128-bit path (rare)
                    u128 v = ((u128)bs.hi << 64) | bs.lo;
                    const uint32_t sh = (d >> 4) & 127u, w = (d >> 11) & 255u, ls = (d >> 19) & 127u;
                    v >>= sh;
                    if (w < 128) v &= (((u128)1 << w) - 1);
                    v <<= ls;
                    vlo.lo = (ull)v; vlo.hi = (ull)(v >> 64);
                } else {
                    const uint32_t sh = is_const ? 0u : (d >> 4) & 63u, w = is_const ? 128u : (d >> 11) & 255u;
                    ull nlo = (bs.lo >> sh) | ((bs.hi << 1) << (63u - sh)), nhi = bs.hi >> sh;
                    const ull m = ((ull)1 << (w & 63u)) - 1;
                    vlo.lo = w < 64 ? (nlo & m) : nlo;
                    vlo.hi = w < 64 ? 0 : (w < 128 ? (nhi & m) : nhi);
                }
                ull cellidx = ri.coff + sidx;
                if constexpr (COLS) { if (sidx >= s_split[i]) cellidx += s_dd[i]; }
                u128s *dst = (u128s *)(out + cellidx);
                dst[0] = vlo; dst[1] = vhi;
            }
        }
        __syncthreads();
        tile = dyn ? (uint64_t)s_next : tile + gridDim.x;
        if (dyn) __syncthreads();          // s_next is rewritten at the top of the next tile
    }
}

static int expand_variant() { static int v = -1; if (v < 0) { const char *e = getenv("H2W_EXPAND_VARIANT"); v = e ? atoi(e) : 0; } return v; }

void launch_expand(const ExpandArgs &A, uint64_t nproofs, int grid_x, hipStream_t stream) {
    if (A.nrec == 0 || nproofs == 0) return;
    const int TILE_RECS = expand_variant() == 16 ? 16 : (expand_variant() == 64 || (expand_variant() == 364 && !A.cm.starts)) ? 64 : (expand_variant() == 1 || expand_variant() == 2) ? 64 : 32;
    uint64_t ntiles = (A.nrec + TILE_RECS - 1) / TILE_RECS;
    { static int ov = -2; if (ov == -2) { const char *e = getenv("H2W_EXPAND_BLOCKS"); ov = e ? atoi(e) : -1; } if (ov > 0) grid_x = (int)((uint64_t)ov / nproofs) + 1; }
    uint64_t gx = (uint64_t)grid_x; if (gx > ntiles) gx = ntiles; if (gx < 1) gx = 1;
    dim3 grid((unsigned)gx, (unsigned)nproofs);
    const size_t dyn = ((size_t)A.nconsts * 32 + (size_t)A.nslots * 4 + 15) & ~(size_t)15;      // expand_kernel_t's tables
    const bool hinted = A.pool == nullptr && A.max_cells > 0 && A.max_cells <= 512 && (expand_variant() == 3 || expand_variant() == 364);     // (literal runs can be longer than a chunk table: expand_kernel_t keeps those)
    const size_t dyn_h = ((size_t)A.nconsts * 32 + ((size_t)A.nslots + hint_words(A.max_cells, TILE_RECS)) * 4 + 15) & ~(size_t)15;
    if (hinted && A.cm.starts) hipLaunchKernelGGL((expand_kernel_h<true, 32>), grid, dim3(EXPAND_THREADS), dyn_h, stream, A);
    else if (hinted && TILE_RECS == 64) hipLaunchKernelGGL((expand_kernel_h<false, 64>), grid, dim3(EXPAND_THREADS), dyn_h, stream, A);
    else if (hinted) hipLaunchKernelGGL((expand_kernel_h<false, 32>), grid, dim3(EXPAND_THREADS), dyn_h, stream, A);
    else if (A.cm.starts) hipLaunchKernelGGL((expand_kernel_t<0, 32, 5, true>), grid, dim3(EXPAND_THREADS), dyn, stream, A);      // the A/B variants write flat only
    else if (expand_variant() == 2) hipLaunchKernelGGL(expand_kernel_r, grid, dim3(EXPAND_THREADS), 0, stream, A);
    else if (expand_variant() == 1) hipLaunchKernelGGL(expand_kernel_w, grid, dim3(EXPAND_THREADS), 0, stream, A);
    else if (expand_variant() == 11) hipLaunchKernelGGL((expand_kernel_t<1, 32, 5>), grid, dim3(EXPAND_THREADS), dyn, stream, A);
    else if (expand_variant() == 12) hipLaunchKernelGGL((expand_kernel_t<2, 32, 5>), grid, dim3(EXPAND_THREADS), dyn, stream, A);
    else if (expand_variant() == 64) hipLaunchKernelGGL((expand_kernel_t<0, 64, 6>), grid, dim3(EXPAND_THREADS), dyn, stream, A);
    else if (expand_variant() == 16) hipLaunchKernelGGL((expand_kernel_t<0, 16, 4>), grid, dim3(EXPAND_THREADS), dyn, stream, A);
    else hipLaunchKernelGGL((expand_kernel_t<0, 32, 5>), grid, dim3(EXPAND_THREADS), dyn, stream, A);
}

}  // namespace h2w
