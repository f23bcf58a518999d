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
template <int TILE_RECS, int NSTEP, bool COLS = false> __global__ __launch_bounds__(EXPAND_THREADS) void expand_kernel_t(ExpandArgs A) {
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
            if (mine && A.shard_world > 1) {              // unit = (proof, query), round-robin; the prologue block belongs to rank proof mod world
                const uint64_t r = r0 + tid;
                if (r >= A.q_rec0_first) {
                    const uint64_t q = r < A.q_rec0_rest ? 0 : 1 + (r - A.q_rec0_rest) / A.q_nrec_rest;
                    mine = (proof * A.nq + q) % A.shard_world == A.shard_rank;
                } else mine = proof % A.shard_world == A.shard_rank;      // prologue block: the proof's owner
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
#pragma unroll
            for (int it = 0; it < NSTEP; it++) { int mid = (lo + hi) >> 1; if (s_pre[mid] <= j) lo = mid; else hi = mid; }
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
            ull cellidx = s_coff[i] + s;
            if constexpr (COLS) { if (s >= s_split[i]) cellidx += s_dd[i]; }
            u128s *dst = (u128s *)(out + cellidx);
            dst[0] = vlo; dst[1] = vhi;
        }
        __syncthreads();
        tile = dyn ? (uint64_t)s_next : tile + gridDim.x;
        if (dyn) __syncthreads();          // s_next is rewritten at the top of the next tile
    }
}



// ---------------------------------------------------------------------------------------------------------------
// expand_fast: the batched hot path's expansion kernel.  Same records, same cells as expand_kernel_t, ~8x fewer
// instructions per cell: the generic kernel decodes a slot descriptor per CELL (cell -> record search, descriptor
// fetch, variable 128-bit shifts); here one lane owns one RECORD and the cell index is a compile-time constant.
//
//   * Every Goldilocks-level template is a sub-range [vs, ve) of ONE virtual cell list (L = lookup_bits,
//     NL = ceil(64 / L) limbs, RB = NL * L, RC = 3 NL - 2 cells of range_check(x, RB), LW = 1 + RC + 7 + RC):
//         0: prefix (A or B)   1..4: gate [C, A, B, V]   5..5+LW: load_witness(q)   ..+LW: load_witness(r)
//         then [p] [r, q, p, q p + r]                                  (GoldilocksChip::reduce, base.rs:346-368)
//     T_KA/KB_GLOP [0, VT), T_GLOP [1, VT), T_REDUCE [5, VT), T_LOADW [5, 5+LW), T_LOADW2 [5, 5+2LW),
//     T_CLT_SAFE [6, 5+LW), T_GATE [1, 5), T_KB_GATE [0, 5), T_CONST1 [2, 3).
//   * A wavefront takes a tile of 64 consecutive records (lane = record), derives q, r, q+2^RB-p, ... once, then walks the
//     virtual cells eight at a time: each lane puts the low 16 bytes of its eight cells into a per-wavefront LDS tile
//     (144-byte rows: conflict-free ds_write_b128), and the wavefront flushes the tile with lane = (record, 16-byte piece): one
//     ds_read_b128 and one 16-byte store per lane, 256 contiguous bytes per record and store instruction (the write path
//     is bound by distinct cache lines per store instruction, not by bytes: record-contiguous runs keep it at >= 90 bytes per
//     line touched; the high halves are zero except for the one constant -2^RB and come from a 32-byte LDS block).
//     LDS orders a wavefront's own accesses, so there is no barrier anywhere in the loop.
//   * The two irregular fixed templates (T_CONST4, T_REP12; < 1 % of the cells) are written by their own lane.
// Tiles are handed out by a per-proof counter.  Flat layout only (column-major emission keeps the generic kernel).
template <int L> struct FastMap {
    static constexpr int NL = (64 + L - 1) / L, RB = NL * L, RC = 3 * NL - 2, LW = 1 + RC + 7 + RC, VT = 5 + 2 * LW + 5, NCH = (VT + 3) / 4;
    static_assert(NL >= 2 && RB >= 64 && RB < 128, "lookup_bits outside the fast kernel's range");
};
struct FastBases { ull pre, A, B, C, Vlo, Vhi, q, r, qplo, qphi, qpplo, qpphi, rplo, rphi, rpplo, rpphi, Wlo, Whi; };
struct lohi_t { ull lo, hi; };

template <int L> __device__ __forceinline__ lohi_t fast_rc_cell(int p, ull lo, ull hi) {      // cell p of range_check(x, RB), x = hi:lo
    const ull LM = (1ull << L) - 1;
    if (p == 0) return lohi_t{lo & LM, 0};
    const int j = (p + 2) / 3, m = (p + 2) % 3, sh = j * L, w = (j + 1) * L;
    if (m == 0) {                                           // limb j
        ull v;
        if (sh + L <= 64) v = lo >> sh; else if (sh >= 64) v = hi >> (sh - 64); else v = (lo >> sh) | (hi << (64 - sh));
        return lohi_t{v & LM, 0};
    }
    if (m == 1) return lohi_t{1ull << sh, 0};               // 2^(jL): (NL - 1) L < 64
    if (w >= 128) return lohi_t{lo, hi};                    // x mod 2^((j+1)L)
    if (w < 64) return lohi_t{lo & ((1ull << w) - 1), 0};
    if (w == 64) return lohi_t{lo, 0};
    return lohi_t{lo, hi & ((1ull << (w - 64)) - 1)};
}
template <int L> __device__ __forceinline__ lohi_t fast_loadw_cell(int i, ull x, ull xplo, ull xphi, ull xpplo, ull xpphi, ull neglo, ull neghi) {
    typedef FastMap<L> M;
    if (i == 0) return lohi_t{x, 0};
    if (i <= M::RC) return fast_rc_cell<L>(i - 1, x, 0);
    const int k = i - 1 - M::RC;
    if (k == 0) return lohi_t{xplo, xphi};
    if (k == 1) return lohi_t{GL_P, 0};
    if (k == 2 || k == 5) return lohi_t{1, 0};
    if (k == 3) return lohi_t{xpplo, xpphi};
    if (k == 4) return lohi_t{neglo, neghi};
    if (k == 6) return lohi_t{x, 0};
    return fast_rc_cell<L>(k - 7, xplo, xphi);
}
template <int L> __device__ __forceinline__ lohi_t fast_vcell(int v, const FastBases &b, ull neglo, ull neghi) {
    typedef FastMap<L> M;
    if (v == 0) return lohi_t{b.pre, 0};
    if (v == 1) return lohi_t{b.C, 0};
    if (v == 2) return lohi_t{b.A, 0};
    if (v == 3) return lohi_t{b.B, 0};
    if (v == 4) return lohi_t{b.Vlo, b.Vhi};
    if (v < 5 + M::LW) return fast_loadw_cell<L>(v - 5, b.q, b.qplo, b.qphi, b.qpplo, b.qpphi, neglo, neghi);
    if (v < 5 + 2 * M::LW) return fast_loadw_cell<L>(v - 5 - M::LW, b.r, b.rplo, b.rphi, b.rpplo, b.rpphi, neglo, neghi);
    const int k = v - 5 - 2 * M::LW;
    if (k == 0 || k == 3) return lohi_t{GL_P, 0};
    if (k == 1) return lohi_t{b.r, 0};
    if (k == 2) return lohi_t{b.q, 0};
    if (k == 4) return lohi_t{b.Wlo, b.Whi};
    return lohi_t{0, 0};
}
template <int L> constexpr bool fast_is_neg(int v) {        // virtual cells holding the 254-bit constant -2^RB (mod r)
    return v == 5 + 1 + FastMap<L>::RC + 4 || v == 5 + FastMap<L>::LW + 1 + FastMap<L>::RC + 4;
}
constexpr int FAST_CH = 8;                                  // virtual cells per flush step (256 B per record and store instruction)
#ifndef H2W_FAST_T
#define H2W_FAST_T 16
#endif
#ifndef H2W_FAST_K
#define H2W_FAST_K 4
#endif
constexpr int FAST_K = H2W_FAST_K;                          // 64-record rows per fetch of a wavefront (its work unit: 256 records)
constexpr int FAST_T = H2W_FAST_T;                          // records per pass of a wavefront (LDS per block scales with it: 73 KB at 16)
// Work unit `tile` of a proof -> its records [rbeg, rend) and the pointer its cells' in-proof offsets are added to.  Unsharded: the proof is
// one block of nrec records.  Sharded: the units walk the blocks this rank owns - the prologue block first (if owned), then
// the owned query blocks q0, q0 + world, ... - so a rank streams 1 / world of the records and nothing else (tiles past its last
// block: false).
constexpr uint32_t FAST_TR = 64 * FAST_K;
__device__ __forceinline__ bool fast_tile_range(const ExpandArgs &A, uint64_t proof, uint32_t tile, uint32_t q_tiles, uint64_t &rbeg, uint64_t &rend, fr_t *&outp) {
    outp = A.out + proof * A.cell_stride;
    if (A.shard_world <= 1) { rbeg = (uint64_t)tile * FAST_TR; rend = rbeg + FAST_TR < A.nrec ? rbeg + FAST_TR : A.nrec; return rbeg < A.nrec; }
    const uint64_t W = A.shard_world, r = A.shard_rank, u0 = proof * A.nq;
    const bool own_pro = proof % W == r;
    const uint32_t tp = own_pro ? (uint32_t)((A.pro_nrec + FAST_TR - 1) / FAST_TR) : 0u;
    const uint64_t units_before = (u0 + W - 1 - r) / W;
    uint64_t local = ((proof + W - 1 - r) / W) * A.pro_ncell + units_before * A.q_slot;      // the rank's packed buffer: owned blocks before this proof
    if (tile < tp) {
        rbeg = (uint64_t)tile * FAST_TR; rend = rbeg + FAST_TR < A.pro_nrec ? rbeg + FAST_TR : A.pro_nrec;
        if (A.shard_compact) outp = A.out + local;
        return true;
    }
    const uint32_t t2 = tile - tp, j = t2 / q_tiles, tt = t2 % q_tiles;
    const uint64_t q = (r + W - u0 % W) % W + (uint64_t)j * W;                                   // this rank's j-th query of the proof
    if (q >= A.nq) return false;
    const uint64_t b0 = q == 0 ? A.q_rec0_first : A.q_rec0_rest + (q - 1) * A.q_nrec_rest, bn = q == 0 ? A.q_nrec_first : A.q_nrec_rest;
    rbeg = b0 + (uint64_t)tt * FAST_TR; rend = rbeg + FAST_TR < b0 + bn ? rbeg + FAST_TR : b0 + bn;
    if (rbeg >= rend) return false;
    if (A.shard_compact) {
        local += (own_pro ? A.pro_ncell : 0) + (uint64_t)j * A.q_slot;
        outp = A.out + local - (q == 0 ? A.q_cell0_first : A.q_cell0_rest + (q - 1) * A.q_ncell_rest);
    }
    return true;
}
constexpr int FAST_MAX_COLS = 64;                           // column-major emission through the fast kernel: the column table lives in LDS
template <int L, bool ROAM, bool COLS> __global__ __launch_bounds__(EXPAND_THREADS, 2) void expand_fast(ExpandArgs A) {
    typedef FastMap<L> M;
    constexpr int NCH = (M::VT + FAST_CH - 1) / FAST_CH;
    constexpr int ROW = M::VT * 16 + 16;                      // bytes per record row of the LDS tile: the low halves of its virtual cells (+ padding); the last step's lanes past VT never read
    constexpr int TILE = (FAST_T + 1) * ROW;                  // + one row of HIGH halves: zeros, and the high half of -2^RB at its two cells (what the odd pieces store)
    __shared__ __attribute__((aligned(16))) unsigned char s_tile[EXPAND_THREADS / 64][TILE];
    __shared__ uint2 s_vbr[EXPAND_THREADS / 64][FAST_T];     // per record: {flat index of its virtual cell 0, vs | ve << 8}
    // column-major emission: cell i of the flat stream lives at i + delta(column of i), a piecewise-constant shift.  A record's cells
    // are contiguous and cross at most one column boundary: per record the shift of its first cell, the flat index where the next column starts
    // (none: 2^32 - 1) and the change of the shift there.  The column starts are staged in LDS: a vector load here would wait for every cell store
    // the wavefront has in flight.
    __shared__ ull s_cstart[COLS ? FAST_MAX_COLS : 1];
    __shared__ ull s_cd0[EXPAND_THREADS / 64][COLS ? FAST_T : 1], s_cdd[EXPAND_THREADS / 64][COLS ? FAST_T : 1]; __shared__ uint32_t s_csplit[EXPAND_THREADS / 64][COLS ? FAST_T : 1];
    if constexpr (COLS) { for (uint32_t i = threadIdx.x; i < A.cm.ncols; i += EXPAND_THREADS) s_cstart[i] = A.cm.starts[i]; __syncthreads(); }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // which proof: one column of the grid per proof (static), or - roam - a grid of as many blocks as the chip holds at once, whose wavefronts
    // start spread over the proofs and move on to the next proof that has tiles left.  With more blocks than fit, the late ones start when
    // the first proofs are done and ramp up again: 5.5-5.8 TB/s for a launch alone against 6.3 with every block resident from the start
    // (profiles/r02_expand_grid.txt).
    uint64_t proof = ROAM ? (uint64_t)((blockIdx.x * (EXPAND_THREADS / 64) + wv) % A.nproofs) : (uint64_t)blockIdx.y;
    const uint32_t ntiles = A.ntiles, q_tiles = A.q_tiles;      // work units a wavefront takes from its proof's counter (sharded: an upper bound; the units past a proof's last owned block are empty)
    // -2^RB mod r (the one 254-bit constant of check_less_than): limbs of r with bit RB taken out of limb 1 (no borrow)
    const ull neg0 = H2W_FR_M0, neg1 = H2W_FR_M1 - (1ull << (M::RB - 64)), neg2 = H2W_FR_M2, neg3 = H2W_FR_M3;
    unsigned char *my_row = &s_tile[wv][(lane & (FAST_T - 1)) * ROW];
    // flush roles: lane -> 16-byte piece (lane % 16) of a 256-byte step of record lane / 16 of a group of four records
    const int piece = lane & 15, kc = piece >> 1; const bool hi = piece & 1;
    const unsigned char *rd_base = &s_tile[wv][(lane >> 4) * ROW + kc * 16];
    const unsigned char *hi_base = &s_tile[wv][FAST_T * ROW + kc * 16];
    for (int v = lane; v < M::VT; v += 64)
        *reinterpret_cast<u128s *>(&s_tile[wv][FAST_T * ROW + v * 16]) = fast_is_neg<L>(v) ? u128s{neg2, neg3} : u128s{0, 0};

  for (;;) {
    const rec_t *recs = A.recs + proof * A.rec_stride;
    uint32_t tile;
    { uint32_t t0 = 0; if (lane == 0) t0 = atomicAdd(&A.tile_ctr[proof], 1u); tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)t0); }
    while (tile < ntiles) {
        uint32_t nxt = 0; if (lane == 0) nxt = atomicAdd(&A.tile_ctr[proof], 1u);       // next tile: requested now, read at the bottom
        uint64_t sbase, rend; fr_t *outp;
        if (!fast_tile_range(A, proof, tile, q_tiles, sbase, rend, outp)) { tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)nxt); continue; }
        unsigned char *outb = reinterpret_cast<unsigned char *>(outp);
        unsigned char *const out_half = outb + (hi ? 16 : 0);
        // The records of FAST_K * 64 consecutive records in one fetch, one record (+ meta) per lane and 64-record row; the passes take theirs
        // from those registers with cross-lane reads.  Loads and stores retire in order on this family, so every vector load a wavefront
        // consumes costs it a wait for all the cell stores it has in flight: one such wait per 16 passes instead of one per pass
        // (profiles/r02_expand_grid.txt: a timing build without the per-pass loads ran the whole job 9 % faster).
        uint64_t mk[FAST_K]; rec_t rk[FAST_K];
#pragma unroll
        for (int j = 0; j < FAST_K; j++) {
            const uint64_t r = sbase + (uint64_t)j * 64 + (uint64_t)lane;
            mk[j] = 0; rk[j].a = rk[j].b = rk[j].c = rk[j].d = 0;
            if (r < rend) { mk[j] = A.meta[r]; rk[j] = recs[r]; }
        }
#pragma unroll 1
        for (int pass = 0; pass < FAST_K * (64 / FAST_T); pass++) {
            const int fj = pass / (64 / FAST_T), fq = pass % (64 / FAST_T);
            const uint64_t r0 = sbase + (uint64_t)fj * 64 + (uint64_t)fq * FAST_T;
            if (r0 >= rend) break;
            uint64_t cm = mk[0]; rec_t cr = rk[0];
#pragma unroll
            for (int j = 1; j < FAST_K; j++) if (fj == j) { cm = mk[j]; cr.a = rk[j].a; cr.b = rk[j].b; cr.c = rk[j].c; cr.d = rk[j].d; }      // (wave-uniform selects)
            const int src = fq * FAST_T + (lane & (FAST_T - 1));
            const uint64_t pm = __shfl((unsigned long long)cm, src, 64);
            rec_t prc; prc.a = __shfl((unsigned long long)cr.a, src, 64); prc.b = __shfl((unsigned long long)cr.b, src, 64);
            prc.c = __shfl((unsigned long long)cr.c, src, 64); prc.d = __shfl((unsigned long long)cr.d, src, 64);
            const bool mine = lane < FAST_T && r0 + (uint64_t)lane < rend;
            uint32_t t = T_LITERAL; rec_t rc; rc.a = rc.b = rc.c = rc.d = 0; ull coff = 0;
            if (mine) { rc = prc; t = meta_tmpl(pm); coff = meta_off(pm); }
            // template -> range of virtual cells
            int vs = 0, ve = 0; FastBases b; b.pre = 0;
            const bool glop = t == T_GLOP || t == T_KA_GLOP || t == T_KB_GLOP;
            if (glop) { vs = t == T_GLOP ? 1 : 0; ve = M::VT; b.pre = t == T_KA_GLOP ? rc.a : rc.b; }
            else if (t == T_REDUCE) { vs = 5; ve = M::VT; }
            else if (t == T_LOADW) { vs = 5; ve = 5 + M::LW; }
            else if (t == T_LOADW2) { vs = 5; ve = 5 + 2 * M::LW; }
            else if (t == T_CLT_SAFE) { vs = 6; ve = 5 + M::LW; }
            else if (t == T_GATE) { vs = 1; ve = 5; }
            else if (t == T_KB_GATE) { vs = 0; ve = 5; b.pre = rc.b; }
            else if (t == T_CONST1) { vs = 2; ve = 3; }
            const bool slow = mine && (t == T_CONST4 || t == T_REP12);
            if (lane < FAST_T) {
                {   // bases (expand_kernel_t's record phase, per lane)
                    u128 V, X0, X1;
                    if (t == T_REDUCE) V = ((u128)rc.b << 64) | rc.a; else V = (u128)rc.a * rc.b + rc.c;
                    if (!(glop || t == T_REDUCE)) { X0 = rc.a; X1 = rc.b; }
                    else {      // hint of GoldilocksChip::reduce (base.rs:349-352), divider-free: see expand_kernel_t
                        const uint64_t r = gl_reduce128(V);
                        const u128 Dv = V - r;
                        const uint64_t dl = (uint64_t)Dv;
                        const uint64_t qlo = dl + (dl << 32);
                        const uint64_t qhi = ((u128)qlo * GL_P != Dv) ? 1 : 0;
                        X0 = gl_reduce128(((u128)qhi << 64) | qlo); X1 = r;
                    }
                    const u128 two_rb = (u128)1 << M::RB;
                    const u128 x0p = X0 + two_rb - GL_P, x0pp = X0 + two_rb, x1p = X1 + two_rb - GL_P, x1pp = X1 + two_rb, W = X0 * (u128)GL_P + X1;
                    b.A = rc.a; b.B = rc.b; b.C = rc.c; b.Vlo = (ull)V; b.Vhi = (ull)(V >> 64); b.q = (ull)X0; b.r = (ull)X1;
                    b.qplo = (ull)x0p; b.qphi = (ull)(x0p >> 64); b.qpplo = (ull)x0pp; b.qpphi = (ull)(x0pp >> 64);
                    b.rplo = (ull)x1p; b.rphi = (ull)(x1p >> 64); b.rpplo = (ull)x1pp; b.rpphi = (ull)(x1pp >> 64);
                    b.Wlo = (ull)W; b.Whi = (ull)(W >> 64);
                }
                s_vbr[wv][lane] = make_uint2((uint32_t)coff - (uint32_t)vs, (uint32_t)vs | ((uint32_t)ve << 8));      // (index mod 2^32: a proof's stream is shorter)
                if constexpr (COLS) {
                    uint32_t a = 0, b = A.cm.ncols;                // largest c with starts[c] <= coff
                    while (b - a > 1) { const uint32_t mid = (a + b) >> 1; if (s_cstart[mid] <= coff) a = mid; else b = mid; }
                    const ull d0 = ((ull)a << A.cm.k) - s_cstart[a], ncell = slow ? (t == T_CONST4 ? 4 : 12) : (ull)(ve - vs);
                    const bool cross = a + 1 < A.cm.ncols && coff + ncell > s_cstart[a + 1];
                    s_cd0[wv][lane] = d0; s_csplit[wv][lane] = cross ? (uint32_t)s_cstart[a + 1] : 0xffffffffu;
                    s_cdd[wv][lane] = cross ? ((ull)(a + 1) << A.cm.k) - s_cstart[a + 1] - d0 : 0;
                }
#pragma clang loop unroll(full)
                for (int c = 0; c < NCH; c++) {
#pragma clang loop unroll(full)
                    for (int k = 0; k < FAST_CH; k++) {
                        if (FAST_CH * c + k >= M::VT) continue;
                        const lohi_t x = fast_vcell<L>(FAST_CH * c + k, b, neg0, neg1);
                        *reinterpret_cast<u128s *>(my_row + (FAST_CH * c + k) * 16) = u128s{x.lo, x.hi};
                    }
                }
            }
            // flush, record-major: a group of four records is written out completely (NCH steps of 256 B per record) before the next group -
            // a wavefront's stores must sweep memory sequentially (tools/ubench/ubench_store6.hip: revisiting a record later costs a third of the rate)
#pragma unroll 1
            for (int g = 0; g < FAST_T / 4; g++) {
                const uint2 br = s_vbr[wv][g * 4 + (lane >> 4)];
                const int vs_ = (int)(br.y & 255u), ve_ = (int)(br.y >> 8);
                // steps in which this lane's cell 8 c + kc is one of the record's: c in [c_lo, c_hi]
                const int c_lo = (vs_ - kc + 7) >> 3, c_hi = ve_ - 1 - kc >= 0 ? (ve_ - 1 - kc) >> 3 : -1;
                const unsigned char *rd = hi ? hi_base : rd_base + g * 4 * ROW;                   // low halves from the record's row, high halves from the template row
                // (br.x = cell index of virtual cell 0, mod 2^32: negative when the stream starts inside the record's virtual list, e.g. a LOADW
                //  block at cell 0; the steps that are stored, c >= c_lo, land at or after the record's first real cell)
                long long first = (long long)(uint32_t)(br.x + (uint32_t)kc + 16u) - 16;      // flat index of this lane's cell of step 0 (>= -8; streams of up to 2^32 - 32 cells)
                uint32_t split = 0xffffffffu; ull dd = 0;
                if constexpr (COLS) { const int ri = g * 4 + (lane >> 4); first += (long long)s_cd0[wv][ri]; split = s_csplit[wv][ri]; dd = s_cdd[wv][ri]; }
                unsigned char *const dst = out_half + first * 32;       // + 256 c per step: an immediate offset
                if (COLS && __any(split != 0xffffffffu)) {          // a record of the group runs into the next column (once per column and proof): per-cell shifts
                    const long long flat0 = (long long)(uint32_t)(br.x + (uint32_t)kc + 16u) - 16;
#pragma unroll 1
                    for (int c = 0; c < NCH; c++)
                        if (c >= c_lo && c <= c_hi) {
                            const bool over = split != 0xffffffffu && flat0 + 8 * c >= (long long)split;
                            *reinterpret_cast<u128s *>(dst + c * FAST_CH * 32 + (over ? (long long)dd * 32 : 0)) = *reinterpret_cast<const u128s *>(rd + c * FAST_CH * 16);
                        }
                } else if (__all(c_lo <= 1 && c_hi >= NCH - 2)) {       // four whole Goldilocks-op blocks (the usual case): only the first and the last step are ragged
                    if (c_lo <= 0) *reinterpret_cast<u128s *>(dst) = *reinterpret_cast<const u128s *>(rd);
#pragma unroll
                    for (int c = 1; c < NCH - 1; c++) *reinterpret_cast<u128s *>(dst + c * FAST_CH * 32) = *reinterpret_cast<const u128s *>(rd + c * FAST_CH * 16);
                    if (c_hi >= NCH - 1) *reinterpret_cast<u128s *>(dst + (NCH - 1) * FAST_CH * 32) = *reinterpret_cast<const u128s *>(rd + (NCH - 1) * FAST_CH * 16);
                } else {
#pragma unroll
                    for (int c = 0; c < NCH; c++)
                        if (c >= c_lo && c <= c_hi) *reinterpret_cast<u128s *>(dst + c * FAST_CH * 32) = *reinterpret_cast<const u128s *>(rd + c * FAST_CH * 16);
                }
            }
            if (__any(slow)) {
                if (slow) {
                    const u128s z{0, 0}; ull d0 = 0, dd = 0; uint32_t split = 0xffffffffu;
                    if constexpr (COLS) { d0 = s_cd0[wv][lane]; dd = s_cdd[wv][lane]; split = s_csplit[wv][lane]; }
                    auto at = [&](int i) { return reinterpret_cast<u128s *>(outb + (coff + (ull)i + d0 + ((COLS && coff + (ull)i >= split) ? dd : 0)) * 32); };
                    if (t == T_CONST4) { const ull w4[4] = {rc.a, rc.b, rc.c, rc.d}; for (int i = 0; i < 4; i++) { u128s *d = at(i); d[0] = u128s{w4[i], 0}; d[1] = z; } }
                    else { for (int i = 0; i < 12; i++) { u128s *d = at(i); d[0] = u128s{rc.a, 0}; d[1] = z; } }
                }
            }
        }
        tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)nxt);
    }
    if (!ROAM) break;
    // the next proof (cyclically) whose counter says tiles are left: 64 counters per look.  A stale or racing read only costs a visit.
    int nx = -1;
    for (uint32_t base = 0; base < A.nproofs && nx < 0; base += 64) {
        const uint32_t k = base + (uint32_t)lane; uint32_t idx = (uint32_t)proof + 1 + k; idx %= A.nproofs;
        const uint32_t c = k < A.nproofs ? __hip_atomic_load(&A.tile_ctr[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
        const unsigned long long m = __ballot(c < ntiles);
        if (m) nx = (int)(((uint32_t)proof + 1 + base + (uint32_t)(__ffsll((long long)m) - 1)) % A.nproofs);
    }
    if (nx < 0) break;
    proof = (uint64_t)__builtin_amdgcn_readfirstlane(nx);
  }
}

// every template id a batched plan emits is fixed (< T_DYNAMIC; dynamic range-check templates and literal runs are eager-context
// features), so the fast kernel serves the batched path whenever it is instantiated for the plan's lookup_bits
int launch_expand(const ExpandArgs &A, uint64_t nproofs, int grid_x, hipStream_t stream) {
    if (A.nrec == 0 || nproofs == 0) return 0;
    const bool fast_ok = A.tile_ctr != nullptr && A.pool == nullptr && (A.cm.starts == nullptr || (A.cm.ncols <= (uint32_t)FAST_MAX_COLS && A.shard_world <= 1)) && A.ntmpl <= T_DYNAMIC;
    if (fast_ok && (A.lookup_bits == 21 || A.lookup_bits == 13 || A.lookup_bits == 8)) {
        ExpandArgs B = A; B.nproofs = (uint32_t)nproofs; B.roam = 0;
        // work units per proof: all of its records, or - sharded - the prologue block plus ceil(nq / world) query blocks (an upper bound per proof)
        uint64_t ntiles = (A.nrec + FAST_TR - 1) / FAST_TR;
        if (A.shard_world > 1) {
            const uint64_t qn = A.q_nrec_first > A.q_nrec_rest ? A.q_nrec_first : A.q_nrec_rest;
            B.q_tiles = (uint32_t)((qn + FAST_TR - 1) / FAST_TR); if (B.q_tiles < 1) B.q_tiles = 1;
            ntiles = (A.pro_nrec + FAST_TR - 1) / FAST_TR + (uint64_t)B.q_tiles * ((A.nq + A.shard_world - 1) / A.shard_world);
        }
        B.ntiles = (uint32_t)ntiles;
        uint64_t gx = (uint64_t)grid_x; if (gx * (EXPAND_THREADS / 64) > ntiles) gx = (ntiles + EXPAND_THREADS / 64 - 1) / (EXPAND_THREADS / 64); if (gx < 1) gx = 1;
        dim3 grid((unsigned)gx, (unsigned)nproofs);
        // roaming wavefronts: a batch of proofs with many tiles each (the witness path); many small instances (h2w_chipbatch) keep one column each
        const uint64_t waves = EXPAND_THREADS / 64;
        const bool roam_ok = A.roam_per_cu > 0 && ntiles >= 16 && nproofs <= 4096 && grid_x > 0;
        if (roam_ok) {
            static int cus[64] = {0}; int dev = 0; (void)hipGetDevice(&dev);
            if (dev >= 0 && dev < 64 && !cus[dev]) { int n = 0; (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); cus[dev] = n > 0 ? n : 256; }
            const int ncu = (dev >= 0 && dev < 64) ? cus[dev] : 256;
            int per_cu = A.lookup_bits == 21 ? (FAST_T <= 8 ? 4 : 2) : 1;        // blocks of this kernel one CU holds (LDS)
            if (A.roam_per_cu > 0 && (int)A.roam_per_cu < per_cu) per_cu = (int)A.roam_per_cu;
            uint64_t nb = (uint64_t)ncu * per_cu;
            const uint64_t want = (ntiles * nproofs + waves - 1) / waves; if (nb > want) nb = want;
            if (nb < 1) nb = 1;
            B.roam = 1; grid = dim3((unsigned)nb, 1);
        }
        const bool cols = A.cm.starts != nullptr;
#define H2W_LAUNCH_FAST(LB, RM, CL) hipLaunchKernelGGL((expand_fast<LB, RM, CL>), grid, dim3(EXPAND_THREADS), 0, stream, B)
#define H2W_LAUNCH_FAST_L(LB) do { if (B.roam) { if (cols) H2W_LAUNCH_FAST(LB, true, true); else H2W_LAUNCH_FAST(LB, true, false); } \
                                   else { if (cols) H2W_LAUNCH_FAST(LB, false, true); else H2W_LAUNCH_FAST(LB, false, false); } } while (0)
        if (A.lookup_bits == 21) H2W_LAUNCH_FAST_L(21); else if (A.lookup_bits == 13) H2W_LAUNCH_FAST_L(13); else H2W_LAUNCH_FAST_L(8);
#undef H2W_LAUNCH_FAST_L
#undef H2W_LAUNCH_FAST
        return 0;
    }
    if (A.shard_compact) { set_error("expansion: the packed shard layout needs the fast expansion kernel (flat layout, lookup_bits 21 / 13 / 8)"); return -1; }
    const int TILE_RECS = 32;
    const uint64_t ntiles = (A.nrec + TILE_RECS - 1) / TILE_RECS;
    uint64_t gx = (uint64_t)grid_x; if (gx > ntiles) gx = ntiles; if (gx < 1) gx = 1;
    const dim3 grid((unsigned)gx, (unsigned)nproofs);
    const size_t dyn = ((size_t)A.nconsts * 32 + (size_t)A.nslots * 4 + 15) & ~(size_t)15;      // expand_kernel_t's tables
    if (A.cm.starts) hipLaunchKernelGGL((expand_kernel_t<32, 5, true>), grid, dim3(EXPAND_THREADS), dyn, stream, A);
    else hipLaunchKernelGGL((expand_kernel_t<32, 5, false>), grid, dim3(EXPAND_THREADS), dyn, stream, A);
    return 0;
}

}  // namespace h2w
