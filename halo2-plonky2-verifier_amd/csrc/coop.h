// coop.h — wavefront-cooperative value kernels (device only).
//
// CoopSink: one 64-lane wavefront executes one strand of verifier.h *wave-uniformly* (every lane runs the same
// gadget code on the same values; lane 0 alone writes records / direct cells), and the lanes split up the one
// wide thing in it: the Goldilocks Poseidon permutation (hash/poseidon/permutation.rs:43-284), 12 lanes = the 12
// state elements / MDS rows.
//
// A strand's sponge is a serial chain of permutations, but only in its VALUES: the 2,604 block records of a permutation
// are a function of its input state alone.  So the strand kernels run in two phases:
//   values (CoopSinkT<COLS, true>)  the strand's wavefront computes each permutation's output with the state spread over 12
//                                   lanes (no records, no LDS exchange: broadcasts are v_readlane, the one cross-lane sum of a
//                                   partial round is a DPP row scan) and appends {record index, input state} to the proof's
//                                   permutation list;
//   records (k_glp_emit)            one wavefront per LISTED permutation replays it and writes its records - every permutation
//                                   of every strand of the launch side by side (CoopSinkT<COLS, false>::coop_poseidon_permute).
// Together they emit exactly the records the sequential template code (chips.h PoseidonPermutationChip) emits, at the same
// indices — tests/test_gpu_batch.py compares the resulting advice with the oracle byte for byte.
#pragma once
#include <type_traits>
#include "valbackend.h"
#include "bntab.h"
#include "glptab.h"

namespace h2w {

constexpr int GLP_RECS_FULL = 12 + 48 + 1 + 12 * 14;           // constant_layer, sbox_layer, mds_layer
constexpr int GLP_RECS_PARTIAL_ROUND = 4 + 1 + 1 + 11 + 1 + 11; // sbox, +const, d = m00*s0, d chain, zeros, v row
constexpr int GLP_RECS_PARTIAL = 12 + 1 + 121 + N_PARTIAL_ROUNDS * GLP_RECS_PARTIAL_ROUND;
constexpr int GLP_CONST_WORDS = 360 + 12 + 12 + 12 + 22 + 121 + 242 + 242;   // u64 words of the Goldilocks block of h2w_poseidon_consts_t
constexpr int GLP_RECS = 2 * HALF_N_FULL_ROUNDS * GLP_RECS_FULL + GLP_RECS_PARTIAL;   // 2604

// The cooperating block is ONE wavefront: its LDS operations execute in program order and its lanes run in lockstep, so
// lanes exchange the Poseidon state through LDS with no barrier.  __syncthreads() would also wait for vmcnt(0), i.e. for
// every outstanding record STORE to be acknowledged by memory (~1-2 us) at each of the ~90 exchange points of a
// permutation; this waits for LDS only.
__device__ __forceinline__ void wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }

// Goldilocks-Poseidon round constants + MDS, staged in LDS by the cooperative kernels (stage_glp_consts) and read with
// ds_read (a namespace-scope __shared__ array keeps the LDS address space; a pointer member would decay to flat).
__shared__ uint64_t s_glp_k[GLP_CONST_WORDS];
__shared__ uint64_t s_glp_a[SPONGE_WIDTH], s_glp_b[SPONGE_WIDTH];      // the permutation's state exchange buffers
__shared__ uint64_t s_glp_in[CH_BUF];                                  // the challenger's input buffer (values phase of the prologue)
__shared__ uint64_t s_glp_m[SPONGE_WIDTH * SPONGE_WIDTH];              // MDS as a dense matrix, row-major: M[r][j] = circ[(j - r) mod 12] + [j == r] diag[r]  (values phase)
// LDS pointers are handed to the (out-of-line) permutation through the sink: a __shared__ array that several kernels use is reached
// from a non-kernel function through llvm.amdgcn.lds.offset.table - a GLOBAL load of the offset at every use, i.e. a vmcnt(0) wait
// for every record store in flight, inside the round loops (that was 60 of the 89 us of a permutation).  A kernel knows the
// addresses at compile time and passes them down.
__shared__ uint64_t s_glp_x[GLP_AUX_WORDS];                             // the derived tables of the partial rounds (glptab.h; values phase)
typedef __attribute__((address_space(3))) uint64_t lds64_t;
constexpr int KO_ARC = 0, KO_CIRC = 360, KO_DIAG = 372, KO_FIRST = 384, KO_PRC = 396, KO_INIT = 418, KO_WHAT = 539, KO_VS = 781;
static_assert(KO_VS + 242 == GLP_CONST_WORDS, "Goldilocks constant block layout");
// AUX: also the derived tables (the kernels of the values phase; they sit behind the struct: h2w_plan's device copy)
template <bool AUX = false> __device__ __forceinline__ void stage_glp_consts(const h2w_poseidon_consts_t *k, int tid, int nthreads) {
    const uint64_t *src = reinterpret_cast<const uint64_t *>(k);
    for (int i = tid; i < GLP_CONST_WORDS; i += nthreads) s_glp_k[i] = g_load_u64(src + i);
    if constexpr (AUX) { const uint64_t *ax = reinterpret_cast<const uint64_t *>(k + 1); for (int i = tid; i < GLP_AUX_WORDS; i += nthreads) s_glp_x[i] = g_load_u64(ax + i); }
    for (int i = tid; i < SPONGE_WIDTH * SPONGE_WIDTH; i += nthreads) {
        const int r = i / SPONGE_WIDTH, j = i % SPONGE_WIDTH; int d = j - r; if (d < 0) d += SPONGE_WIDTH;
        const uint64_t c = g_load_u64(src + KO_CIRC + d);
        s_glp_m[i] = j == r ? gl_reduce128((u128)c + g_load_u64(src + KO_DIAG + r)) : c;      // (c + diag) v = c v + diag v in the field
    }
    __syncthreads();
}
// plonky2's MDS entries are tiny (<= 41): a row is then two 64-bit sums of 32-bit halves and ONE reduction (the prover's trick, prover.hip).
// Below 2^26 each, a dense entry circ + diag is below 2^27 and a sum of 12 terms plus a constant word below 2^63 (glq_reduce96's precondition, glperm.h).
inline bool glp_small_mds(const h2w_poseidon_consts_t &k) {
    for (int i = 0; i < SPONGE_WIDTH; i++) if (k.mds_circ[i] >= (1ull << 26) || k.mds_diag[i] >= (1ull << 26)) return false;
    return true;
}

__device__ __forceinline__ uint64_t gl_add_c(uint64_t x, uint64_t y) {   // canonical x + y mod p
    uint64_t t = x + y;
    if (t < x || t >= GL_P) t -= GL_P;
    return t;
}
__device__ __forceinline__ uint64_t shfl_up64(uint64_t v, int d) { return __shfl_up(v, d, 64); }

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int src) {      // src: wave-uniform
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
template <int N> __device__ __forceinline__ uint64_t row_shr64(uint64_t v) {      // lane i <- lane i - N of its 16-lane row, 0 where there is none
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, 0x110 + N, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), 0x110 + N, 0xf, 0xf, true);
    return ((uint64_t)hi << 32) | lo;
}
}      // namespace h2w
#include "glperm.h"
namespace h2w {
constexpr int GLP_LIST_WORDS = 1 + SPONGE_WIDTH;            // one listed permutation: {index of its first record, input state}

// VALPH: the values phase of a strand (see the top of the file); false: the record-emitting permutation (k_glp_emit)
template <bool COLS, bool VALPH = false, int HASH_MODE = -1> struct CoopSinkT {
    static constexpr bool kCoop = true, kSplitOnly = false, kBnUnits = false, kDevSponge = VALPH; static constexpr int kHashMode = HASH_MODE;
    rec_t *recs; uint64_t nrec; fr_t *out; uint64_t cell_off; const uint16_t *ncells; int lane; ColPolicy<COLS> cc;
    lds64_t *lk = nullptr, *la = nullptr, *lb = nullptr, *lm = nullptr, *lx = nullptr;      // s_glp_k, s_glp_a, s_glp_b, s_glp_m (s_glp_x: values phase) of the running kernel (set by bind_lds)
    uint64_t *glp = nullptr; uint32_t glp_slot = 0; bool small_mds = false;   // values phase: this proof's permutation list, the next slot of this strand
    __device__ __forceinline__ void bind_lds() { lk = (lds64_t *)s_glp_k; la = (lds64_t *)s_glp_a; lb = (lds64_t *)s_glp_b; lm = (lds64_t *)s_glp_m; if constexpr (VALPH) lx = (lds64_t *)s_glp_x; }
    bool emit = true;            // false: values only (a sharded run computes every prologue for its challenges, but only the owning rank emits it)
    __device__ __forceinline__ void rec(int t, uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
        if (lane == 0 && emit) g_store_rec(recs + nrec, a, b, c, d);
        nrec++; cell_off += ncells[t];
    }
    __device__ __forceinline__ void cell(const fr_t &v) { if (lane == 0 && emit) g_store_fr(out + cc.map(cell_off), v); cell_off++; }
    __device__ __forceinline__ void gate() {}
    __device__ __forceinline__ void lookup() {}
    __device__ __forceinline__ void skip(uint64_t nr, uint64_t nc) { nrec += nr; cell_off += nc; }
    __device__ void merkle_begin(int, int, bool, uint64_t) {}
    __device__ void merkle_end(int, int, bool) {}
    __device__ void query_begin(int, uint64_t) {}
    __device__ void query_end(int, uint64_t) {}
    __device__ void bn_perm_begin(bool) {}
    __device__ void bn_perm_end(bool) {}
    __device__ void glp_note() {}
    __device__ void note_load(uint64_t, int) {}
    __device__ bool bn_emit_inline(fr_t *, const ValCfg &, bool &) { return false; }
    __device__ void note_cap_hash(uint64_t) {}
    // WitnessChip::load_proof_with_pis (witness/mod.rs:267-294) and the limb decompositions of the caps' BN254 hashes are k_prologue_load's
    // (every item is independent: one lane each); the strand only steps over their records and cells
    __device__ __forceinline__ bool coop_load_proof(const ValCfg &cfg) { nrec += cfg.load_nrec; cell_off += cfg.load_ncell; return true; }
    // ---- the Fiat-Shamir sponge of the values phase (ChallengerChip, challenger/mod.rs:19-126; overwrite-mode duplex, rate 8): the state
    // lives on the lanes (lane l: element l), the input buffer in LDS; every permutation is listed for k_glp_emit
    uint64_t sx = 0; int sp_in = 0, sp_out = 0; lds64_t *lin = nullptr;
#ifdef H2W_EXP_GLP_CLOCK
    long long dbg_cycles = 0, dbg_t[24], dbg_c[24]; int dbg_n = 0, dbg_k = 0, dbg_m[24];
#endif
    __device__ __forceinline__ void sponge_init() { sx = 0; sp_in = sp_out = 0; lin = (lds64_t *)s_glp_in; }
    __device__ __forceinline__ bool sponge_observe(uint64_t t) { sp_out = 0; if (sp_in >= CH_BUF) return false; lin[sp_in++] = t; return true; }
    template <class WordFn> __device__ __forceinline__ bool sponge_observe_words(const uint64_t *proof, int n, WordFn word) {      // n proof words, one lane each
        sp_out = 0; if (sp_in + n > CH_BUF) return false;
        for (int base = 0; base < n; base += 64) { const int i = base + lane; if (i < n) lin[sp_in + i] = g_load_u64(proof + word(i)); }
        sp_in += n; return true;
    }
    // proof words staged side by side in the (idle) input buffer, one lane a word: what the strand then reads one by one costs an LDS read each instead
    // of a dependent global load (~2.4 k cycles for a wavefront with nothing else to run)
    template <class WordFn> __device__ __forceinline__ void stage_words(const uint64_t *proof, int at, int n, WordFn word) {
        for (int base = 0; base < n; base += 64) { const int i = base + lane; if (i < n) lin[at + i] = g_load_u64(proof + word(i)); }
    }
    __device__ __forceinline__ uint64_t staged_word(int i) const { return lin[i]; }
    // observe_cap (challenger/mod.rs:65-74): hash j of the cap on lane j; Goldilocks hashes are their 4 words, BN254 hashes 5 limbs of 56 bits
    // (HashWire::to_goldilocks_vec, hash/poseidon_bn254/hash.rs:31-43: decompose_le(56, 5) - its cells are k_prologue_load's)
    __device__ __forceinline__ bool sponge_observe_cap(const uint64_t *proof, uint64_t w0, int n, int mode, int L) {      // (inlined, like sponge_challenge: an out-of-line member takes the sink's address, and a sink in memory pays a scratch round trip behind the record stores for every counter it bumps)
        const int per = mode == 0 ? 4 : 5;
        sp_out = 0; if (sp_in + per * n > CH_BUF) return false;
        for (int base = 0; base < n; base += 64) {
            const int j = base + lane;
            if (j < n) {
                fr_t x; for (int i = 0; i < 4; i++) x.l[i] = g_load_u64(proof + w0 + 4ull * j + i);
                if (mode == 0) for (int i = 0; i < 4; i++) lin[sp_in + 4 * j + i] = x.l[i];
                else for (int t = 0; t < 5; t++) lin[sp_in + 5 * j + t] = fr_bits(x, 56 * t, 56);
            }
        }
        sp_in += per * n;
        if (mode == 1) { const int nl = (56 + L - 1) / L, rem = 56 % L; cell_off += (uint64_t)n * (13 + 5ull * ((nl > 1 ? 1 + 3 * (nl - 1) : 0) + (rem ? 4 : 0))); }
        return true;
    }
    __device__ __forceinline__ void sponge_permute() {
        // list the permutation (lane 0: where its records start; lanes 1..12: the input state), then its values
        // (the list entry is STORED by the permutation, behind its entry: an out-of-line function waits at its entry for every memory operation in
        // flight, and a store issued just before the call was 2.5 k cycles of every permutation)
        const uint64_t up = row_shr64<1>(sx), w = lane == 0 ? nrec : up;
        uint64_t *at = emit && lane < GLP_LIST_WORDS ? glp + (uint64_t)glp_slot * GLP_LIST_WORDS + lane : nullptr;
        glp_slot++;
#ifdef H2W_EXP_GLP_CLOCK      // experiment: the cycles of the strand's sponge permutations (printed by k_prologue_values)
        const long long c0 = clock64();
#endif
        sx = glp_permute_lanes(sx, lk, lm, lx, lane, small_mds, at, w);
#ifdef H2W_EXP_GLP_CLOCK
        dbg_cycles += clock64() - c0; dbg_n++;
#endif
        nrec += GLP_RECS;
        cell_off += perm_cell_count();
    }
    // cells of one permutation's records (a constant of the run: read from the template table once - a global load behind the list store above
    // waited, every permutation, for that store to drain)
    uint64_t perm_cells_ = 0;
    __device__ __forceinline__ uint64_t perm_cell_count() {
        if (perm_cells_ == 0) {
            const uint64_t nG = ncells[T_GLOP], nKA = ncells[T_KA_GLOP];
            perm_cells_ = 2 * HALF_N_FULL_ROUNDS * (12 * nKA + 48 * nG + 12 + 12 * (1 + 13 * nKA)) + 12 * nKA + 12 + 121 * nKA + (uint64_t)N_PARTIAL_ROUNDS * (4 * nG + nKA + nKA + 11 * nKA + 12 + 11 * nKA);
        }
        return perm_cells_;
    }
    __device__ __forceinline__ uint64_t sponge_challenge() {                   // ChallengerChip::get_challenge (:92-108, :260-277)
        if (sp_in) {
            for (int off = 0; off < sp_in; off += SPONGE_RATE) {
                const int len = sp_in - off < SPONGE_RATE ? sp_in - off : SPONGE_RATE;
                if ((lane & 15) < len) sx = lin[off + (lane & 15)];      // (every 16-lane row alike: glperm.h)
                sponge_permute();
            }
            sp_in = 0; sp_out = SPONGE_RATE;
        } else if (sp_out == 0) { sponge_permute(); sp_out = SPONGE_RATE; }
        return readlane64(sx, --sp_out);
    }

#if defined(H2W_FLATTEN_CHIPS)      // (glue.hip: the values strands, flattened - the state array stays in registers)
    __device__ __forceinline__
#else
    __device__ __noinline__
#endif
    void coop_poseidon_permute(uint64_t *st, const h2w_poseidon_consts_t *) {
        if constexpr (VALPH) {
            // values phase: list the permutation (lane 0: where its records start; lanes 1..12: the input state), compute its output
            uint64_t w = nrec, x = 0;
#pragma unroll
            for (int i = 0; i < SPONGE_WIDTH; i++) { if (lane == i + 1) w = st[i]; if ((lane & 15) == i) x = st[i]; }      // (every 16-lane row alike: glperm.h)
            uint64_t *at = emit && lane < GLP_LIST_WORDS ? glp + (uint64_t)glp_slot * GLP_LIST_WORDS + lane : nullptr;
            glp_slot++;
            x = glp_permute_lanes(x, lk, lm, lx, lane, small_mds, at, w);
#pragma unroll
            for (int i = 0; i < SPONGE_WIDTH; i++) st[i] = readlane64(x, i);
            nrec += GLP_RECS;
            cell_off += perm_cell_count();
            return;
        }
        lds64_t *const K = lk, *const s_a = la, *const s_b = lb;      // (by value: one read of the sink object per call)
        rec_t *R = recs + nrec;
        const int l = lane, grp13 = lane / 13, idx13 = lane % 13;
        const bool em = emit;
        auto W = [&](int idx, uint64_t A, uint64_t B, uint64_t C) { if (em) g_store_rec(R + idx, A, B, C, 0); };
        if (l < SPONGE_WIDTH) s_a[l] = st[l];
        wave_sync();
        int base = 0, round_ctr = 0;
        auto full_round = [&]() {
            if (l < SPONGE_WIDTH) {
                uint64_t x = s_a[l]; const uint64_t rc = K[KO_ARC + l + SPONGE_WIDTH * round_ctr];
                W(base + l, rc, 1, x); x = gl_add(x, rc);                                   // constant_layer
                const int sb = base + 12 + 4 * l;                                            // sbox_monomial: x^7
                const uint64_t x2 = gl_mul(x, x); W(sb, x, x, 0);
                const uint64_t x4 = gl_mul(x2, x2); W(sb + 1, x2, x2, 0);
                const uint64_t x6 = gl_mul(x4, x2); W(sb + 2, x4, x2, 0);
                const uint64_t x7 = gl_mul(x6, x); W(sb + 3, x6, x, 0);
                s_b[l] = x7;
            }
            wave_sync();
            const int mb = base + 60;                                                        // mds_layer
            // 12 rows x 13 terms: res_i = (sum_{j<=i} c_j v_j) mod p.  Products in parallel (lane = row_in_pass*13 + i, 4 rows
            // per pass), then a segmented inclusive scan over each 13-lane group: 1 product + 4 adds deep instead of 13 mul-adds.
            if (l == 0) W(mb, 0, 0, 0);
            uint64_t rowres = 0;
#pragma unroll
            for (int pass = 0; pass < 3; pass++) {
                const int rr = pass * 4 + grp13;
                uint64_t c = 0, v = 0, x = 0;
                if (l < 52) {
                    if (idx13 < 12) { c = K[KO_CIRC + idx13]; int vi = idx13 + rr; if (vi >= 12) vi -= 12; v = s_b[vi]; }
                    else { c = K[KO_DIAG + rr]; v = s_b[rr]; }
                    x = gl_mul(c, v);
                }
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) { const uint64_t y = shfl_up64(x, d); if (idx13 >= d) x = gl_add_c(x, y); }
                uint64_t prev = shfl_up64(x, 1); if (idx13 == 0) prev = 0;
                if (l < 52) {
                    const int rb = mb + 1 + 14 * rr;
                    if (idx13 == 0) W(rb, 0, 0, 0);
                    W(rb + 1 + idx13, c, v, prev);
                    if (idx13 == 12) rowres = x;
                }
                if (l < 52 && idx13 == 12) s_a[rr] = rowres;
            }
            wave_sync();
            base += GLP_RECS_FULL; round_ctr++;
        };
        for (int i = 0; i < HALF_N_FULL_ROUNDS; i++) full_round();
        // ---- partial rounds (hash/poseidon/permutation.rs:216-239)
        if (l < SPONGE_WIDTH) {                                                              // partial_first_constant_layer
            const uint64_t x = s_a[l], c = K[KO_FIRST + l];
            W(base + l, c, 1, x); s_b[l] = gl_add(x, c);
        }
        wave_sync();
        base += 12;
        if (l == 0) { W(base, 0, 0, 0); s_a[0] = s_b[0]; }                                   // mds_partial_layer_init
        else if (l < SPONGE_WIDTH) {
            uint64_t res = 0;
            for (int r = 1; r < SPONGE_WIDTH; r++) {
                const uint64_t t = K[KO_INIT + (r - 1) * 11 + (l - 1)], v = s_b[r];
                W(base + 1 + (r - 1) * 11 + (l - 1), t, v, res); res = gl_muladd(t, v, res);
            }
            s_a[l] = res;
        }
        wave_sync();
        base += 122;
        for (int r = 0; r < N_PARTIAL_ROUNDS; r++) {
            if (l == 0) {
                const uint64_t x = s_a[0];
                const uint64_t x2 = gl_mul(x, x); W(base, x, x, 0);
                const uint64_t x4 = gl_mul(x2, x2); W(base + 1, x2, x2, 0);
                const uint64_t x6 = gl_mul(x4, x2); W(base + 2, x4, x2, 0);
                const uint64_t x7 = gl_mul(x6, x); W(base + 3, x6, x, 0);
                const uint64_t c = K[KO_PRC + r];
                W(base + 4, c, 1, x7); s_a[0] = gl_add(x7, c);
            }
            wave_sync();
            const uint64_t s0 = s_a[0];                                                      // mds_partial_layer_fast
            {   // d = m00*s0 + sum_i w_hat_i * st_i as an inclusive scan over lanes 0..11 (records need every partial sum)
                uint64_t t = 0, v = 0, x = 0;
                if (l == 0) { t = K[KO_CIRC] + K[KO_DIAG]; v = s0; }
                else if (l < SPONGE_WIDTH) { t = K[KO_WHAT + r * 11 + (l - 1)]; v = s_a[l]; }
                if (l < SPONGE_WIDTH) x = gl_mul(t, v);
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) { const uint64_t y = shfl_up64(x, d); if (l >= d) x = gl_add_c(x, y); }
                uint64_t prev = shfl_up64(x, 1); if (l == 0) prev = 0;
                if (l < SPONGE_WIDTH) W(base + 5 + l, t, v, prev);
                if (l == 0) W(base + 17, 0, 0, 0);
                if (l == SPONGE_WIDTH - 1) s_b[0] = x;
                if (l > 0 && l < SPONGE_WIDTH) {
                    const uint64_t tv = K[KO_VS + r * 11 + (l - 1)];
                    W(base + 17 + l, tv, s0, v); s_b[l] = gl_muladd(tv, s0, v);
                }
            }
            wave_sync();
            if (l < SPONGE_WIDTH) s_a[l] = s_b[l];
            wave_sync();
            base += GLP_RECS_PARTIAL_ROUND;
        }
        round_ctr += N_PARTIAL_ROUNDS;
        for (int i = 0; i < HALF_N_FULL_ROUNDS; i++) full_round();
        for (int i = 0; i < SPONGE_WIDTH; i++) st[i] = s_a[i];
        wave_sync();
        const uint64_t nG = ncells[T_GLOP], nKA = ncells[T_KA_GLOP];
        const uint64_t full_cells = 12 * nKA + 48 * nG + 12 + 12 * (1 + 13 * nKA);
        const uint64_t part_cells = 12 * nKA + 12 + 121 * nKA + (uint64_t)N_PARTIAL_ROUNDS * (4 * nG + nKA + nKA + 11 * nKA + 12 + 11 * nKA);
        nrec += GLP_RECS; cell_off += 2 * HALF_N_FULL_ROUNDS * full_cells + part_cells;
    }
};

// ---------------------------------------------------------------------------------------------------------------
// QuadSink: four adjacent lanes execute a PoseidonBN254 Merkle chain strand quad-uniformly (lane 0 of the quad writes the
// strand's few direct cells) and split the width-4 state between them: lane i owns state element i.
//
// A Merkle path is a serial chain of permutations (merkle/mod.rs:57-78) only in its VALUES; the 4,032 cells of a permutation
// (hash/poseidon_bn254/permutation.rs:48-203) are a function of its input state.  Two phases, like the Goldilocks sponge above:
//   QUAD_VALUES  k_merkle_bn_values: one quad per strand walks the chain on values alone (Montgomery-form state, x^5 in three
//                products: 282 dependent wavefront-level products per permutation instead of the emitter's 352, no staging, no
//                stores but the 128-byte output state of each permutation unit);
//   QUAD_EMIT    k_merkle_bn_emit: one quad per permutation UNIT of every strand - all levels of all paths side by side.  The
//                quad walks its strand's (cheap) control flow, takes the state its unit starts from out of the values phase's
//                buffer, and emits the cells of its window: the direct cells in front of its unit (selects, limb sums) and the
//                permutation's 4,032 cells, computed and streamed by the quad itself as described below.
//
// Memory behaviour (what the round-1 emitter stalled on, profiles/r01_pmc_issue_*):
//   * the PoseidonBN254 tables (canonical for the cells and the adds, pre-multiplied by R for the products) are a per-PLAN device
//     buffer staged into LDS once per block: lane-indexed constant reads are ds_reads (lgkmcnt) and never queue behind the
//     kernel's own cell stores (loads and stores share vmcnt in order on this family);
//   * every VALUE of a layer is written to LDS once (a "value slot", 32 B per quad and slot), not once per cell that shows it:
//     the cells of a layer are a static list of sources (value slot | table entry), and the quad streams them out with all
//     four lanes writing adjacent 16-byte pieces (64 contiguous bytes per quad and store instruction: the form that streams
//     at the HBM ceiling, tools/ubench/ubench_store3.hip).  A partial round stages 7 values instead of 52 cells;
//   * quad-local exchanges are DPP moves (quad_perm), not ds_bpermute: no LDS round trip on the dependent chain;
//   * one wavefront's LDS accesses execute in order, so no barrier or wait separates staging, flushing and re-staging.
// Arithmetic is "hybrid": canonical state, const * var = ONE Montgomery product with the R-premultiplied constant, x^5 in five
// products (X = x R; x2 = x X / R; X2 = X X / R; x4 = x2 X2 / R; x5 = x4 X / R); in the partial rounds the lanes without an
// S-box do their own products inside the S-box's instruction stream (5 wavefront-level products per round).
#ifndef H2W_QUAD_BLOCK
#define H2W_QUAD_BLOCK 256
#endif
constexpr int QUAD_BLOCK = H2W_QUAD_BLOCK;                   // threads per block of k_merkle_bn_fused: its wavefronts share one LDS copy of the tables
constexpr int QUAD_WAVES = QUAD_BLOCK / 64;
struct __attribute__((aligned(16))) sq16_t { unsigned long long x, y; };
__shared__ sq16_t s_bn_tab[BK_X * 2];                        // [form][entry][half], then the BK_XC block: 34.7 KB
__shared__ sq16_t s_bn_val[QUAD_WAVES * BN_NSLOT * BN_SLOT_SQ];   // [wavefront][slot][quad][half]: 10 KB per wavefront
__device__ __forceinline__ void stage_bn_consts(const fr_t *tab, int tid, int nthreads) {
    const sq16_t *src = reinterpret_cast<const sq16_t *>(tab);
    for (int i = tid; i < BK_X * 2; i += nthreads) { sq16_t v; v.x = H2W_GLOAD64(&src[i].x); v.y = H2W_GLOAD64(&src[i].y); s_bn_tab[i] = v; }
    __syncthreads();
}
__device__ __forceinline__ fr_t bnk(int which, int idx) {
    const sq16_t a = s_bn_tab[(which * BK_T + idx) * 2], b = s_bn_tab[(which * BK_T + idx) * 2 + 1];
    fr_t r; r.l[0] = a.x; r.l[1] = a.y; r.l[2] = b.x; r.l[3] = b.y; return r;
}
__device__ __forceinline__ fr_t bnkc(int r) {
    const sq16_t a = s_bn_tab[(BK_XC + r) * 2], b = s_bn_tab[(BK_XC + r) * 2 + 1];
    fr_t v; v.l[0] = a.x; v.l[1] = a.y; v.l[2] = b.x; v.l[3] = b.y; return v;
}
__shared__ uint32_t s_bn_tab9[BK9_N * BK9_W];
__device__ __forceinline__ void stage_bn_consts9(const uint32_t *tab9, int tid, int nthreads) {
    const sq16_t *src = reinterpret_cast<const sq16_t *>(tab9); sq16_t *dst = reinterpret_cast<sq16_t *>(s_bn_tab9);
    for (int i = tid; i < BK9_N * BK9_W / 4; i += nthreads) { sq16_t v; v.x = H2W_GLOAD64(&src[i].x); v.y = H2W_GLOAD64(&src[i].y); dst[i] = v; }
    __syncthreads();
}
__device__ __forceinline__ fr9_t bnk9(int idx) {
    const uint4 a = *reinterpret_cast<const uint4 *>(s_bn_tab9 + idx * BK9_W), b = *reinterpret_cast<const uint4 *>(s_bn_tab9 + idx * BK9_W + 4);
    fr9_t v; v.t[0] = a.x; v.t[1] = a.y; v.t[2] = a.z; v.t[3] = a.w; v.t[4] = b.x; v.t[5] = b.y; v.t[6] = b.z; v.t[7] = b.w; v.t[8] = s_bn_tab9[idx * BK9_W + 8];
    return v;
}
// quad-local exchanges as DPP moves (quad_perm control: two bits per destination lane)
template <int CTRL> __device__ __forceinline__ fr_t quad_dpp(const fr_t &v) {
    fr_t r;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const unsigned lo = (unsigned)v.l[i], hi = (unsigned)(v.l[i] >> 32);
        const unsigned rlo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, CTRL, 0xf, 0xf, false);
        const unsigned rhi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, CTRL, 0xf, 0xf, false);
        r.l[i] = ((unsigned long long)rhi << 32) | rlo;
    }
    return r;
}
template <int J> __device__ __forceinline__ fr_t quad_bcast(const fr_t &v) { return quad_dpp<J * 0x55>(v); }      // every lane <- lane J of its quad
__device__ __forceinline__ fr_t quad_up1(const fr_t &v) { return quad_dpp<0x90>(v); }                            // lane i <- lane i - 1 (lane 0 keeps its own)
__device__ __forceinline__ fr_t quad_up2(const fr_t &v) { return quad_dpp<0x44>(v); }                            // lanes 2, 3 <- lanes 0, 1

template <int CTRL> __device__ __forceinline__ fr9_t quad_dpp9(const fr9_t &v) {
    fr9_t r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.t[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.t[i], CTRL, 0xf, 0xf, false);
    return r;
}
template <int J> __device__ __forceinline__ fr9_t quad_bcast9(const fr9_t &v) { return quad_dpp9<J * 0x55>(v); }
__device__ __forceinline__ fr9_t fr9_sel(bool c, const fr9_t &a, const fr9_t &b) {
    fr9_t r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.t[i] = c ? a.t[i] : b.t[i];
    return r;
}
__device__ __forceinline__ fr9_t fr9_add_if(bool c, const fr9_t &a, const fr9_t &b) {      // a + (c ? b : 0)
    fr9_t r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.t[i] = a.t[i] + (c ? b.t[i] : 0u);
    return r;
}
// limb-wise select: `c ? a : b` on the struct itself can be lowered to a select of two stack addresses plus a scratch load, and a scratch
// load waits (vmcnt is in order on this family) for every cell store issued before it
__device__ __forceinline__ fr_t fr_sel(bool c, const fr_t &a, const fr_t &b) {
    fr_t r;
#pragma unroll
    for (int i = 0; i < 4; i++) r.l[i] = c ? a.l[i] : b.l[i];
    return r;
}
// one cell source of a layer: a value slot of the quad, a table entry, or a table entry relative to one of two per-layer bases
struct BnSrc { int kind, idx; };                              // kind 0: value slot; 1: table entry; 2: entry base A + idx; 3: entry base B + idx
constexpr BnSrc bV(int s) { return BnSrc{0, s}; }
constexpr BnSrc bK(int i) { return BnSrc{1, i}; }
constexpr BnSrc bA(int o) { return BnSrc{2, o}; }
constexpr BnSrc bB(int o) { return BnSrc{3, o}; }
// x^5 of all four state elements, lane l's 12 cells at 12 l: [0, s, s, x2, 0, x2, x2, x4, 0, x4, s, x5]; slots s: l, x2: 4+l, x4: 8+l, x5: 12+l
constexpr BnSrc bn_map_exp5(int c) {
    const int l = c / 12, i = c % 12;
    return (i == 0 || i == 4 || i == 8) ? bK(BK_ZERO) : (i == 1 || i == 2 || i == 10) ? bV(l) : (i == 3 || i == 5 || i == 6) ? bV(4 + l) : (i == 7 || i == 9) ? bV(8 + l) : bV(12 + l);
}
// ark: lane l's 5 cells at 5 l: [c][s, c, 1, s + c]; base A = first round constant; slots s: l, s + c: 4 + l
constexpr BnSrc bn_map_ark(int c) {
    const int l = c / 5, i = c % 5;
    return (i == 0 || i == 2) ? bA(l) : i == 1 ? bV(l) : i == 3 ? bK(BK_ONE) : bV(4 + l);
}
// mix: lane l's row at 16 l: for j: [acc_j, m[j][l], s_j, acc_{j+1}]; base A = the matrix; slots s_j: j, acc_{j+1} of row l: 4 + 4 l + j
constexpr BnSrc bn_map_mix(int c) {
    const int l = c / 16, j = (c % 16) / 4, i = c % 4;
    return i == 0 ? (j == 0 ? bK(BK_ZERO) : bV(4 + 4 * l + j - 1)) : i == 1 ? bA(4 * j + l) : i == 2 ? bV(j) : bV(4 + 4 * l + j);
}
constexpr BnSrc bn_map_consts(int c) { return bA(c); }       // load_constant of M then P (contiguous in the table)
constexpr BnSrc bn_map_zero(int) { return bK(BK_ZERO); }
// partial round (hash/poseidon_bn254/permutation.rs:83-110): base A = the round's 7 sparse-matrix entries, base B = its round constant.
// slots: 0 s0, 1-3 s_k, 4 x2, 5 x4, 6 x5, 7 s0' = x5 + c, 8-11 running sums of the sparse row, 12-14 updated s_k
constexpr BnSrc bn_map_partial(int c) {
    if (c < 12) return (c == 0 || c == 4 || c == 8) ? bK(BK_ZERO) : (c == 1 || c == 2 || c == 10) ? bV(0) : (c == 3 || c == 5 || c == 6) ? bV(4) : (c == 7 || c == 9) ? bV(5) : bV(6);
    if (c < 17) return (c == 12 || c == 14) ? bB(0) : c == 13 ? bV(6) : c == 15 ? bK(BK_ONE) : bV(7);
    if (c < 37) { const int j = (c - 17) / 5, i = (c - 17) % 5; return (i == 0 || i == 2) ? bA(j) : i == 1 ? (j == 0 ? bK(BK_ZERO) : bV(8 + j - 1)) : i == 3 ? (j == 0 ? bV(7) : bV(j)) : bV(8 + j); }
    const int k = 1 + (c - 37) / 5, i = (c - 37) % 5;
    return (i == 0 || i == 2) ? bA(4 + k - 1) : i == 1 ? bV(k) : i == 3 ? bV(7) : bV(12 + k - 1);
}

enum { QUAD_VALUES = 1, QUAD_EMIT = 2, QUAD_FUSED = 3 };      // QUAD_FUSED: one quad walks its strand AND emits every unit of it (one pass, serial in the path's depth)
template <bool COLS, int MODE> struct QuadSinkT {
    static constexpr bool kCoop = false, kSplitOnly = false, kBnUnits = true, kDevSponge = false; static constexpr int kHashMode = 1;
    rec_t *recs; uint64_t nrec; fr_t *out; uint64_t cell_off; const uint16_t *ncells; int l4; ColPolicy<COLS> cc;
    fr_t *ustate;                  // output states of this strand's permutation units, [unit][4] (written by QUAD_VALUES, read by QUAD_EMIT)
    fr_t *sbx;                     // the S-box chains of their partial rounds, [unit][BN_PARTIAL_ROUNDS][3] = canonical x^2, x^4, x^5 (QUAD_VALUES -> QUAD_EMIT)
    int unit_local = 0;            // units of the strand passed so far
    int my_w = 0, last_w = 0;      // QUAD_EMIT: the unit this quad emits; the strand's last unit (its window also holds the cells behind it)
    bool act = true;               // QUAD_EMIT: the cursor is inside this quad's window
    __device__ __forceinline__ void set_window(int w, int n_units) { my_w = w; last_w = n_units > 0 ? n_units - 1 : 0; unit_local = 0; act = w == 0; }
    __device__ __forceinline__ void rec(int t, uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
        if (MODE != QUAD_VALUES && l4 == 0 && act) g_store_rec(recs + nrec, a, b, c, d);
        nrec++; cell_off += ncells[t];
    }
    __device__ __forceinline__ void cell(const fr_t &v) { if (MODE != QUAD_VALUES && l4 == 0 && act) g_store_fr(out + cc.map(cell_off), v); cell_off++; }
    __device__ __forceinline__ void gate() {}
    __device__ __forceinline__ void lookup() {}
    __device__ void note_cap_hash(uint64_t) {}
    __device__ __forceinline__ void skip(uint64_t nr, uint64_t nc) { nrec += nr; cell_off += nc; }
    __device__ void merkle_begin(int, int, bool, uint64_t) {}
    __device__ void merkle_end(int, int, bool) {}
    __device__ void query_begin(int, uint64_t) {}
    __device__ void query_end(int, uint64_t) {}
    __device__ void bn_perm_begin(bool) {}
    __device__ void bn_perm_end(bool) {}
    __device__ void glp_note() {}
    __device__ void note_load(uint64_t, int) {}
    __device__ bool coop_load_proof(const ValCfg &) { return false; }
    __device__ void coop_poseidon_permute(uint64_t *, const h2w_poseidon_consts_t *) {}
    // The emitter's working set, by value (a member read behind an opaque call is a FLAT load of the sink object, and flat loads
    // wait for every cell store in flight).
    struct Em {
        sq16_t *val;                   // this quad's value slots (slot s at val + s * BN_SLOT_SQ)
        const sq16_t *tabh;            // canonical table + (l & 1): the half this lane flushes
        unsigned long long gdst;       // byte address of this lane's first 16-byte piece of the current layer
        uint64_t cell0;                // flat index of the layer's first cell
        int l;
        ColPolicy<COLS> cc;            // column layout: the cursor, by value too (nothing in the flat layout)
    };
    static __device__ __forceinline__ void put(const Em &e, int slot, const fr_t &v) {
        sq16_t *q = e.val + slot * BN_SLOT_SQ; q[0] = sq16_t{v.l[0], v.l[1]}; q[1] = sq16_t{v.l[2], v.l[3]};
    }
    static __device__ __forceinline__ void put_if(const Em &e, bool on, int slot, const fr_t &v) { if (on) put(e, slot, v); }
    template <class MapFn> static __device__ __forceinline__ const sq16_t *src_ptr(const Em &e, MapFn map, int c, const sq16_t *baseA, const sq16_t *baseB) {
        const BnSrc s = map(c);
        return s.kind == 0 ? e.val + s.idx * BN_SLOT_SQ + (e.l & 1) : s.kind == 1 ? e.tabh + s.idx * 2 : s.kind == 2 ? baseA + s.idx * 2 : baseB + s.idx * 2;
    }
    // Streams the N cells of a layer: piece p = 4 k + lane is the (p & 1) half of cell p >> 1, so lanes 0, 1 write cell 2 k and lanes
    // 2, 3 cell 2 k + 1, 64 contiguous bytes per quad.  baseA / baseB: the layer's table bases (already + the lane's half).
    template <int N, class MapFn> __device__ __forceinline__ void flush_layer(Em &e, MapFn map, const sq16_t *baseA, const sq16_t *baseB) {
        constexpr int NP = (N + 1) / 2;
        const bool hiq = e.l >= 2;
        if constexpr (COLS) {
            if (e.cc.hi - e.cell0 < (uint64_t)N || e.cell0 < e.cc.lo) { e = flush_layer_cols<N>(e, map, baseA, baseB); return; }
        }
#pragma unroll
        for (int k0 = 0; k0 < NP; k0 += 4) {                         // four LDS reads in flight, then their stores
            sq16_t t[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int k = k0 + u;
                if (k < NP) {
                    const sq16_t *pa = src_ptr(e, map, 2 * k, baseA, baseB), *pb = (2 * k + 1 < N) ? src_ptr(e, map, 2 * k + 1, baseA, baseB) : pa;
                    t[u] = *(hiq ? pb : pa);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int k = k0 + u;
                if (k < NP) {
                    unsigned long long *g = reinterpret_cast<unsigned long long *>(e.gdst + (unsigned long long)k * 64);
                    if (2 * k + 1 < N || !hiq) { H2W_GSTORE64(g, t[u].x); H2W_GSTORE64(g + 1, t[u].y); }
                }
            }
        }
        e.cell0 += (uint64_t)N; e.gdst += (unsigned long long)N * 32;
    }
    // The same stream in two halves around other work: part_load issues the LDS reads of pieces [K0, K1) of a layer, part_store their
    // stores.  The partial rounds flush round r - 1 this way between the products of round r (four parts), so that no read is waited for
    // and the flush costs its issue slots only (the one wavefront of a SIMD has nobody to hide an s_waitcnt behind).
    template <int K0, int K1, int N, class MapFn> static __device__ __forceinline__ void part_load(const Em &e, MapFn map, const sq16_t *baseA, const sq16_t *baseB, sq16_t *t) {
        const bool hiq = e.l >= 2;
#pragma unroll
        for (int k = K0; k < K1; k++) {
            const sq16_t *pa = src_ptr(e, map, 2 * k, baseA, baseB), *pb = (2 * k + 1 < N) ? src_ptr(e, map, 2 * k + 1, baseA, baseB) : pa;
            t[k - K0] = *(hiq ? pb : pa);
        }
    }
    template <int K0, int K1, int N> static __device__ __forceinline__ void part_store(const Em &e, const sq16_t *t) {
        const bool hiq = e.l >= 2;
#pragma unroll
        for (int k = K0; k < K1; k++) {
            unsigned long long *g = reinterpret_cast<unsigned long long *>(e.gdst + (unsigned long long)k * 64);
            if (2 * k + 1 < N || !hiq) { H2W_GSTORE64(g, t[k - K0].x); H2W_GSTORE64(g + 1, t[k - K0].y); }
        }
    }
    // The partial-round layer as a WAVEFRONT-wide stream: the quad-wise flush above writes 64 contiguous bytes per quad and store instruction,
    // i.e. sixteen 64-byte segments in sixteen different permutations per wavefront instruction - with every level of every path in flight that is
    // ~32 k interleaved write streams and the kernel sat at 3.7 TB/s with its arithmetic at 45 % (profiles/r03_emit_store_bound.txt).  Here the 64 lanes
    // write 1 KB of ONE quad's layer per instruction (the form that streams at the ceiling, tools/ubench/ubench_store*.hip), quad after quad: lane j owns the
    // 16-byte pieces j and 64 + j of every quad's 104-piece layer; where a piece comes from (value slot of that quad | table entry) is a per-lane
    // constant, worked out once per permutation.
    typedef __attribute__((address_space(3))) const sq16_t lds_sq16;
    struct Wide { int off0, off1, kind0, kind1; bool has1; int lane; unsigned long long gq; };      // gq: this lane's quad's layer base (the same on the quad's four lanes)
    static __device__ __forceinline__ void wide_desc(int pc, int &off, int &kind) {
        const int c = pc >> 1; const BnSrc sd = bn_map_partial(c < 52 ? c : 51);
        kind = sd.kind; off = (sd.kind == 0 ? sd.idx * BN_SLOT_SQ : sd.idx * 2) + (pc & 1);
    }
    static __device__ __forceinline__ Wide wide_init(const Em &e) {
        Wide w; w.lane = threadIdx.x & 63; w.has1 = w.lane < 2 * 52 - 64;
        wide_desc(w.lane, w.off0, w.kind0); wide_desc(w.has1 ? 64 + w.lane : 0, w.off1, w.kind1);
        w.gq = e.gdst - (unsigned long long)e.l * 16;
        return w;
    }
    // quads [Q0, Q1) of the wavefront: the LDS reads ...
    template <int Q0, int Q1> static __device__ __forceinline__ void wide_load(const Wide &w, lds_sq16 *valw, lds_sq16 *tab, lds_sq16 *baseA, lds_sq16 *baseB, sq16_t *t0, sq16_t *t1) {
        lds_sq16 *b0 = (w.kind0 == 0 ? valw : w.kind0 == 1 ? tab : w.kind0 == 2 ? baseA : baseB) + w.off0;
        lds_sq16 *b1 = (w.kind1 == 0 ? valw : w.kind1 == 1 ? tab : w.kind1 == 2 ? baseA : baseB) + w.off1;
        const int q0 = w.kind0 == 0 ? 2 : 0, q1 = w.kind1 == 0 ? 2 : 0;      // a value slot holds the sixteen quads' values 32 bytes apart
#pragma unroll
        for (int q = Q0; q < Q1; q++) { t0[q - Q0].x = b0[q * q0].x; t0[q - Q0].y = b0[q * q0].y; t1[q - Q0].x = b1[q * q1].x; t1[q - Q0].y = b1[q * q1].y; }
    }
    // ... and their stores: 1 KB + 640 B contiguous per quad
    template <int Q0, int Q1> static __device__ __forceinline__ void wide_store(const Wide &w, const sq16_t *t0, const sq16_t *t1) {
#pragma unroll
        for (int q = Q0; q < Q1; q++) {
            const unsigned long long gb = readlane64(w.gq, 4 * q) + (unsigned long long)w.lane * 16;
            unsigned long long *g = reinterpret_cast<unsigned long long *>(gb);
            H2W_GSTORE64(g, t0[q - Q0].x); H2W_GSTORE64(g + 1, t0[q - Q0].y);
            if (w.has1) { H2W_GSTORE64(g + 128, t1[q - Q0].x); H2W_GSTORE64(g + 129, t1[q - Q0].y); }
        }
    }
    // column-major layout, a layer that crosses into the next column (rare): per-cell addresses.  The working set goes in and comes back BY VALUE:
    // taken by reference, this one out-of-line call kept the emitter's whole working set on the stack in the column-layout kernels
    template <int N, class MapFn> __device__ __noinline__ Em flush_layer_cols(Em e, MapFn map, const sq16_t *baseA, const sq16_t *baseB) {
        if constexpr (COLS) {
            const bool hiq = e.l >= 2;
            for (int k = 0; k < (N + 1) / 2; k++) {
                const int c = 2 * k + (hiq ? 1 : 0);
                if (c < N) {
                    // (the map is a compile-time function: evaluate it through a small switch-free loop over both candidates)
                    const sq16_t *pa = src_ptr_rt(e, map, 2 * k, baseA, baseB), *pb = (2 * k + 1 < N) ? src_ptr_rt(e, map, 2 * k + 1, baseA, baseB) : pa;
                    const sq16_t t = *(hiq ? pb : pa);
                    const uint64_t a = e.cc.map(e.cell0 + (uint64_t)c);
                    unsigned long long *g = reinterpret_cast<unsigned long long *>(reinterpret_cast<unsigned long long>(out + a) + (unsigned long long)(e.l & 1) * 16);
                    H2W_GSTORE64(g, t.x); H2W_GSTORE64(g + 1, t.y);
                }
            }
            e.cell0 += (uint64_t)N;
            e.gdst = reinterpret_cast<unsigned long long>(out + e.cc.map(e.cell0)) + (unsigned long long)e.l * 16;
        }
        return e;
    }
    template <class MapFn> static __device__ __forceinline__ const sq16_t *src_ptr_rt(const Em &e, MapFn map, int c, const sq16_t *baseA, const sq16_t *baseB) {
        const BnSrc s = map(c);
        return s.kind == 0 ? e.val + s.idx * BN_SLOT_SQ + (e.l & 1) : s.kind == 1 ? e.tabh + s.idx * 2 : s.kind == 2 ? baseA + s.idx * 2 : baseB + s.idx * 2;
    }
    // A whole Merkle level (two selects, the four state constants, one permutation unit: merkle/mod.rs:66-74) that is not this quad's:
    // step over it - its cells belong to the quad of that unit - and, if this quad's unit is the next one, pick up the node it starts from.
    __device__ __forceinline__ bool level_skip(fr_t &node, bool &zc_ref) {
        if constexpr (MODE != QUAD_EMIT) return false;
        else {
            const int k = unit_local;
            if (k == my_w) return false;
            if (k + 1 == my_w) node = g_load_fr(ustate + (uint64_t)k * 4);
            if (!zc_ref) { cell_off += 1; zc_ref = true; }
            cell_off += 2 * 8 + 4 + BN_PERM_CELLS;
            unit_local = k + 1;
            const int cur = unit_local < last_w ? unit_local : last_w;
            act = cur == my_w;
            return true;
        }
    }
    __device__ __forceinline__ bool tail_skip() const { return MODE == QUAD_VALUES || (MODE == QUAD_EMIT && !act); }      // the cap lookup: cells only (no value anyone uses)
    // One permutation unit of the strand (PoseidonBN254PermutationChip::permute, hash/poseidon_bn254/permutation.rs:190-203).
    __device__ __forceinline__ bool bn_emit_inline(fr_t *st, const ValCfg &cfg, bool &zc_ref) {
        if constexpr (MODE == QUAD_VALUES) { bn_values(st, cfg); unit_local++; return true; }
        else if constexpr (MODE == QUAD_FUSED) { bn_emit_cells<false>(st, cfg, zc_ref); return true; }
        else {
            const int k = unit_local++;
            if (k == my_w) bn_emit_cells<true>(st, cfg, zc_ref);
            else {
                if (k + 1 == my_w) {              // the state this quad's unit starts from: the output of the unit before it
                    const fr_t *u = ustate + (uint64_t)k * 4;
#pragma unroll
                    for (int i = 0; i < 4; i++) st[i] = g_load_fr(u + i);
                }
                if (!zc_ref) { cell_off += 1; zc_ref = true; }      // (the Context's first load_zero cell sits in that unit's first mix)
                cell_off += BN_PERM_CELLS;
            }
            const int cur = unit_local < last_w ? unit_local : last_w;
            act = cur == my_w;
            return true;
        }
    }
    // values phase: lane l owns state element l, times R, in lazy nine-limb form (field.h fr9_t: no reduction below r, no packing between
    // products).  x^5 = three products; a full round = 3 + 4 wavefront-level products; a partial round = THREE: the row product
    // S_0 s0' = S_0 (x^5 + c) is formed as x^4 (s0 S_0) + S_0 c - its second factor beside the chain's second squaring, S_0 c from the table
    // (BK9_X) - so that it does not wait for x^5; the column update of round r - 1 rides in round r's first slot.
    // Sizes, in multiples of r (R / r = 2^7.4; fr9_mont(a, b) < a b / R + r):
    //   full rounds: mix = sum of four products with a constant < 4 (1 + 6 / 169) r, + c < 5.3 r; X2 < 5.3^2 / 169 + 1 = 1.17 r, X4, X5 smaller;
    //   partial rounds: s0' = X5 + c < 2.1 r; U_k = s0' S'_k < 1.02 r; s_k grows by U_k per round: < 5.3 + 56 * 1.02 = 62.5 r = 2^259.6 (top limb
    //   < 2^28); W_j = s_j S_j < (62.5 / 169 + 1) r = 1.37 r; XA = s0 S_0 < 1.04 r; T0 = X4 XA < 1.01 r; s0 = T0 + S_0 c + W_1 + W_2 + W_3 < 6.2 r,
    //   X2 = s0 s0 < 1.23 r; after the partial rounds X2 = s_k s_k < 62.5^2 / 169 + 1 = 24.2 r, X4 < 4.5 r, X5 = X4 s_k < 2.7 r.
    // Operands of a product are normalised, or a sum of two normalised values against a constant (field.h).
    // The output state (canonical) goes to the strand's unit buffer; the partial rounds' S-box values x^2, x^4, x^5 to its S-box buffer AS THEY
    // ARE (times R, below 2 r): k_sbox_canon turns them into the canonical values the emission shows, every one of them in parallel.
    __device__ __noinline__ void bn_values(fr_t *st, const ValCfg &cfg) {
        const int l = l4, lm = l > 0 ? l - 1 : 0; const uint32_t ninv = (uint32_t)cfg.P.ninv & ((1u << 29) - 1);
        fr_t *const sbu = sbx + (uint64_t)unit_local * (BN_PARTIAL_ROUNDS * 3);
        fr9_t S = fr9_mont(fr9_from(fr_sel(l < 2, fr_sel(l == 0, st[0], st[1]), fr_sel(l == 2, st[2], st[3]))), fr9_from(cfg.P.r2), ninv);
        auto sbox = [&]() { const fr9_t X2 = fr9_mont(S, S, ninv), X4 = fr9_mont(X2, X2, ninv); S = fr9_mont(X4, S, ninv); };
        auto mix = [&](int mb) {
            const fr9_t s0 = quad_bcast9<0>(S), s1 = quad_bcast9<1>(S), s2 = quad_bcast9<2>(S), s3 = quad_bcast9<3>(S);
            fr9_t acc = fr9_mont(s0, bnk9(mb + l), ninv);
            acc = fr9_add(fr9_mont(s1, bnk9(mb + 4 + l), ninv), acc);
            acc = fr9_add(fr9_mont(s2, bnk9(mb + 8 + l), ninv), acc);
            S = fr9_norm(fr9_add(fr9_mont(s3, bnk9(mb + 12 + l), ninv), acc));
        };
        S = fr9_norm(fr9_add(S, bnk9(BK_C + l)));
#pragma unroll 1
        for (int half = 0; half < 2; half++) {
            if (half == 1) {
                fr9_t s0p; for (int i = 0; i < 9; i++) s0p.t[i] = 0;      // s0' = s0^5 + c of the previous round (the same on every lane); nothing is pending before round 0
                // the round's five table entries are read a round ahead (an LDS read that is waited for where it is used costs ~130 cycles of a
                // wavefront that has its SIMD to itself)
                struct RoundK { fr9_t kp, k2, k1, kc, kx; };
                auto round_k = [&](int r) {
                    const int ix = BK_S + (BN_WIDTH * 2 - 1) * r, ic = BK_C + (BN_FULL_ROUNDS / 2 + 1) * BN_WIDTH + r;
                    RoundK k; k.kp = bnk9(ix - (BN_WIDTH * 2 - 1) + BN_WIDTH + lm);      // (round 0: s0p = 0, the entry read is in the table)
                    k.k2 = bnk9(ix + (l == 1 ? 0 : l)); k.k1 = bnk9(ix + 1); k.kc = bnk9(ic); k.kx = bnk9(BK9_X + r);
                    return k;
                };
                RoundK kc_ = round_k(0);
#pragma unroll 2
                for (int r = 0; r < BN_PARTIAL_ROUNDS; r++) {
                    //   1: lane 0  X2 = s0 s0         | lanes k  S'_k s0' of round r - 1 (its column update: s_k += ...)
                    //   2: lane 0  X4 = X2 X2         | lane 1   XA = s0 S_0             | lanes 2, 3  S_j s_j
                    //   3: lane 0  T0 = X4 XA         | lane 1   S_1 s_1                 | lanes 2, 3  X5 = X4 s0
                    //   new s0 = T0 + S_0 c + S_1 s_1 + S_2 s_2 + S_3 s_3;  s0' = X5 + c
                    const RoundK kn = round_k(r + 1 < BN_PARTIAL_ROUNDS ? r + 1 : r);
                    fr_t *const sb = sbu + (uint64_t)r * 3;
                    const fr9_t s0b = quad_bcast9<0>(S);
                    const fr9_t A_ = fr9_mont(fr9_sel(l == 0, S, s0p), fr9_sel(l == 0, S, kc_.kp), ninv);
                    if (l == 0) g_store_fr(sb, fr9_pack(A_));
                    const fr9_t X2b = quad_bcast9<0>(A_);
                    S = fr9_add_if(l > 0, S, A_);                                       // lanes k: a sum of two until the end of the round
                    const fr9_t B_ = fr9_mont(fr9_sel(l == 0, X2b, fr9_sel(l == 1, s0b, S)), fr9_sel(l == 0, X2b, kc_.k2), ninv);
                    if (l == 0) g_store_fr(sb + 1, fr9_pack(B_));
                    const fr9_t X4b = quad_bcast9<0>(B_), XAb = quad_bcast9<1>(B_);
                    const fr9_t C_ = fr9_mont(fr9_sel(l == 1, S, X4b), fr9_sel(l == 0, XAb, fr9_sel(l == 1, kc_.k1, s0b)), ninv);
                    if (l == 2) g_store_fr(sb + 2, fr9_pack(C_));
                    s0p = fr9_add(quad_bcast9<2>(C_), kc_.kc);                          // a sum of two: only ever multiplied by a constant
                    fr9_t incl = fr9_add_if(l == 0, fr9_sel(l < 2, C_, B_), kc_.kx);    // T0 + S_0 c, S_1 s_1, S_2 s_2, S_3 s_3
                    incl = fr9_add(incl, quad_dpp9<0xB1>(incl));                         // butterfly over the quad: every lane ends with the sum of the four
                    incl = fr9_add(incl, quad_dpp9<0x4E>(incl));
                    S = fr9_norm(fr9_sel(l == 0, incl, S));
                    kc_ = kn;
                }
                {   // the last round's column update
                    const fr9_t U = fr9_mont(s0p, bnk9(BK_S + (BN_WIDTH * 2 - 1) * (BN_PARTIAL_ROUNDS - 1) + BN_WIDTH + lm), ninv);
                    S = fr9_norm(fr9_add_if(l > 0, S, U));
                }
            }
#pragma unroll 1
            for (int r = 0; r < BN_FULL_ROUNDS / 2; r++) {
                const bool last = r == BN_FULL_ROUNDS / 2 - 1;
                sbox();
                if (!(half == 1 && last)) S = fr9_add(S, bnk9(BK_C + (half == 0 ? (r + 1) * BN_WIDTH : (BN_FULL_ROUNDS / 2 + 1) * BN_WIDTH + BN_PARTIAL_ROUNDS + r * BN_WIDTH) + l));
                mix(half == 0 && last ? BK_P : BK_M);
            }
        }
        fr9_t one9; for (int i = 0; i < 9; i++) one9.t[i] = i == 0 ? 1u : 0u;
        fr_t s = fr9_pack(fr9_mont(S, one9, ninv));                                        // back to canonical: S / R < r + 1
        if (fr_geq_mod(s)) s = fr_sub_mod_raw(s);
        g_store_fr(ustate + (uint64_t)unit_local * 4 + l, s);
        st[0] = quad_bcast<0>(s); st[1] = quad_bcast<1>(s); st[2] = quad_bcast<2>(s); st[3] = quad_bcast<3>(s);
    }
    // PoseidonBN254 permutation with its 4,032 cells emitted by the quad itself.
    // PRE: the partial rounds' S-box values come from the values phase (sbx): two wavefront-level products per partial round, none of them
    // dependent on another of the same round; !PRE (the one-pass kernel): the emitter walks the S-box chain itself, five products.
    template <bool PRE> __device__ __noinline__ void bn_emit_cells(fr_t *st, const ValCfg &cfg, bool &zc_ref) {
        bool zc = zc_ref;                      // by value: a reference would be re-read with a flat load (vmcnt(0)) at every mix
        const int l = l4; const uint64_t ninv = cfg.P.ninv; const fr_t r2 = cfg.P.r2;
        Em e; e.l = l;
        const int val_off = ((threadIdx.x >> 6) * BN_NSLOT * BN_SLOT_SQ) + ((threadIdx.x & 63) >> 2) * 2;
        e.val = s_bn_val + val_off;
        e.tabh = s_bn_tab + (l & 1);
        e.cell0 = cell_off;
        e.cc = cc;
        e.gdst = reinterpret_cast<unsigned long long>(out + e.cc.map(cell_off)) + (unsigned long long)l * 16;
        fr_t s = fr_sel(l < 2, fr_sel(l == 0, st[0], st[1]), fr_sel(l == 2, st[2], st[3]));
        auto exp5_all = [&]() {                          // S-box on every lane
            const fr_t X = fr_mont_mul(s, r2, ninv);
            const fr_t x2 = fr_mont_mul(s, X, ninv), X2 = fr_mont_mul(X, X, ninv);
            const fr_t x4 = fr_mont_mul(x2, X2, ninv), x5 = fr_mont_mul(x4, X, ninv);
            put(e, l, s); put(e, 4 + l, x2); put(e, 8 + l, x4); put(e, 12 + l, x5);
            flush_layer<48>(e, bn_map_exp5, e.tabh, e.tabh);
            s = x5;
        };
        auto ark = [&](int it) {
            const fr_t ns = fr_add(s, bnk(0, BK_C + it + l));
            put(e, l, s); put(e, 4 + l, ns);
            flush_layer<20>(e, bn_map_ark, e.tabh + (BK_C + it) * 2, e.tabh);
            s = ns;
        };
        auto mix = [&](int which) {                      // which 0: M, 1: P
            if (!zc) { flush_layer<1>(e, bn_map_zero, e.tabh, e.tabh); zc = true; }     // the Context's first load_zero (halo2-base caches it afterwards)
            const int mb = which ? BK_P : BK_M;
            put(e, l, s);
            fr_t acc = fr_zero();
#pragma unroll 1
            for (int j = 0; j < 4; j++) {
                const sq16_t a0 = e.val[j * BN_SLOT_SQ], a1 = e.val[j * BN_SLOT_SQ + 1];       // s_j, staged above (an LDS read: j is a loop index)
                fr_t sj; sj.l[0] = a0.x; sj.l[1] = a0.y; sj.l[2] = a1.x; sj.l[3] = a1.y;
                acc = fr_add(fr_mont_mul(sj, bnk(1, mb + 4 * j + l), ninv), acc);
                put(e, 4 + 4 * l + j, acc);
            }
            flush_layer<64>(e, bn_map_mix, e.tabh + mb * 2, e.tabh);
            s = acc;
        };
        ark(0);
#pragma unroll 1
        for (int half = 0; half < 2; half++) {
            if (half == 1) {
                if constexpr (PRE) {
                    // The S-box values come in one value per lane (lane 0: x^2, 1: x^4, 2 and 3: x^5), two rounds per request, requested two rounds
                    // (~110 vector-memory instructions) before they are used.  Loads and stores retire in order on one counter of at most 63 in
                    // flight: a load is back once the cell stores issued before it have drained, and by then it is.  Requested one round ahead
                    // through loop-carried registers the compiler's wait was vmcnt(6) every round, i.e. a drain of the wavefront's whole store queue
                    // (profiles/r03_pmc_emit.txt: 64 % of the wavefront cycles parked in s_waitcnt).
                    const fr_t *sb = sbx + (uint64_t)my_w * (BN_PARTIAL_ROUNDS * 3) + (l < 3 ? l : 2);
                    Wide wd = wide_init(e);
                    lds_sq16 *const valw = (lds_sq16 *)s_bn_val + (threadIdx.x >> 6) * BN_NSLOT * BN_SLOT_SQ;
                    auto round = [&](int r, const fr_t &qv, auto pend_tag) {
                        //   P: lane 0  S_0 * s0'  | lanes j  S_j * s_j        (the sparse row)           s0' = x^5 + c: known from the values phase
                        //   Q:                    | lanes k  S'_k * s0'       (the column update)
                        const int ix = BK_S + (BN_WIDTH * 2 - 1) * r, lm = l > 0 ? l - 1 : 0, ic = BK_C + (BN_FULL_ROUNDS / 2 + 1) * BN_WIDTH + r;
                        constexpr bool pend = !COLS && decltype(pend_tag)::value;      // the previous round's layer is still staged: it leaves between the products
                        lds_sq16 *const tab = (lds_sq16 *)s_bn_tab, *const wA = tab + (ix - (BN_WIDTH * 2 - 1)) * 2, *const wB = tab + (ic - 1) * 2;
                        sq16_t t0[6], t1[6], u0[5], u1[5], v0[5], v1[5];
                        const fr_t x2 = quad_bcast<0>(qv), x4 = quad_bcast<1>(qv), x5 = quad_bcast<2>(qv);
                        const fr_t s0n = fr_add(x5, bnk(0, ic));
                        const fr_t ksxm = bnk(1, ix + l);
                        if (pend) wide_load<0, 6>(wd, valw, tab, wA, wB, t0, t1);
                        const fr_t P_ = fr_mont_mul(fr_sel(l == 0, s0n, s), ksxm, ninv);
                        if (pend) { wide_store<0, 6>(wd, t0, t1); wide_load<6, 11>(wd, valw, tab, wA, wB, u0, u1); }
                        const fr_t Q_ = fr_mont_mul(s0n, bnk(1, ix + BN_WIDTH + lm), ninv);
                        if (pend) { wide_store<6, 11>(wd, u0, u1); wide_load<11, 16>(wd, valw, tab, wA, wB, v0, v1); }
                        fr_t incl = P_;                                      // S[j] * s_j, then the running sums over the quad
                        { const fr_t t = quad_up1(incl); if (l >= 1) incl = fr_add(incl, t); }
                        { const fr_t t = quad_up2(incl); if (l >= 2) incl = fr_add(incl, t); }
                        const fr_t nv = fr_add(Q_, s);
                        if (pend) { wide_store<11, 16>(wd, v0, v1); e.cell0 += 52; e.gdst += 52ull * 32; wd.gq += 52ull * 32; }
                        put(e, l, s);                                        // slots 0-3: the state before the round
                        put_if(e, l == 0, 4, x2); put_if(e, l == 0, 5, x4); put_if(e, l == 0, 6, x5); put_if(e, l == 0, 7, s0n);
                        put(e, 8 + l, incl);
                        put_if(e, l > 0, 12 + lm, nv);
                        const fr_t ns0 = quad_bcast<3>(incl);
                        s = fr_sel(l > 0, nv, ns0);
                        if (COLS || r == BN_PARTIAL_ROUNDS - 1) flush_layer<52>(e, bn_map_partial, e.tabh + ix * 2, e.tabh + ic * 2);
                    };
                    // round 0 (nothing staged before it), rounds 1 .. 54 in pairs (no branch inside: a join makes the compiler's wait conservative),
                    // round 55
                    static_assert(BN_PARTIAL_ROUNDS % 2 == 0, "rounds are walked in pairs");
                    fr_t qc0 = g_load_fr(sb + 3), qc1 = g_load_fr(sb + 6);
                    { const fr_t q = g_load_fr(sb); round(0, q, std::false_type()); }
                    asm volatile("" :: "v"(qc0.l[0]), "v"(qc0.l[2]), "v"(qc1.l[0]), "v"(qc1.l[2]));      // (both are back by now: nothing is pending at the head of the loop)
#pragma unroll 1
                    for (int r = 1; r < BN_PARTIAL_ROUNDS - 1; r += 2) {
                        const int rn = r + 2 < BN_PARTIAL_ROUNDS - 1 ? r + 2 : r;
                        const fr_t qn0 = g_load_fr(sb + 3 * rn), qn1 = g_load_fr(sb + 3 * (rn + 1));
                        round(r, qc0, std::true_type()); round(r + 1, qc1, std::true_type());
                        qc0 = qn0; qc1 = qn1;
                    }
                    { const fr_t q = g_load_fr(sb + 3 * (BN_PARTIAL_ROUNDS - 1)); round(BN_PARTIAL_ROUNDS - 1, q, std::true_type()); }
                } else {
                    // One pass: FOUR wavefront-level Montgomery products per partial round (five until round 3): the row product S_0 s0' = S_0 (x^5 + c)
                    // is formed as x^4 (s0 S_0) + S_0 c - XS = X S_0 beside the squaring of X, S_0 c from the table (bnkc) - so that it does not
                    // wait for x^5, and the column update of round r - 1 (its last three cells) rides in round r's first product:
                    //   1: lane 0  X = s0 R            | lanes k  U_k = S'_k s0' of round r - 1  ->  its nv_k = U_k + s_k, staged into the pending layer
                    //   2: lane 0  X2 = X X            | lane 1   x2 = s0 X      | lane 2  XS = X S_0     | lane 3  S_3 s_3
                    //   3: lane 0  x4 = x2 X2          | lane 1   S_1 s_1        | lane 2  S_2 s_2
                    //   4: lane 0  T0 = x4 XS = S_0 x5 | lane 1   x5 = x4 X
                    //   s0' = x5 + c;  running sums of the row: T0 + S_0 c, + S_1 s_1, + S_2 s_2, + S_3 s_3 (the new s0)
                    fr_t s0p = fr_zero();
                    const int lm = l > 0 ? l - 1 : 0;
#pragma unroll 1
                    for (int r = 0; r < BN_PARTIAL_ROUNDS; r++) {
                        const int ix = BK_S + (BN_WIDTH * 2 - 1) * r, ic = BK_C + (BN_FULL_ROUNDS / 2 + 1) * BN_WIDTH + r;
                        // the previous round's layer (still staged: this round's values are put after product 4) leaves between the products
                        const bool pend = !COLS && r > 0;
                        const sq16_t *pA = e.tabh + (ix - (BN_WIDTH * 2 - 1)) * 2, *pB = e.tabh + (ic - 1) * 2;
                        sq16_t t0[7], t1[7], t2[6], t3[6];
                        if (pend) part_load<0, 7, 52>(e, bn_map_partial, pA, pB, t0);
                        const fr_t A_ = fr_mont_mul(fr_sel(l == 0, s, s0p), fr_sel(l == 0, r2, bnk(1, ix - (BN_WIDTH * 2 - 1) + BN_WIDTH + lm)), ninv);      // (round 0: s0p = 0, the entry read is in the table)
                        if (r > 0) { const fr_t nv = fr_add(A_, s); put_if(e, l > 0, 12 + lm, nv); s = fr_sel(l > 0, nv, s); }
                        if (COLS && r > 0) flush_layer<52>(e, bn_map_partial, pA, pB);
                        if (pend) { part_store<0, 7, 52>(e, t0); part_load<7, 14, 52>(e, bn_map_partial, pA, pB, t1); }
                        const fr_t Xb = quad_bcast<0>(A_), s0b = quad_bcast<0>(s);
                        const fr_t B_ = fr_mont_mul(fr_sel(l == 1, s0b, fr_sel(l == 3, s, Xb)), fr_sel(l < 2, Xb, bnk(1, ix + (l == 2 ? 0 : l))), ninv);
                        if (pend) { part_store<7, 14, 52>(e, t1); part_load<14, 20, 52>(e, bn_map_partial, pA, pB, t2); }
                        const fr_t x2b = quad_bcast<1>(B_), XSb = quad_bcast<2>(B_);
                        const fr_t C_ = fr_mont_mul(fr_sel(l == 0, x2b, s), fr_sel(l == 0, B_, bnk(1, ix + l)), ninv);
                        if (pend) { part_store<14, 20, 52>(e, t2); part_load<20, 26, 52>(e, bn_map_partial, pA, pB, t3); }
                        const fr_t D_ = fr_mont_mul(quad_bcast<0>(C_), fr_sel(l == 0, XSb, Xb), ninv);
                        if (pend) { part_store<20, 26, 52>(e, t3); e.cell0 += 52; e.gdst += 52ull * 32; }
                        const fr_t s0n = fr_add(quad_bcast<1>(D_), bnk(0, ic));
                        put(e, l, s);                                        // slots 0-3: the state before the round
                        put_if(e, l == 1, 4, B_); put_if(e, l == 0, 5, C_); put_if(e, l == 1, 6, D_); put_if(e, l == 0, 7, s0n);
                        fr_t incl = fr_sel(l == 0, fr_add(D_, bnkc(r)), fr_sel(l == 3, B_, C_));      // S_0 s0', then S_j s_j: the running sums over the quad
                        { const fr_t t = quad_up1(incl); if (l >= 1) incl = fr_add(incl, t); }
                        { const fr_t t = quad_up2(incl); if (l >= 2) incl = fr_add(incl, t); }
                        put(e, 8 + l, incl);
                        s0p = s0n;
                        s = fr_sel(l == 0, quad_bcast<3>(incl), s);
                    }
                    {   // the last round's column update, then its layer
                        const fr_t U = fr_mont_mul(s0p, bnk(1, BK_S + (BN_WIDTH * 2 - 1) * (BN_PARTIAL_ROUNDS - 1) + BN_WIDTH + lm), ninv);
                        const fr_t nv = fr_add(U, s); put_if(e, l > 0, 12 + lm, nv); s = fr_sel(l > 0, nv, s);
                        const int ixl = BK_S + (BN_WIDTH * 2 - 1) * (BN_PARTIAL_ROUNDS - 1), icl = BK_C + (BN_FULL_ROUNDS / 2 + 1) * BN_WIDTH + BN_PARTIAL_ROUNDS - 1;
                        flush_layer<52>(e, bn_map_partial, e.tabh + ixl * 2, e.tabh + icl * 2);
                    }
                }
            }
            // full_rounds(is_first = half == 0) (:112-160): load_constant of M and P, then 4 x (x^5, [ark], mix)
            flush_layer<32>(e, bn_map_consts, e.tabh + BK_M * 2, e.tabh);
#pragma unroll 1
            for (int r = 0; r < BN_FULL_ROUNDS / 2; r++) {
                const bool last = r == BN_FULL_ROUNDS / 2 - 1;
                // (the layers' ~80 source addresses are loop invariants the compiler would hoist out of this loop and then spill - every reload
                //  a drain of the store queue; an opaque offset per round keeps them base + immediate)
                { int vo = val_off, ll = l; asm volatile("" : "+v"(vo), "+v"(ll)); e.val = s_bn_val + vo; e.tabh = s_bn_tab + (ll & 1); e.l = ll; }
                exp5_all();
                if (!(half == 1 && last)) ark(half == 0 ? (r + 1) * BN_WIDTH : (BN_FULL_ROUNDS / 2 + 1) * BN_WIDTH + BN_PARTIAL_ROUNDS + r * BN_WIDTH);
                mix(half == 0 && last ? 1 : 0);
            }
        }
        st[0] = quad_bcast<0>(s); st[1] = quad_bcast<1>(s); st[2] = quad_bcast<2>(s); st[3] = quad_bcast<3>(s);
        cell_off = e.cell0; zc_ref = zc; cc = e.cc;
    }
};

}  // namespace h2w
