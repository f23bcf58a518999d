// prover.hip — SURVEY §8(f) row 3: the step BEFORE the path.  Valid synthetic FRI instances generated on the GPU, written in
// the flat proof layout h2w_fri_witness_batch consumes (chips.h ProofLayout = WitnessChip load order, witness/mod.rs:236-294).
//
// What the reference gets from starky's prover (stark/mod.rs:405-426) is produced here with the plonky2 prover conventions the
// verifier gadget checks (fri/mod.rs:148-444, merkle/mod.rs:57-115, challenger/mod.rs:168-222):
//   * PolynomialBatch: LDE on the coset 7<w>, |<w>| = 2^(degree_bits + rate_bits); Merkle leaves in bit-reversed order
//     (a decimation-in-frequency NTT leaves its output in exactly that order, so nothing is ever permuted);
//   * transcript order trace cap, permutation challenges, permutation cap, alphas, quotient cap, zeta, openings, alpha,
//     commit-phase caps / betas, final polynomial, proof-of-work witness, query indices;
//   * batched quotient sum_b (G_b - G_b(z_b)) / (X - z_b) with the verifier's alpha shifts (fri/mod.rs:169-220), as a suffix scan;
//   * commit phase: leaves of 2^arity_bits extension evaluations, coefficient folding with beta, shift <- shift^arity.
// Device work: NTTs, leaf hashing and tree levels (one lane per permutation), opening evaluation (blocked Horner), the quotient
// scan, folding, the proof-of-work search and the query gather.  Host work: the serial sponge (tens of permutations).
// All arithmetic is exact integer arithmetic, so any schedule gives the oracle's (oracle/prover.inc) words bit for bit.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>
#include "chips.h"
#include "common.h"

namespace h2w {
namespace {

struct BnMont { fr_t c[88], s[392], m[4][4], p[4][4]; };                 // PoseidonBN254 tables in Montgomery form (x * 2^261)
struct ProverConsts { h2w_poseidon_consts_t k; BnMont bn; fr_t r2; uint64_t ninv; };
struct H4 { uint64_t w[4]; };

HD gle_t gle_add(gle_t a, gle_t b) { gle_t r; r.c[0] = gl_add(a.c[0], b.c[0]); r.c[1] = gl_add(a.c[1], b.c[1]); return r; }
HD gle_t gle_sub(gle_t a, gle_t b) { gle_t r; r.c[0] = gl_sub(a.c[0], b.c[0]); r.c[1] = gl_sub(a.c[1], b.c[1]); return r; }
HD gle_t gle_scale(gle_t a, uint64_t b) { gle_t r; r.c[0] = gl_mul(a.c[0], b); r.c[1] = gl_mul(a.c[1], b); return r; }
HD gle_t gle_of(uint64_t a) { gle_t r; r.c[0] = a; r.c[1] = 0; return r; }
HD gle_t gle_pow(gle_t a, uint64_t e) { gle_t acc = gle_of(1); while (e) { if (e & 1) acc = gle_mul(acc, a); a = gle_mul(a, a); e >>= 1; } return acc; }

// ---------------------------------------------------------------- Goldilocks Poseidon, plonky2 "fast" layout (hash/poseidon/permutation.rs:43-314)
// dot products of full-width words: 128-bit accumulator + overflow count, reduced once (2^128 = -2^32 mod p)
struct GlAcc { u128 lo; uint32_t carries; };
HD void glacc_mul(GlAcc &a, uint64_t x, uint64_t y) { const u128 pr = (u128)x * y; a.lo += pr; a.carries += a.lo < pr; }
HD uint64_t glacc_reduce(const GlAcc &a) { return gl_sub(gl_reduce128(a.lo), (uint64_t)a.carries << 32); }
// SMALL_MDS: every MDS_MATRIX_CIRC / _DIAG entry is below 2^28 (plonky2's are <= 41): a row is then two 64-bit accumulations over the
// 32-bit halves of the state (13 terms < 2^32 * 2^28 each) and ONE reduction, instead of 13 reduced 64x64 products - the full
// rounds' MDS layers are two thirds of a permutation's instructions otherwise.  Same values either way (exact arithmetic).
template <bool SMALL_MDS> HDN inline void gl_permute_t(const h2w_poseidon_consts_t *k, uint64_t *st) {
    int rc = 0;
    for (int half = 0; half < 2; half++) {
        if (half == 1) {
#pragma unroll
            for (int i = 0; i < 12; i++) st[i] = gl_add(st[i], k->fast_partial_first_round_constant[i]);
            uint64_t res[12];
            res[0] = st[0];
#pragma unroll
            for (int c = 1; c < 12; c++) {
                GlAcc a; a.lo = 0; a.carries = 0;
#pragma unroll
                for (int r = 1; r < 12; r++) glacc_mul(a, k->fast_partial_round_initial_matrix[r - 1][c - 1], st[r]);
                res[c] = glacc_reduce(a);
            }
#pragma unroll
            for (int c = 0; c < 12; c++) st[c] = res[c];
            const uint64_t m00 = gl_add(k->mds_circ[0], k->mds_diag[0]);
#pragma unroll 1
            for (int r = 0; r < N_PARTIAL_ROUNDS; r++) {
                uint64_t x = st[0], x2 = gl_mul(x, x), x4 = gl_mul(x2, x2), x6 = gl_mul(x4, x2);
                const uint64_t s0 = gl_add(gl_mul(x6, x), k->fast_partial_round_constants[r]);
                GlAcc a; a.lo = (u128)m00 * s0; a.carries = 0;
#pragma unroll
                for (int i = 1; i < 12; i++) glacc_mul(a, k->fast_partial_round_w_hats[r][i - 1], st[i]);
#pragma unroll
                for (int i = 1; i < 12; i++) st[i] = gl_muladd(k->fast_partial_round_vs[r][i - 1], s0, st[i]);
                st[0] = glacc_reduce(a);
            }
            rc += N_PARTIAL_ROUNDS;
        }
#pragma unroll 1
        for (int f = 0; f < HALF_N_FULL_ROUNDS; f++) {
#pragma unroll
            for (int i = 0; i < 12; i++) {
                uint64_t x = gl_add(st[i], k->all_round_constants[i + 12 * rc]);
                uint64_t x2 = gl_mul(x, x), x4 = gl_mul(x2, x2), x6 = gl_mul(x4, x2); st[i] = gl_mul(x6, x);
            }
            uint64_t res[12];
            if (SMALL_MDS) {
                uint32_t lo32[12], hi32[12];
#pragma unroll
                for (int i = 0; i < 12; i++) { lo32[i] = (uint32_t)st[i]; hi32[i] = (uint32_t)(st[i] >> 32); }
#pragma unroll
                for (int r = 0; r < 12; r++) {
                    const uint32_t dg = (uint32_t)k->mds_diag[r];
                    uint64_t lo = (uint64_t)lo32[r] * dg, hi = (uint64_t)hi32[r] * dg;
#pragma unroll
                    for (int i = 0; i < 12; i++) { const uint32_t c = (uint32_t)k->mds_circ[i]; lo += (uint64_t)lo32[(i + r) % 12] * c; hi += (uint64_t)hi32[(i + r) % 12] * c; }
                    res[r] = gl_reduce128((u128)lo + ((u128)hi << 32));
                }
            } else {
#pragma unroll
                for (int r = 0; r < 12; r++) {
                    uint64_t acc = 0;
#pragma unroll
                    for (int i = 0; i < 12; i++) acc = gl_muladd(k->mds_circ[i], st[(i + r) % 12], acc);
                    res[r] = gl_muladd(k->mds_diag[r], st[r], acc);
                }
            }
#pragma unroll
            for (int r = 0; r < 12; r++) st[r] = res[r];
            rc++;
        }
    }
}

HDN inline bool gl_mds_is_small(const h2w_poseidon_consts_t *k) {
    bool small = true;
    for (int i = 0; i < 12; i++) small = small && k->mds_circ[i] < (1ull << 28) && k->mds_diag[i] < (1ull << 28);
    return small;
}
HDN inline void gl_permute(const h2w_poseidon_consts_t *k, uint64_t *st) { if (gl_mds_is_small(k)) gl_permute_t<true>(k, st); else gl_permute_t<false>(k, st); }

// ---------------------------------------------------------------- PoseidonBN254 (hash/poseidon_bn254/permutation.rs:48-203), Montgomery-form state
HNI fr_t mmul(fr_t a, fr_t b, uint64_t ninv) { return fr_mont_mul(a, b, ninv); }
HDN inline fr_t bn_pow5(fr_t x, uint64_t ni) { fr_t x2 = mmul(x, x, ni), x4 = mmul(x2, x2, ni); return mmul(x4, x, ni); }
HDN inline void bn_mix(fr_t *st, const fr_t m[4][4], uint64_t ni) {
    fr_t ns[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        ns[i] = mmul(m[0][i], st[0], ni);
#pragma unroll
        for (int j = 1; j < 4; j++) ns[i] = fr_add(ns[i], mmul(m[j][i], st[j], ni));
    }
#pragma unroll
    for (int i = 0; i < 4; i++) st[i] = ns[i];
}
HDN inline void bn_permute_m(const ProverConsts *K, fr_t *st) {
    const uint64_t ni = K->ninv; const BnMont &B = K->bn;
#pragma unroll
    for (int j = 0; j < 4; j++) st[j] = fr_add(st[j], B.c[j]);
    for (int half = 0; half < 2; half++) {
        if (half == 1) {
#pragma unroll 1
            for (int r = 0; r < BN_PARTIAL_ROUNDS; r++) {
                st[0] = fr_add(bn_pow5(st[0], ni), B.c[(BN_FULL_ROUNDS / 2 + 1) * BN_WIDTH + r]);
                fr_t ns0 = mmul(B.s[7 * r], st[0], ni);
#pragma unroll
                for (int j = 1; j < 4; j++) ns0 = fr_add(ns0, mmul(B.s[7 * r + j], st[j], ni));
#pragma unroll
                for (int kk = 1; kk < 4; kk++) st[kk] = fr_add(st[kk], mmul(B.s[7 * r + 4 + kk - 1], st[0], ni));
                st[0] = ns0;
            }
        }
#pragma unroll 1
        for (int i = 0; i < BN_FULL_ROUNDS / 2 - 1; i++) {
            const int it = half == 0 ? (i + 1) * BN_WIDTH : (BN_FULL_ROUNDS / 2 + 1) * BN_WIDTH + BN_PARTIAL_ROUNDS + i * BN_WIDTH;
#pragma unroll
            for (int j = 0; j < 4; j++) st[j] = fr_add(bn_pow5(st[j], ni), B.c[it + j]);
            bn_mix(st, B.m, ni);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) st[j] = bn_pow5(st[j], ni);
        if (half == 0) {
#pragma unroll
            for (int j = 0; j < 4; j++) st[j] = fr_add(st[j], B.c[(BN_FULL_ROUNDS / 2) * BN_WIDTH + j]);
            bn_mix(st, B.p, ni);
        } else bn_mix(st, B.m, ni);
    }
}
HDN inline fr_t to_mont(const ProverConsts *K, fr_t x) { return mmul(x, K->r2, K->ninv); }
HDN inline fr_t from_mont(const ProverConsts *K, fr_t x) { return mmul(x, fr_from_u64(1), K->ninv); }

// ---------------------------------------------------------------- hashers (hash/poseidon/hash.rs:63-215, hash/poseidon_bn254/hash.rs:67-210): a hash is 4 words
template <int MAXN> HDN inline H4 hash_no_pad(const ProverConsts *K, int mode, const uint64_t (&in)[MAXN], int n) {
    H4 h;
    if (mode == 0) {
        uint64_t st[12];
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = 0;
#pragma unroll
        for (int off = 0; off < MAXN; off += SPONGE_RATE) {
            if (off < n) {
#pragma unroll
                for (int i = 0; i < SPONGE_RATE; i++) if (off + i < MAXN && off + i < n) st[i] = in[off + i];
                gl_permute(&K->k, st);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) h.w[i] = st[i];
    } else {
        fr_t st[4];
#pragma unroll
        for (int i = 0; i < 4; i++) st[i] = fr_zero();
#pragma unroll
        for (int off = 0; off < MAXN; off += 9) {
            if (off < n) {
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const int o = off + 3 * j;
                    if (o < MAXN && o < n) {
                        fr_t v = fr_zero();
#pragma unroll
                        for (int l = 0; l < 3; l++) if (o + l < MAXN && o + l < n) v.l[l] = in[o + l];
                        st[j + 1] = to_mont(K, v);
                    }
                }
                bn_permute_m(K, st);
            }
        }
        const fr_t o = from_mont(K, st[0]);
#pragma unroll
        for (int i = 0; i < 4; i++) h.w[i] = o.l[i];
    }
    return h;
}
template <int MAXN> HDN inline H4 hash_or_noop(const ProverConsts *K, int mode, const uint64_t (&in)[MAXN], int n) {
    if (n > (mode == 0 ? 4 : 3)) return hash_no_pad<MAXN>(K, mode, in, n);
    H4 h;
#pragma unroll
    for (int i = 0; i < 4; i++) h.w[i] = (i < MAXN && i < n) ? in[i < MAXN ? i : 0] : 0;
    return h;
}
HDN inline H4 two_to_one(const ProverConsts *K, int mode, const H4 &l, const H4 &r) {
    H4 h;
    if (mode == 0) {
        uint64_t st[12];
#pragma unroll
        for (int i = 0; i < 4; i++) { st[i] = l.w[i]; st[4 + i] = r.w[i]; st[8 + i] = 0; }
        gl_permute(&K->k, st);
#pragma unroll
        for (int i = 0; i < 4; i++) h.w[i] = st[i];
    } else {
        fr_t st[4], a, b;
#pragma unroll
        for (int i = 0; i < 4; i++) { a.l[i] = l.w[i]; b.l[i] = r.w[i]; }
        st[0] = fr_zero(); st[1] = fr_zero(); st[2] = to_mont(K, a); st[3] = to_mont(K, b);
        bn_permute_m(K, st);
        const fr_t o = from_mont(K, st[0]);
#pragma unroll
        for (int i = 0; i < 4; i++) h.w[i] = o.l[i];
    }
    return h;
}

// ================================================================= kernels
__global__ void k_twiddles(uint64_t *tw, uint64_t half_n, uint64_t w) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < half_n) tw[j] = gl_exp(w, j);
}
// out[p][i] = coef[p][i] * shift^i for i < nc, 0 above (zero padding = low-degree extension)
__global__ void k_scale_pad(const uint64_t *coef, uint64_t nc, uint64_t shift, uint64_t n, uint64_t *out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, p = blockIdx.y;
    if (i < n) out[p * n + i] = i < nc ? gl_mul(coef[p * nc + i], gl_exp(shift, i)) : 0;
}
// one decimation-in-frequency stage (block length len = 2^s) of every array of the batch; tw[j << tw_shift] = w_len^j
__global__ void k_dif_stage(uint64_t *a, uint64_t n, int s, const uint64_t *tw, int tw_shift) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n / 2) return;
    uint64_t *x = a + (uint64_t)blockIdx.y * n;
    const uint64_t half = 1ull << (s - 1), blk = t >> (s - 1), j = t & (half - 1), i0 = (blk << s) + j, i1 = i0 + half;
    const uint64_t u = x[i0], v = x[i1];
    x[i0] = gl_add(u, v); x[i1] = gl_mul(gl_sub(u, v), tw[j << tw_shift]);
}
// the last tb stages (len = 2^tb .. 2) inside LDS tiles of 2^tb contiguous elements
constexpr int NTT_TILE_BITS = 11;
__global__ __launch_bounds__(256) void k_dif_local(uint64_t *a, uint64_t n, int tb, const uint64_t *tw, int max_bits) {
    __shared__ uint64_t tile[1 << NTT_TILE_BITS];
    const uint32_t T = 1u << tb;
    uint64_t *x = a + (uint64_t)blockIdx.y * n + (uint64_t)blockIdx.x * T;
    for (uint32_t i = threadIdx.x; i < T; i += 256) tile[i] = x[i];
    __syncthreads();
    for (int s = tb; s >= 1; s--) {
        const uint32_t half = 1u << (s - 1);
        for (uint32_t t = threadIdx.x; t < T / 2; t += 256) {
            const uint32_t blk = t >> (s - 1), j = t & (half - 1), i0 = (blk << s) + j, i1 = i0 + half;
            const uint64_t u = tile[i0], v = tile[i1];
            tile[i0] = gl_add(u, v); tile[i1] = gl_mul(gl_sub(u, v), tw[(uint64_t)j << (max_bits - s)]);
        }
        __syncthreads();
    }
    for (uint32_t i = threadIdx.x; i < T; i += 256) x[i] = tile[i];
}
// Every kernel below takes a batch of proofs (blockIdx.y or .z = proof): a batch advances through the phases in lockstep, so
// the latency-bound pieces (upper tree levels, small fold steps, the host transcript's round trips) are paid once per batch.
// leaf j of an initial oracle = hash_or_noop(values of its polynomials at position j of the bit-reversed LDE)
__global__ void k_leaves_initial(const ProverConsts *K, int mode, const uint64_t *vals, int npoly, uint64_t n, H4 *leaves, uint64_t val_stride, uint64_t leaf_stride) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    vals += blockIdx.y * val_stride; leaves += blockIdx.y * leaf_stride;
    uint64_t buf[MAX_BATCH_POLYS];
#pragma unroll
    for (int p = 0; p < MAX_BATCH_POLYS; p++) buf[p] = p < npoly ? vals[(uint64_t)p * n + j] : 0;
    leaves[j] = hash_or_noop<MAX_BATCH_POLYS>(K, mode, buf, npoly);
}
// commit-phase leaf j = hash of the 2^ab extension evaluations j*2^ab .. (v = the two coordinates [2][n], bit-reversed order)
__global__ void k_leaves_fri(const ProverConsts *K, int mode, const uint64_t *v, int ab, uint64_t n, H4 *leaves, uint64_t leaf_stride) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= (n >> ab)) return;
    const uint64_t *v0 = v + blockIdx.y * 2 * n, *v1 = v0 + n; leaves += blockIdx.y * leaf_stride;
    uint64_t buf[2 * MAX_ARITY]; const int ar = 1 << ab;
#pragma unroll
    for (int t = 0; t < MAX_ARITY; t++) { buf[2 * t] = t < ar ? v0[(j << ab) + t] : 0; buf[2 * t + 1] = t < ar ? v1[(j << ab) + t] : 0; }
    leaves[j] = hash_or_noop<2 * MAX_ARITY>(K, mode, buf, 2 * ar);
}
// one level of gridDim.y trees that sit tree_stride hashes apart (trees of the same shape are built level by level together)
__global__ void k_tree_level(const ProverConsts *K, int mode, const H4 *in, H4 *out, uint64_t m, uint64_t tree_stride) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    in += blockIdx.y * tree_stride; out += blockIdx.y * tree_stride;
    out[i] = two_to_one(K, mode, in[2 * i], in[2 * i + 1]);
}
// partial[proof][p][block] = sum over the block's coefficients c_i x^i (blocked Horner, then a block reduction); x = xs[proof]
constexpr int EVAL_PER_THREAD = 16;
__global__ __launch_bounds__(256) void k_eval_partial(const uint64_t *coef, uint64_t N, const gle_t *xs, gle_t *partial, uint64_t coef_stride, uint64_t partial_stride) {
    __shared__ gle_t red[256];
    const uint64_t p = blockIdx.y, base = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * EVAL_PER_THREAD;
    const gle_t x = xs[blockIdx.z];
    gle_t acc = gle_of(0);
    if (base < N) {
        const uint64_t *c = coef + blockIdx.z * coef_stride + p * N + base;
        for (int i = EVAL_PER_THREAD - 1; i >= 0; i--) acc = gle_add(gle_mul(acc, x), gle_of(base + i < N ? c[i] : 0));
        acc = gle_mul(acc, gle_pow(x, base));
    }
    red[threadIdx.x] = acc; __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] = gle_add(red[threadIdx.x], red[threadIdx.x + s]); __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.z * partial_stride + p * gridDim.x + blockIdx.x] = red[0];
}
// batched quotient, step 1: T[m] = G[m] z^m with G = sum_i alpha^i f_i (minus G(z) at m = 0); block sums.  One QuotArgs per proof.
struct QuotArgs { const uint64_t *coef; uint64_t N; int npoly; gle_t ap[MAX_BATCH_POLYS]; gle_t gz, z, zinv, an; };
constexpr int SCAN_PER_THREAD = 8, SCAN_TILE = 256 * SCAN_PER_THREAD;
__global__ __launch_bounds__(256) void k_quot_terms(const QuotArgs *Av, gle_t *T, gle_t *totals) {
    __shared__ gle_t red[256];
    const QuotArgs &A = Av[blockIdx.y]; T += blockIdx.y * A.N; totals += (uint64_t)blockIdx.y * gridDim.x;
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_PER_THREAD;
    gle_t sum = gle_of(0);
    if (base < A.N) {
        const gle_t z = A.z; gle_t zp = gle_pow(z, base);
        for (int e = 0; e < SCAN_PER_THREAD && base + e < A.N; e++) {
            const uint64_t m = base + e;
            gle_t g = gle_of(0);
            for (int i = 0; i < A.npoly; i++) g = gle_add(g, gle_scale(A.ap[i], A.coef[(uint64_t)i * A.N + m]));
            if (m == 0) g = gle_sub(g, A.gz);
            const gle_t t = gle_mul(g, zp);
            T[m] = t; sum = gle_add(sum, t); zp = gle_mul(zp, z);
        }
    }
    red[threadIdx.x] = sum; __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] = gle_add(red[threadIdx.x], red[threadIdx.x + s]); __syncthreads(); }
    if (threadIdx.x == 0) totals[blockIdx.x] = red[0];
}
// step 2: carry[b] = sum of the totals of the blocks after b (one block per proof)
__global__ void k_suffix_totals(const gle_t *totals, gle_t *carry, int nblk) {
    if (threadIdx.x) return;
    totals += (uint64_t)blockIdx.x * nblk; carry += (uint64_t)blockIdx.x * nblk;
    gle_t acc = gle_of(0);
    for (int b = nblk - 1; b >= 0; b--) { carry[b] = acc; acc = gle_add(acc, totals[b]); }
}
// step 3: H[j] = sum_{m >= j} T[m];  Q[j-1] = H[j] z^-j;  F[j-1] = F[j-1] * alpha^{n_b} + Q[j-1]  (Q[N-1] = 0)
__global__ __launch_bounds__(256) void k_quot_finish(const QuotArgs *Av, const gle_t *T, const gle_t *carry, gle_t *F, uint64_t F_stride) {
    __shared__ gle_t tot[256];
    const QuotArgs &A = Av[blockIdx.y]; T += blockIdx.y * A.N; carry += (uint64_t)blockIdx.y * gridDim.x; F += blockIdx.y * F_stride;
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_PER_THREAD;
    gle_t loc[SCAN_PER_THREAD]; gle_t acc = gle_of(0);
#pragma unroll
    for (int e = SCAN_PER_THREAD - 1; e >= 0; e--) { if (base + e < A.N) acc = gle_add(acc, T[base + e]); loc[e] = acc; }
    tot[threadIdx.x] = acc; __syncthreads();
    if (threadIdx.x == 0) { gle_t run = carry[blockIdx.x]; for (int t = 255; t >= 0; t--) { const gle_t mine = tot[t]; tot[t] = run; run = gle_add(run, mine); } }
    __syncthreads();
    const gle_t after = tot[threadIdx.x];
    if (base < A.N) {
        const gle_t zinv = A.zinv, an = A.an; gle_t zi = gle_pow(zinv, base);
#pragma unroll
        for (int e = 0; e < SCAN_PER_THREAD; e++) {
            const uint64_t j = base + e;
            if (j < A.N) {
                if (j >= 1) { const gle_t q = gle_mul(gle_add(loc[e], after), zi); F[j - 1] = gle_add(gle_mul(F[j - 1], an), q); }
                if (j == A.N - 1) F[j] = gle_mul(F[j], an);
                zi = gle_mul(zi, zinv);
            }
        }
    }
}
// v[0][i] = F[i].c0 shift^i, v[1][i] = F[i].c1 shift^i (the coset is in the base field: the two coordinates transform separately)
__global__ void k_split_scale(const gle_t *F, uint64_t n, uint64_t shift, uint64_t *v, uint64_t F_stride) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F += blockIdx.y * F_stride; v += blockIdx.y * 2 * n;
    const uint64_t s = gl_exp(shift, i); const gle_t f = F[i];
    v[i] = gl_mul(f.c[0], s); v[n + i] = gl_mul(f.c[1], s);
}
__global__ void k_fold(const gle_t *Fin, gle_t *Fout, uint64_t nl, int ab, const gle_t *betas, uint64_t F_stride) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nl) return;
    Fin += blockIdx.y * F_stride; Fout += blockIdx.y * F_stride;
    const gle_t beta = betas[blockIdx.y]; gle_t acc = gle_of(0);
    for (int t = (1 << ab) - 1; t >= 0; t--) acc = gle_add(gle_mul(acc, beta), Fin[(j << ab) + t]);
    Fout[j] = acc;
}
// proof of work: the smallest witness whose challenge has pow_bits leading zero bits (challenger/mod.rs:92-108 pops the LAST rate word)
struct PowArgs { uint64_t state[12]; uint64_t tail[SPONGE_RATE]; int ntail, pow_bits; };
__global__ void k_pow(const ProverConsts *K, const PowArgs *Av, uint64_t first, unsigned long long *best) {
    if (best[blockIdx.y] < first) return;                       // found in an earlier range
    const PowArgs &A = Av[blockIdx.y];
    const uint64_t w = first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t st[12];
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = A.state[i];
#pragma unroll
    for (int i = 0; i < SPONGE_RATE; i++) if (i < A.ntail) st[i] = A.tail[i]; else if (i == A.ntail) st[i] = w;
    gl_permute(&K->k, st);
    if (A.pow_bits == 0 || (st[SPONGE_RATE - 1] >> (64 - A.pow_bits)) == 0) atomicMin(&best[blockIdx.y], (unsigned long long)w);
}
// query openings: one block per (query, proof)
struct GatherArgs {
    const uint64_t *lde; uint64_t L, lde_stride; int lb, capb, no, npoly[3], poly0[3]; const H4 *tree; uint64_t tree_stride;
    int n_steps, ab[MAX_STEPS]; const uint64_t *fv[MAX_STEPS]; uint64_t fn[MAX_STEPS]; const H4 *ftree[MAX_STEPS];
    uint64_t *proof; uint64_t proof_stride, q_base, q_words, init_off[3], step_off[MAX_STEPS]; const uint64_t *x_index;
};
HD uint64_t level_off(int bits, int l) { return (2ull << bits) - (2ull << (bits - l)); }     // hashes before level l of a tree with 2^bits leaves
__global__ void k_gather(GatherArgs G) {
    const uint64_t pr = blockIdx.y;
    uint64_t *w = G.proof + pr * G.proof_stride + G.q_base + (uint64_t)blockIdx.x * G.q_words;
    const uint64_t x = G.x_index[pr * gridDim.x + blockIdx.x];
    for (int o = 0; o < G.no; o++) {
        uint64_t *wo = w + G.init_off[o];
        const H4 *tree = G.tree + (pr * G.no + o) * G.tree_stride;
        for (int p = threadIdx.x; p < G.npoly[o]; p += blockDim.x) wo[p] = G.lde[pr * G.lde_stride + (uint64_t)(G.poly0[o] + p) * G.L + x];
        for (int l = threadIdx.x; l < G.lb - G.capb; l += blockDim.x) {
            const H4 h = tree[level_off(G.lb, l) + ((x >> l) ^ 1)];
            for (int i = 0; i < 4; i++) wo[G.npoly[o] + 4 * l + i] = h.w[i];
        }
    }
    uint64_t idx = x; int bits = G.lb;
    for (int st = 0; st < G.n_steps; st++) {
        const int ab = G.ab[st], ar = 1 << ab; const uint64_t coset = idx >> ab; bits -= ab;
        uint64_t *ws = w + G.step_off[st];
        const uint64_t *fv = G.fv[st] + pr * 2 * G.fn[st]; const H4 *ft = G.ftree[st] + pr * 2 * (G.fn[st] >> ab);
        for (int t = threadIdx.x; t < ar; t += blockDim.x) { ws[2 * t] = fv[(coset << ab) + t]; ws[2 * t + 1] = fv[G.fn[st] + (coset << ab) + t]; }
        for (int l = threadIdx.x; l < bits - G.capb; l += blockDim.x) {
            const H4 h = ft[level_off(bits, l) + ((coset >> l) ^ 1)];
            for (int i = 0; i < 4; i++) ws[2 * ar + 4 * l + i] = h.w[i];
        }
        idx = coset;
    }
}

// ================================================================= host: transcript (challenger/mod.rs:45-277) and driver
struct HostChallenger {
    const h2w_poseidon_consts_t *k; uint64_t state[12]; std::vector<uint64_t> in; uint64_t out[SPONGE_RATE]; int n_out;
    explicit HostChallenger(const h2w_poseidon_consts_t *kk) : k(kk), n_out(0) { memset(state, 0, sizeof(state)); }
    void observe(uint64_t v) { n_out = 0; in.push_back(v); }
    void observe_hash(int mode, const H4 &h) {
        if (mode == 0) { for (int i = 0; i < 4; i++) observe(h.w[i]); return; }
        fr_t v; for (int i = 0; i < 4; i++) v.l[i] = h.w[i];
        for (int i = 0; i < 5; i++) observe(fr_bits(v, 56 * i, 56));                       // hash/poseidon_bn254/hash.rs:31-43
    }
    void observe_ext(const gle_t &e) { observe(e.c[0]); observe(e.c[1]); }
    void absorb() {
        if (in.empty()) return;
        for (size_t off = 0; off < in.size(); off += SPONGE_RATE) {
            const size_t len = in.size() - off < (size_t)SPONGE_RATE ? in.size() - off : (size_t)SPONGE_RATE;
            memcpy(state, in.data() + off, len * 8); gl_permute(k, state);
        }
        memcpy(out, state, sizeof(out)); n_out = SPONGE_RATE; in.clear();
    }
    uint64_t challenge() {
        absorb();
        if (n_out == 0) { gl_permute(k, state); memcpy(out, state, sizeof(out)); n_out = SPONGE_RATE; }
        return out[--n_out];
    }
    gle_t ext_challenge() { gle_t r; r.c[0] = challenge(); r.c[1] = challenge(); return r; }
};

}  // namespace
}  // namespace h2w

using namespace h2w;

struct h2w_prover {
    h2w_shape_t shape; Derived d; ProofLayout pl; int device = 0;
    ProverConsts hk; ProverConsts *dk = nullptr;
    int np = 0, poly0[3] = {0, 0, 0};
    uint64_t N = 0, L = 0, cap = 0;                              // cap: proofs the batch buffers below are sized for
    uint64_t *tw = nullptr, *lde = nullptr, *fv[MAX_STEPS] = {nullptr}, *x_index = nullptr;
    H4 *tree = nullptr, *ftree[MAX_STEPS] = {nullptr};
    gle_t *Fa = nullptr, *Fb = nullptr, *T = nullptr, *totals = nullptr, *carry = nullptr, *partial = nullptr, *pts = nullptr;
    QuotArgs *qargs = nullptr; PowArgs *pargs = nullptr;
    unsigned long long *best = nullptr;
    hipEvent_t ev[8] = {nullptr};
    float ms[7] = {0, 0, 0, 0, 0, 0, 0};
    int eval_blocks = 0, scan_blocks = 0;
};

namespace {
inline unsigned blocks(uint64_t n, unsigned per) { return (unsigned)((n + per - 1) / per); }
// in-place DIF NTT of `count` arrays of 2^bits words each: natural order in, bit-reversed order out
void ntt_dif(const h2w_prover *p, uint64_t *a, int bits, uint64_t count, hipStream_t s) {
    const uint64_t n = 1ull << bits; const int tb = bits < NTT_TILE_BITS ? bits : NTT_TILE_BITS;
    for (int st = bits; st > tb; st--) hipLaunchKernelGGL(k_dif_stage, dim3(blocks(n / 2, 256), (unsigned)count), dim3(256), 0, s, a, n, st, p->tw, p->d.lde_bits - st);
    if (tb >= 1) hipLaunchKernelGGL(k_dif_local, dim3((unsigned)(n >> tb), (unsigned)count), dim3(256), 0, s, a, n, tb, p->tw, p->d.lde_bits);
}
// all levels above the leaves up to the cap (level bits - cap_height) of `count` trees tree_stride hashes apart
void build_trees(const h2w_prover *p, H4 *base, int bits, uint64_t count, uint64_t tree_stride, hipStream_t s) {
    for (int l = 1; l <= bits - p->shape.cap_height; l++) {
        const uint64_t m = 1ull << (bits - l);
        hipLaunchKernelGGL(k_tree_level, dim3(blocks(m, 64), (unsigned)count), dim3(64), 0, s, p->dk, p->shape.hash_mode, base + level_off(bits, l - 1), base + level_off(bits, l), m, tree_stride);
    }
}
void free_batch_buffers(h2w_prover *p) {
    hipFree(p->lde); hipFree(p->tree); hipFree(p->Fa); hipFree(p->Fb); hipFree(p->T); hipFree(p->totals); hipFree(p->carry); hipFree(p->partial);
    hipFree(p->pts); hipFree(p->qargs); hipFree(p->pargs); hipFree(p->x_index); hipFree(p->best);
    for (int st = 0; st < MAX_STEPS; st++) { hipFree(p->fv[st]); hipFree(p->ftree[st]); p->fv[st] = nullptr; p->ftree[st] = nullptr; }
    p->lde = nullptr; p->tree = nullptr; p->Fa = p->Fb = p->T = p->totals = p->carry = p->partial = p->pts = nullptr; p->qargs = nullptr; p->pargs = nullptr;
    p->x_index = nullptr; p->best = nullptr; p->cap = 0;
}
// device buffers of a batch of n proofs: [proof][...] everywhere
int ensure_capacity(h2w_prover *p, uint64_t n) {
    if (n <= p->cap) return 0;
    free_batch_buffers(p);
    const Derived &d = p->d; const uint64_t L = p->L, N = p->N;
    bool ok = true;
    auto dm = [&](void **q, size_t bytes) { if (ok && hipMalloc(q, bytes ? bytes : 8) != hipSuccess) ok = false; };
    dm((void **)&p->lde, n * p->np * L * 8);
    dm((void **)&p->tree, n * d.n_oracles * 2 * L * sizeof(H4));
    dm((void **)&p->Fa, n * L * sizeof(gle_t)); dm((void **)&p->Fb, n * L * sizeof(gle_t)); dm((void **)&p->T, n * N * sizeof(gle_t));
    dm((void **)&p->totals, n * p->scan_blocks * sizeof(gle_t)); dm((void **)&p->carry, n * p->scan_blocks * sizeof(gle_t));
    dm((void **)&p->partial, n * 2 * p->np * p->eval_blocks * sizeof(gle_t));
    dm((void **)&p->pts, n * 3 * sizeof(gle_t)); dm((void **)&p->qargs, n * sizeof(QuotArgs)); dm((void **)&p->pargs, n * sizeof(PowArgs));
    { uint64_t m = L; for (int st = 0; st < d.n_steps; st++) { dm((void **)&p->fv[st], n * 2 * m * 8); dm((void **)&p->ftree[st], n * 2 * (m >> d.arity[st]) * sizeof(H4)); m >>= d.arity[st]; } }
    dm((void **)&p->x_index, n * (size_t)(p->shape.num_queries ? p->shape.num_queries : 1) * 8); dm((void **)&p->best, n * 8);
    if (!ok) { free_batch_buffers(p); set_error("h2w_prove_fri_batch: out of device memory for this batch size"); (void)hipGetLastError(); return -1; }
    p->cap = n;
    return 0;
}
}  // namespace

extern "C" {

h2w_prover *h2w_prover_new(const h2w_shape_t *shape, const h2w_poseidon_consts_t *consts, int device_id) {
    if (!shape || !consts) { set_error("h2w_prover_new: null argument"); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_error("no HIP device: libh2w has no CPU fallback"); return nullptr; }
    const Derived d = derive_shape(*shape);
    if (d.lde_bits > 24 || d.lde_bits < 1 || shape->cap_height > d.lde_bits || shape->n_pis > 64 || shape->n_pis < 0 || shape->num_queries > 4096 || shape->num_queries < 0 ||
        shape->n_cols + shape->n_perm_z + shape->n_quotient > MAX_BATCH_POLYS || shape->arity_bits > 4 || shape->arity_bits < 1 || d.final_poly_len > MAX_FINAL_POLY * 16) {
        set_error("h2w_prover_new: unsupported shape"); return nullptr;
    }
    h2w_prover *p = new h2w_prover();
    p->shape = *shape; p->d = d; p->pl = proof_layout(*shape, d); p->device = device_id;
    p->N = 1ull << shape->degree_bits; p->L = 1ull << d.lde_bits;
    for (int o = 0; o < d.n_oracles; o++) { p->poly0[o] = p->np; p->np += d.oracle_polys[o]; }
    p->scan_blocks = (int)blocks(p->N, SCAN_TILE); p->eval_blocks = (int)blocks(p->N, 256 * EVAL_PER_THREAD);
    // constants: caller's tables + PoseidonBN254 tables in Montgomery form
    const FrParams P = fr_params_init();
    p->hk.k = *consts; p->hk.r2 = P.r2; p->hk.ninv = P.ninv;
    for (int i = 0; i < 88; i++) p->hk.bn.c[i] = fr_mont_mul(consts->bn_c[i], P.r2, P.ninv);
    for (int i = 0; i < 392; i++) p->hk.bn.s[i] = fr_mont_mul(consts->bn_s[i], P.r2, P.ninv);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { p->hk.bn.m[i][j] = fr_mont_mul(consts->bn_m[i][j], P.r2, P.ninv); p->hk.bn.p[i][j] = fr_mont_mul(consts->bn_p[i][j], P.r2, P.ninv); }
    bool ok = hipSetDevice(device_id) == hipSuccess;
    if (ok) ok = hipMalloc((void **)&p->dk, sizeof(ProverConsts)) == hipSuccess;
    if (ok) ok = hipMalloc((void **)&p->tw, (p->L / 2 ? p->L / 2 : 1) * 8) == hipSuccess;
    for (int i = 0; i < 8 && ok; i++) ok = hipEventCreate(&p->ev[i]) == hipSuccess;
    if (ok) ok = hipMemcpy(p->dk, &p->hk, sizeof(ProverConsts), hipMemcpyHostToDevice) == hipSuccess;
    if (ok && p->L >= 2) { hipLaunchKernelGGL(k_twiddles, dim3(blocks(p->L / 2, 256)), dim3(256), 0, 0, p->tw, p->L / 2, gl_primitive_root_of_unity(d.lde_bits)); ok = hipDeviceSynchronize() == hipSuccess; }
    if (!ok) { set_error(std::string("h2w_prover_new: ") + hipGetErrorString(hipGetLastError())); h2w_prover_free(p); return nullptr; }
    return p;
}
void h2w_prover_free(h2w_prover *p) {
    if (!p) return;
    free_batch_buffers(p);
    hipFree(p->dk); hipFree(p->tw);
    for (int i = 0; i < 8; i++) if (p->ev[i]) hipEventDestroy(p->ev[i]);
    delete p;
}
uint64_t h2w_prover_num_polys(const h2w_prover *p) { return p ? (uint64_t)p->np : 0; }
uint64_t h2w_prover_proof_words(const h2w_prover *p) { return p ? p->pl.total : 0; }

int h2w_prove_fri_batch(h2w_prover *p, const uint64_t *coeffs_dev, const uint64_t *public_inputs, uint64_t *proofs_dev, uint64_t n, void *stream_) {
    if (!p || !coeffs_dev || !proofs_dev || (p->shape.n_pis > 0 && !public_inputs)) { set_error("h2w_prove_fri_batch: null argument"); return -1; }
    if (n == 0) return 0;
    if (n > 2048) { set_error("h2w_prove_fri_batch: at most 2048 proofs per call"); return -1; }
    const auto t_begin = std::chrono::steady_clock::now();
    hipStream_t s = (hipStream_t)stream_;
    const h2w_shape_t &sh = p->shape; const Derived &d = p->d; const ProofLayout &pl = p->pl;
    const int mode = sh.hash_mode, lb = d.lde_bits, capb = sh.cap_height, cs = d.cap_size, no = d.n_oracles, np = p->np;
    const uint64_t N = p->N, L = p->L; const unsigned nb = (unsigned)n;
    H2W_HIP(hipSetDevice(p->device));
    if (ensure_capacity(p, n)) return -1;
    std::vector<uint64_t> head(n * pl.queries, 0);             // per proof: the words before the per-query blocks
    // ---- LDE: lde[proof][poly][j] = f(7 w^bitrev(j))
    H2W_HIP(hipEventRecord(p->ev[0], s));
    hipLaunchKernelGGL(k_scale_pad, dim3(blocks(L, 256), (unsigned)(n * np)), dim3(256), 0, s, coeffs_dev, N, (uint64_t)7, L, p->lde);
    ntt_dif(p, p->lde, lb, n * np, s);
    H2W_HIP(hipEventRecord(p->ev[1], s));
    // ---- Merkle commitments of the oracles: tree[proof][oracle]
    const uint64_t tstride = 2 * L;
    for (int o = 0; o < no; o++)
        hipLaunchKernelGGL(k_leaves_initial, dim3(blocks(L, 64), nb), dim3(64), 0, s, p->dk, mode, p->lde + (uint64_t)p->poly0[o] * L, d.oracle_polys[o], L, p->tree + o * tstride, (uint64_t)np * L, no * tstride);
    build_trees(p, p->tree, lb, n * no, tstride, s);
    std::vector<H4> caps(n * no * cs);
    H2W_HIP(hipMemcpy2DAsync(caps.data(), (size_t)cs * sizeof(H4), p->tree + level_off(lb, lb - capb), tstride * sizeof(H4), (size_t)cs * sizeof(H4), n * no, hipMemcpyDeviceToHost, s));
    H2W_HIP(hipEventRecord(p->ev[2], s));
    H2W_HIP(hipStreamSynchronize(s));
    const int o_trace = 0, o_perm = sh.n_perm_z > 0 ? 1 : -1, o_quot = no - 1;
    // ---- transcript up to zeta (challenger/mod.rs:168-222 in stark/mod.rs order)
    std::vector<HostChallenger> chs(n, HostChallenger(&p->hk.k));
    std::vector<gle_t> pts(3 * n);                              // [zeta of every proof | g zeta of every proof | beta of every proof]
    for (uint64_t b = 0; b < n; b++) {
        HostChallenger &ch = chs[b]; const H4 *cp = &caps[b * no * cs]; uint64_t *hd = &head[b * pl.queries];
        memcpy(hd + pl.trace_cap, cp + (size_t)o_trace * cs, (size_t)cs * 32);
        memcpy(hd + pl.quotient_cap, cp + (size_t)o_quot * cs, (size_t)cs * 32);
        if (o_perm >= 0) memcpy(hd + pl.perm_cap, cp + (size_t)o_perm * cs, (size_t)cs * 32);
        for (int i = 0; i < cs; i++) ch.observe_hash(mode, cp[(size_t)o_trace * cs + i]);
        if (o_perm >= 0) {
            for (int set = 0; set < sh.perm_batch_size; set++) for (int i = 0; i < sh.num_challenges; i++) { ch.challenge(); ch.challenge(); }
            for (int i = 0; i < cs; i++) ch.observe_hash(mode, cp[(size_t)o_perm * cs + i]);
        }
        for (int i = 0; i < sh.num_challenges; i++) ch.challenge();
        for (int i = 0; i < cs; i++) ch.observe_hash(mode, cp[(size_t)o_quot * cs + i]);
        pts[b] = ch.ext_challenge(); pts[n + b] = gle_mul(gle_of(gl_primitive_root_of_unity(sh.degree_bits)), pts[b]);
    }
    // ---- openings: batch 0 = every polynomial at zeta, batch 1 = trace and permutation Zs at g zeta
    const int nz = np, nzn = np - d.oracle_polys[o_quot], eb = p->eval_blocks; const uint64_t pstride = (uint64_t)(nz + nzn) * eb;
    H2W_HIP(hipMemcpyAsync(p->pts, pts.data(), 2 * n * sizeof(gle_t), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_eval_partial, dim3((unsigned)eb, (unsigned)nz, nb), dim3(256), 0, s, coeffs_dev, N, p->pts, p->partial, (uint64_t)np * N, pstride);
    if (nzn > 0) hipLaunchKernelGGL(k_eval_partial, dim3((unsigned)eb, (unsigned)nzn, nb), dim3(256), 0, s, coeffs_dev, N, p->pts + n, p->partial + (size_t)nz * eb, (uint64_t)np * N, pstride);
    std::vector<gle_t> part(n * pstride);
    H2W_HIP(hipMemcpyAsync(part.data(), p->partial, part.size() * sizeof(gle_t), hipMemcpyDeviceToHost, s));
    H2W_HIP(hipStreamSynchronize(s));
    std::vector<gle_t> opens(n * 2 * MAX_BATCH_POLYS), alphas(n);
    for (uint64_t b = 0; b < n; b++) {
        HostChallenger &ch = chs[b]; gle_t *open0 = &opens[b * 2 * MAX_BATCH_POLYS], *open1 = open0 + MAX_BATCH_POLYS;
        for (int i = 0; i < nz + nzn; i++) { gle_t a = gle_of(0); for (int e = 0; e < eb; e++) a = gle_add(a, part[b * pstride + (size_t)i * eb + e]); if (i < nz) open0[i] = a; else open1[i - nz] = a; }
        for (int i = 0; i < nz; i++) ch.observe_ext(open0[i]);
        for (int i = 0; i < nzn; i++) ch.observe_ext(open1[i]);
        alphas[b] = ch.ext_challenge();
        // local_values, next_values, permutation_zs, permutation_zs_next, quotient_polys (witness/mod.rs:236-266 order)
        uint64_t *w = &head[b * pl.queries + pl.openings]; const int nc = sh.n_cols, npz = sh.n_perm_z, nq = sh.n_quotient;
        for (int i = 0; i < nc; i++) { *w++ = open0[i].c[0]; *w++ = open0[i].c[1]; }
        for (int i = 0; i < nc; i++) { *w++ = open1[i].c[0]; *w++ = open1[i].c[1]; }
        for (int i = 0; i < npz; i++) { *w++ = open0[nc + i].c[0]; *w++ = open0[nc + i].c[1]; }
        for (int i = 0; i < npz; i++) { *w++ = open1[nc + i].c[0]; *w++ = open1[nc + i].c[1]; }
        for (int i = 0; i < nq; i++) { *w++ = open0[nc + npz + i].c[0]; *w++ = open0[nc + npz + i].c[1]; }
    }
    // ---- F = (G_0 - G_0(zeta)) / (X - zeta) * alpha^{n_1} + (G_1 - G_1(g zeta)) / (X - g zeta)
    gle_t *F = p->Fa, *Fo = p->Fb;
    H2W_HIP(hipMemsetAsync(F, 0, n * L * sizeof(gle_t), s));
    std::vector<QuotArgs> qa(n);
    for (int bt = 0; bt < 2; bt++) {
        const int nbp = bt == 0 ? nz : nzn; if (nbp == 0) continue;
        for (uint64_t b = 0; b < n; b++) {
            QuotArgs &A = qa[b]; const gle_t *open = &opens[b * 2 * MAX_BATCH_POLYS + (size_t)bt * MAX_BATCH_POLYS];
            A.coef = coeffs_dev + b * np * N; A.N = N; A.npoly = nbp; A.z = pts[bt * n + b]; A.zinv = gle_inv(A.z);
            gle_t ap = gle_of(1); A.gz = gle_of(0);
            for (int i = 0; i < nbp; i++) { A.ap[i] = ap; A.gz = gle_add(A.gz, gle_mul(ap, open[i])); ap = gle_mul(ap, alphas[b]); }
            A.an = ap;
        }
        H2W_HIP(hipMemcpyAsync(p->qargs, qa.data(), n * sizeof(QuotArgs), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_quot_terms, dim3((unsigned)p->scan_blocks, nb), dim3(256), 0, s, p->qargs, p->T, p->totals);
        hipLaunchKernelGGL(k_suffix_totals, dim3(nb), dim3(64), 0, s, p->totals, p->carry, p->scan_blocks);
        hipLaunchKernelGGL(k_quot_finish, dim3((unsigned)p->scan_blocks, nb), dim3(256), 0, s, p->qargs, p->T, p->carry, F, L);
        H2W_HIP(hipStreamSynchronize(s));                      // qa is reused by the second batch
    }
    H2W_HIP(hipEventRecord(p->ev[3], s));
    // ---- commit phase
    uint64_t shift = 7; int cur = lb;
    std::vector<H4> fcap(n * cs);
    for (int st = 0; st < d.n_steps; st++) {
        const int ab = d.arity[st]; const uint64_t m = 1ull << cur, nl = m >> ab;
        hipLaunchKernelGGL(k_split_scale, dim3(blocks(m, 256), nb), dim3(256), 0, s, F, m, shift, p->fv[st], L);
        ntt_dif(p, p->fv[st], cur, 2 * n, s);
        hipLaunchKernelGGL(k_leaves_fri, dim3(blocks(nl, 64), nb), dim3(64), 0, s, p->dk, mode, p->fv[st], ab, m, p->ftree[st], 2 * nl);
        build_trees(p, p->ftree[st], cur - ab, n, 2 * nl, s);
        H2W_HIP(hipMemcpy2DAsync(fcap.data(), (size_t)cs * sizeof(H4), p->ftree[st] + level_off(cur - ab, cur - ab - capb), 2 * nl * sizeof(H4), (size_t)cs * sizeof(H4), n, hipMemcpyDeviceToHost, s));
        H2W_HIP(hipStreamSynchronize(s));
        for (uint64_t b = 0; b < n; b++) {
            memcpy(&head[b * pl.queries + pl.commit_caps + (uint64_t)st * cs * 4], &fcap[b * cs], (size_t)cs * 32);
            for (int i = 0; i < cs; i++) chs[b].observe_hash(mode, fcap[b * cs + i]);
            pts[2 * n + b] = chs[b].ext_challenge();
        }
        H2W_HIP(hipMemcpyAsync(p->pts + 2 * n, &pts[2 * n], n * sizeof(gle_t), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_fold, dim3(blocks(nl, 256), nb), dim3(256), 0, s, F, Fo, nl, ab, p->pts + 2 * n, L);
        H2W_HIP(hipStreamSynchronize(s));                      // the betas are overwritten by the next step
        gle_t *t = F; F = Fo; Fo = t;
        shift = gl_exp(shift, 1ull << ab); cur -= ab;
    }
    const int fl = d.final_poly_len;
    std::vector<gle_t> fin(n * fl);
    H2W_HIP(hipMemcpy2DAsync(fin.data(), (size_t)fl * sizeof(gle_t), F, L * sizeof(gle_t), (size_t)fl * sizeof(gle_t), n, hipMemcpyDeviceToHost, s));
    H2W_HIP(hipEventRecord(p->ev[4], s));
    H2W_HIP(hipStreamSynchronize(s));
    // ---- proof of work: absorb the complete rate blocks of the pending inputs once, search the witnesses on the device
    std::vector<PowArgs> pa(n); std::vector<unsigned long long> wit(n, ~0ull);
    for (uint64_t b = 0; b < n; b++) {
        HostChallenger &ch = chs[b]; uint64_t *hd = &head[b * pl.queries];
        for (int i = 0; i < fl; i++) { hd[pl.final_poly + 2 * i] = fin[b * fl + i].c[0]; hd[pl.final_poly + 2 * i + 1] = fin[b * fl + i].c[1]; ch.observe_ext(fin[b * fl + i]); }
        PowArgs &PA = pa[b]; PA.pow_bits = sh.pow_bits; PA.ntail = (int)(ch.in.size() % SPONGE_RATE);
        uint64_t st[12]; memcpy(st, ch.state, sizeof(st));
        const size_t full = ch.in.size() - (size_t)PA.ntail;
        for (size_t off = 0; off < full; off += SPONGE_RATE) { memcpy(st, ch.in.data() + off, SPONGE_RATE * 8); gl_permute(&p->hk.k, st); }
        memcpy(PA.state, st, sizeof(st));
        for (int i = 0; i < SPONGE_RATE; i++) PA.tail[i] = i < PA.ntail ? ch.in[full + i] : 0;
    }
    H2W_HIP(hipMemcpyAsync(p->pargs, pa.data(), n * sizeof(PowArgs), hipMemcpyHostToDevice, s));
    H2W_HIP(hipMemcpyAsync(p->best, wit.data(), n * 8, hipMemcpyHostToDevice, s));
    // 2^pow_bits candidates per round (63 % of the proofs finish in each; finished proofs' blocks exit at once)
    const int round_bits = sh.pow_bits < 8 ? 8 : sh.pow_bits > 22 ? 22 : sh.pow_bits;
    for (uint64_t first = 0;; first += 1ull << round_bits) {
        if (first >> 40) { set_error("h2w_prove_fri_batch: no proof-of-work witness found"); return -1; }
        hipLaunchKernelGGL(k_pow, dim3(1u << (round_bits - 6), nb), dim3(64), 0, s, p->dk, p->pargs, first, p->best);
        H2W_HIP(hipMemcpyAsync(wit.data(), p->best, n * 8, hipMemcpyDeviceToHost, s));
        H2W_HIP(hipStreamSynchronize(s));
        bool all = true; for (uint64_t b = 0; b < n; b++) all = all && wit[b] != ~0ull;
        if (all) break;
    }
    H2W_HIP(hipEventRecord(p->ev[5], s));
    // ---- query openings
    const int nq = sh.num_queries;
    std::vector<uint64_t> xs(n * (size_t)(nq ? nq : 1));
    for (uint64_t b = 0; b < n; b++) {
        HostChallenger &ch = chs[b];
        head[b * pl.queries + pl.pow_witness] = wit[b];
        ch.observe(wit[b]); (void)ch.challenge();
        for (int q = 0; q < nq; q++) xs[b * nq + q] = ch.challenge() & (L - 1);
    }
    H2W_HIP(hipMemcpyAsync(p->x_index, xs.data(), xs.size() * 8, hipMemcpyHostToDevice, s));
    GatherArgs G; G.lde = p->lde; G.L = L; G.lde_stride = (uint64_t)np * L; G.lb = lb; G.capb = capb; G.no = no; G.tree = p->tree; G.tree_stride = tstride;
    for (int o = 0; o < 3; o++) { G.npoly[o] = o < no ? d.oracle_polys[o] : 0; G.poly0[o] = p->poly0[o]; G.init_off[o] = pl.init_off[o]; }
    G.n_steps = d.n_steps;
    { uint64_t m = L; for (int st = 0; st < MAX_STEPS; st++) { G.ab[st] = st < d.n_steps ? d.arity[st] : 0; G.fv[st] = p->fv[st]; G.fn[st] = m; G.ftree[st] = p->ftree[st]; G.step_off[st] = pl.step_off[st]; if (st < d.n_steps) m >>= d.arity[st]; } }
    G.proof = proofs_dev; G.proof_stride = pl.total; G.q_base = pl.queries; G.q_words = pl.query_words; G.x_index = p->x_index;
    if (nq > 0) hipLaunchKernelGGL(k_gather, dim3((unsigned)nq, nb), dim3(64), 0, s, G);
    H2W_HIP(hipMemcpy2DAsync(proofs_dev, pl.total * 8, head.data(), pl.queries * 8, pl.queries * 8, n, hipMemcpyHostToDevice, s));
    if (sh.n_pis > 0) H2W_HIP(hipMemcpy2DAsync(proofs_dev + pl.pis, pl.total * 8, public_inputs, (size_t)sh.n_pis * 8, (size_t)sh.n_pis * 8, n, hipMemcpyHostToDevice, s));
    H2W_HIP(hipEventRecord(p->ev[6], s));
    H2W_HIP(hipStreamSynchronize(s));
    H2W_HIP(hipGetLastError());
    for (int i = 0; i < 6; i++) { float t = 0; H2W_HIP(hipEventElapsedTime(&t, p->ev[i], p->ev[i + 1])); p->ms[i] = t; }
    p->ms[6] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return 0;
}
int h2w_prove_fri(h2w_prover *p, const uint64_t *coeffs_dev, const uint64_t *public_inputs, uint64_t *proof_dev, void *stream) {
    return h2w_prove_fri_batch(p, coeffs_dev, public_inputs, proof_dev, 1, stream);
}
int h2w_prover_timing(h2w_prover *p, float ms[7]) {
    if (!p || !ms) { set_error("h2w_prover_timing: null argument"); return -1; }
    for (int i = 0; i < 7; i++) ms[i] = p->ms[i];
    return 0;
}

}  // extern "C"
