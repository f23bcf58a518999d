// Host restatement of the two reductions csrc/glperm.h writes in assembly (glq_reduce, glq_reduce96), instruction by instruction in 32 / 64-bit integer
// arithmetic, against x mod p computed with 128-bit integers: the ALGORITHM (one multiply-add folds the 2^64 word; its carry and the borrow of the 2^96
// word become one 64-bit correction; no second wrap) on edge values and random ones.  The device parity tests check the instructions themselves.
// g++ -O2 -std=c++17 tests/cpp/glq_reduce_check.cpp
#include <cstdint>
#include <cstdio>
typedef unsigned __int128 u128;
static const uint64_t P = 0xFFFFFFFF00000001ull, EPS = 0xFFFFFFFFull;

// lo + p2 2^64 + p3 2^96 -> a 64-bit representative (glperm.h glq_reduce)
static bool reduce(uint64_t lo, uint32_t p2, uint32_t p3, uint64_t &r) {
    const u128 m = (u128)p2 * EPS + lo;                       // v_mad_u64_u32 u, c, p2, -1, lo
    const uint64_t u = (uint64_t)m; const bool c = (m >> 64) != 0;
    const uint32_t v0 = (uint32_t)u - p3; const bool b0 = (uint32_t)u < p3;            // v_sub_co_u32
    const uint32_t u1 = (uint32_t)(u >> 32); const uint32_t v1 = u1 - (b0 ? 1u : 0u); const bool b = b0 && u1 == 0;   // v_subbrev_co_u32
    const uint32_t mc = c ? 0xFFFFFFFFu : 0u, mb = b ? 0xFFFFFFFFu : 0u;              // v_cndmask x 2
    const uint32_t kh = c ? 0u : mb, kl = mc - mb;                                    // v_cndmask, v_sub_u32
    const uint64_t v = ((uint64_t)v1 << 32) | v0, k = ((uint64_t)kh << 32) | kl;
    r = v + k;                                                                        // v_lshl_add_u64
    // the claim "r does not wrap": the true value (as an integer) of v + (c - b) eps lies in [0, 2^64)
    const __int128 t = (__int128)v + ((__int128)(c ? 1 : 0) - (b ? 1 : 0)) * (__int128)EPS;
    return t >= 0 && t < ((__int128)1 << 64) && (uint64_t)t == r;
}
// lo + h0 2^32 + h1 2^64 with lo + h1 eps < 2^64 (glperm.h glq_reduce96)
static bool reduce96(uint64_t lo, uint32_t h0, uint32_t h1, uint64_t &r) {
    const u128 m = (u128)h1 * EPS + lo; if (m >> 64) return false;                    // (the precondition)
    const uint64_t a = (uint64_t)m; const uint64_t s = (uint64_t)(uint32_t)(a >> 32) + h0;      // v_add_co_u32 on the high word
    const bool c = (s >> 32) != 0; const uint64_t x = ((uint64_t)(uint32_t)s << 32) | (uint32_t)a;
    r = x + (c ? EPS : 0);
    return !c || x + EPS >= x;                                                        // (no second wrap)
}
int main() {
    uint64_t seed = 0x243F6A8885A308D3ull; auto rnd = [&] { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return seed; };
    const uint64_t e64[] = {0, 1, 2, EPS - 1, EPS, EPS + 1, P - 1, P, P + 1, ~0ull, ~0ull - 1, 1ull << 32, (1ull << 32) - 2, 1ull << 63, 0xFFFFFFFF00000000ull, 0x00000000FFFFFFFEull};
    const uint32_t e32[] = {0, 1, 2, 0x7FFFFFFFu, 0x80000000u, 0xFFFFFFFEu, 0xFFFFFFFFu};
    long bad = 0, n = 0, borrows = 0, carries = 0;
    auto one = [&](uint64_t lo, uint32_t p2, uint32_t p3) {
        uint64_t r; const bool ok = reduce(lo, p2, p3, r);
        const u128 want = ((u128)lo + (u128)p2 * EPS % P + (u128)(P - p3 % P)) % P;      // 2^64 = eps, 2^96 = -1 (mod p)
        if (!ok || r % P != (uint64_t)want) { if (bad < 5) printf("reduce(%llx, %x, %x) = %llx\n", (unsigned long long)lo, p2, p3, (unsigned long long)r); bad++; }
        const u128 m = (u128)p2 * EPS + lo; carries += (m >> 64) != 0; borrows += ((uint64_t)m >> 32) == 0 && (uint32_t)m < p3; n++;
    };
    for (uint64_t lo : e64) for (uint32_t p2 : e32) for (uint32_t p3 : e32) one(lo, p2, p3);
    for (int i = 0; i < 2000000; i++) one(rnd(), (uint32_t)rnd(), (uint32_t)rnd());
    for (int i = 0; i < 200000; i++) one(rnd() & 0xFFFFFFFFull, 0, (uint32_t)rnd());               // the borrow: u below 2^32 and below p3
    for (int i = 0; i < 200000; i++) { const uint32_t p2 = (uint32_t)rnd(); one((uint64_t)(0 - (u128)p2 * EPS) + (rnd() & 0xFFFF), p2, (uint32_t)rnd()); }      // carry AND borrow: u just past 2^64
    for (int i = 0; i < 1000000; i++) {                                                          // products, as glq_mul forms them
        const uint64_t a = i & 1 ? rnd() : e64[rnd() % 16], b = i & 2 ? rnd() : e64[rnd() % 16]; const u128 pr = (u128)a * b;
        uint64_t r; if (!reduce((uint64_t)pr, (uint32_t)(pr >> 64), (uint32_t)(pr >> 96), r) || r % P != (uint64_t)(pr % P)) bad++;
        n++;
    }
    for (int i = 0; i < 1000000; i++) {                                                          // the sums glq_mds_small / glq_dense12 hand to glq_reduce96
        const uint64_t lo = rnd() >> (i % 3 ? 1 : 5), hi = rnd() >> (i % 3 ? 1 : 5);
        uint64_t r; const u128 want = ((u128)lo + ((u128)hi << 32)) % P;
        if ((u128)(uint32_t)(hi >> 32) * EPS + lo >> 64) continue;                                   // outside the precondition (sums below 2^63 are inside)
        if (!reduce96(lo, (uint32_t)hi, (uint32_t)(hi >> 32), r) || r % P != (uint64_t)want) bad++;
        n++;
    }
    if (borrows < 1000 || carries < 1000) { printf("the borrow / carry paths were not exercised (%ld, %ld)\n", borrows, carries); bad++; }
    printf(bad ? "FAILED (%ld of %ld)\n" : "OK %ld cases, %ld with a borrow, %ld with a carry\n", bad ? bad : n, bad ? n : borrows, carries);
    return bad != 0;
}
