// keygen.h — keygen-side bookkeeping of the eager context: what halo2-base's Context records when
// witness_gen_only is false (semantics [R]): the selector of every vertical gate, the copy constraints between an
// `Existing` cell and the cell it is copied into (advice_equalities), the constant equalities of `Constant` cells, and the cells
// registered for the range lookup.  Purely structural: driven by cell offsets, it mirrors Context::assign_region and the
// GateChip / RangeChip primitives region by region (the cell VALUES are produced by the records / expansion kernel as before).
// The eager ops call it with the offsets carried by their h2w_assigned_t handles; `row` must track the context's cell count.
#pragma once
#include <vector>
#include <cstdint>
#include "field.h"

namespace h2w {

struct MetaRecorder {
    struct Spec { int kind; int64_t src; fr_t c; };            // 0 witness, 1 existing(src cell), 2 constant(c)
    static Spec W() { return Spec{0, -1, fr_zero()}; }
    static Spec EX(int64_t cell) { return Spec{1, cell, fr_zero()}; }
    static Spec K(const fr_t &v) { return Spec{2, -1, v}; }
    static Spec Ku(uint64_t v) { return K(fr_from_u64(v)); }

    int L = 21; uint64_t row = 0;
    std::vector<uint64_t> sel, lookups; std::vector<uint64_t> eq;                  // eq: pairs (a, b) flattened
    std::vector<uint64_t> ceq_cell; std::vector<fr_t> ceq_val;

    void equal(int64_t a, int64_t b) { eq.push_back((uint64_t)a); eq.push_back((uint64_t)b); }
    // Context::assign_region: push cells, record copy / constant equalities, enable gates (offsets relative to the region)
    uint64_t region(const Spec *cells, int n, const int *gates, int ng) {
        const uint64_t r0 = row;
        for (int i = 0; i < n; i++) {
            if (cells[i].kind == 1 && cells[i].src >= 0) equal((int64_t)(r0 + i), cells[i].src);
            else if (cells[i].kind == 2) { ceq_cell.push_back(r0 + i); ceq_val.push_back(cells[i].c); }
        }
        for (int g = 0; g < ng; g++) sel.push_back(r0 + (uint64_t)gates[g]);
        row += (uint64_t)n; return r0;
    }
    int64_t load_witness() { Spec s = W(); return (int64_t)region(&s, 1, nullptr, 0); }
    int64_t load_constant(const fr_t &v) { Spec s = K(v); return (int64_t)region(&s, 1, nullptr, 0); }
    int64_t gate4(Spec a, Spec b, Spec c, Spec d) { const int g = 0; Spec s[4] = {a, b, c, d}; return (int64_t)region(s, 4, &g, 1) + 3; }
    int64_t add(Spec a, Spec b) { return gate4(a, b, Ku(1), W()); }                       // [a, b, 1, a+b]
    int64_t sub(Spec a, Spec b) { return gate4(W(), b, Ku(1), a) - 3; }                   // [a-b, b, 1, a] -> cell -4
    int64_t mul(Spec a, Spec b) { return gate4(Ku(0), a, b, W()); }                       // [0, a, b, a*b]
    int64_t mul_add(Spec a, Spec b, Spec c) { return gate4(c, a, b, W()); }               // [c, a, b, a*b+c]
    int64_t select(Spec a, Spec b, Spec s) {
        const int g[2] = {0, 4}; Spec c[8] = {W(), Ku(1), b, a, b, s, W(), W()};
        const uint64_t r0 = region(c, 8, g, 2); equal((int64_t)r0, (int64_t)r0 + 6); equal((int64_t)r0 + 2, (int64_t)r0 + 4); return (int64_t)r0 + 7;
    }
    int64_t is_zero(Spec a, bool idx_variant) {                                            // [z, a, inv, 1, 0, a, z, 0] -> cell -2
        const int g[2] = {0, 4}; Spec c[8] = {W(), a, W(), Ku(1), Ku(0), a, W(), Ku(0)};
        const uint64_t r0 = region(c, 8, g, 2); equal((int64_t)r0, (int64_t)r0 + 6); if (idx_variant) equal((int64_t)r0 + 1, (int64_t)r0 + 5);
        return (int64_t)r0 + 6;
    }
    void idx_to_indicator(int64_t idx_cell, size_t len, int64_t *out) {
        Spec idx = EX(idx_cell);
        for (size_t i = 0; i < len; i++) {
            if (i == 0) { out[0] = is_zero(idx, true); idx = EX((int64_t)row - 3); }
            else { const int64_t d = sub(idx, Ku((uint64_t)i)); out[i] = is_zero(EX(d), false); }
        }
    }
    int64_t select_by_indicator(const int64_t *a, size_t stride, const int64_t *ind, size_t len) {   // [0, a0, ind0, s0, ...] gates @3i
        std::vector<Spec> c(1 + 3 * len); std::vector<int> g(len); c[0] = Ku(0);
        for (size_t i = 0; i < len; i++) { c[1 + 3 * i] = EX(a[i * stride]); c[2 + 3 * i] = EX(ind[i]); c[3 + 3 * i] = W(); g[i] = (int)(3 * i); }
        return (int64_t)region(c.data(), (int)c.size(), g.data(), (int)len) + (int64_t)c.size() - 1;
    }
    // inner_product(a, constants b): short form when b[0] == 1; returns the accumulator cell; first = region start
    int64_t inner_product_const(const Spec *a, const fr_t *b, size_t n, uint64_t *first = nullptr) {
        std::vector<Spec> c; std::vector<int> g; size_t start = 0;
        if (n > 0 && fr_eq(b[0], fr_from_u64(1))) { c.push_back(a[0]); start = 1; } else c.push_back(Ku(0));
        for (size_t i = start; i < n; i++) { g.push_back((int)c.size() - 1); c.push_back(a[i]); c.push_back(K(b[i])); c.push_back(W()); }
        const uint64_t r0 = region(c.data(), (int)c.size(), g.data(), (int)g.size());
        if (first) *first = r0;
        return (int64_t)r0 + (int64_t)c.size() - 1;
    }
    void assert_bit(int64_t x) { const int g = 0; Spec c[4] = {Ku(0), EX(x), EX(x), EX(x)}; region(c, 4, &g, 1); }
    void num_to_bits(int64_t a, size_t nbits, int64_t *out) {
        std::vector<Spec> q(nbits, W()); std::vector<fr_t> b(nbits); for (size_t i = 0; i < nbits; i++) b[i] = fr_pow2((int)i);
        uint64_t r0; const int64_t acc = inner_product_const(q.data(), b.data(), nbits, &r0); equal(a, acc);
        out[0] = (int64_t)r0; for (size_t i = 1; i < nbits; i++) out[i] = (int64_t)r0 + 1 + 3 * ((int64_t)i - 1);
        for (size_t i = 0; i < nbits; i++) assert_bit(out[i]);
    }
    int64_t bits_or_limbs_to_num(const int64_t *in, size_t n, int limb_bits) {
        std::vector<Spec> q(n); std::vector<fr_t> b(n); for (size_t i = 0; i < n; i++) { q[i] = EX(in[i]); b[i] = fr_pow2((int)(i * (size_t)limb_bits)); }
        return inner_product_const(q.data(), b.data(), n);
    }
    void range_check(int64_t a, size_t bits) {
        if (bits == 0) { ceq_cell.push_back((uint64_t)a); ceq_val.push_back(fr_zero()); return; }
        const size_t n = (bits + (size_t)L - 1) / (size_t)L, rem = bits % (size_t)L; int64_t last = a;
        if (n == 1) lookups.push_back((uint64_t)a);
        else {
            std::vector<Spec> q(n, W()); std::vector<fr_t> b(n); for (size_t i = 0; i < n; i++) b[i] = fr_pow2((int)(i * (size_t)L));
            uint64_t r0; const int64_t acc = inner_product_const(q.data(), b.data(), n, &r0); equal(a, acc);
            lookups.push_back(r0); for (size_t i = 0; i + 1 < n; i++) lookups.push_back(r0 + 1 + 3 * i);
            last = (int64_t)r0 + 1 + 3 * ((int64_t)n - 2);
        }
        if (rem == 1) assert_bit(last);
        else if (rem > 1) lookups.push_back((uint64_t)mul(EX(last), K(fr_pow2((int)((size_t)L - rem)))));
    }
    void check_less_than(Spec a, Spec b, size_t bits) {     // [a+2^n-b, b, 1, a+2^n, -2^n, 1, a] g@0,@3 ; range_check(cell -7, n)
        const int g[2] = {0, 3}; Spec c[7] = {W(), b, Ku(1), W(), K(fr_neg(fr_pow2((int)bits))), Ku(1), a};
        const uint64_t r0 = region(c, 7, g, 2); range_check((int64_t)r0, bits);
    }
    static int bit_length(uint64_t b) { int n = 0; while (b) { n++; b >>= 1; } return n; }
    void check_less_than_safe(int64_t a, uint64_t b) {
        const size_t rb = (size_t)((bit_length(b) + L - 1) / L * L);
        range_check(a, rb); check_less_than(EX(a), Ku(b), rb);
    }
    void decompose_le(int64_t a, size_t limb_bits, size_t n, int64_t *out) {
        std::vector<Spec> q(n, W()); std::vector<fr_t> b(n); for (size_t i = 0; i < n; i++) b[i] = fr_pow2((int)(i * limb_bits));
        uint64_t r0; const int64_t acc = inner_product_const(q.data(), b.data(), n, &r0); equal(a, acc);
        out[0] = (int64_t)r0; for (size_t i = 0; i + 1 < n; i++) out[i + 1] = (int64_t)r0 + 1 + 3 * (int64_t)i;
        for (size_t i = 0; i < n; i++) range_check(out[i], limb_bits);
    }
    // ---- GoldilocksChip (field/goldilocks/base.rs)
    static constexpr uint64_t P = 0xFFFFFFFF00000001ULL;
    int64_t gl_load_witness() { const int64_t w = load_witness(); check_less_than_safe(w, P); return w; }      // :107-119
    int64_t gl_reduce(int64_t a) {                                                                               // :346-368 -> remainder cell
        const int64_t q = gl_load_witness(), r = gl_load_witness(), p = load_constant(fr_from_u64(P));
        const int64_t rhs = mul_add(EX(q), EX(p), EX(r)); equal(a, rhs); return r;
    }
};

}  // namespace h2w
