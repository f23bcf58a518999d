// records.h — the compact "cell block record" format shared by every producer (eager C-ABI on the host,
// value kernels on the device) and the one consumer (the HBM-bound expansion kernel, expand.hip).
//
// A record is 32 bytes {a,b,c,d} of u64 operands plus a static 8-byte meta word
// (template id << 56 | first cell offset).  A template is a list of per-cell "slots"; every slot is either a
// 256-bit constant or a bit-field ((base >> shift) & mask(width)) << lshift of one of 11 per-record 128-bit
// bases, so expansion is branch-free: one lane per output cell, coalesced 32-byte stores.
//
// Cell templates restate halo2-base and GoldilocksChip::reduce
// (verifier/src/field/goldilocks/base.rs:346-368; worked layout in SURVEY.md Appendix C.1).
#pragma once
#include <stdint.h>
#include <vector>
#include <map>
#include "field.h"

namespace h2w {

struct rec_t { uint64_t a, b, c, d; };
HD void g_store_rec(rec_t *p, uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
    unsigned long long *q = reinterpret_cast<unsigned long long *>(p);
    H2W_GSTORE64(q, a); H2W_GSTORE64(q + 1, b); H2W_GSTORE64(q + 2, c); H2W_GSTORE64(q + 3, d);
}

// Column-major emission (fused): the advice is written straight into the FlexGate column layout.  Column c holds the
// flat cells [starts[c], starts[c+1]] (the boundary cell is shared: it is the last row of column c and row 0 of column c+1);
// a cell i lives at (c << k) + (i - starts[c]) for the LAST column c with starts[c] <= i; a fix-up kernel repeats the boundary cells
// in the previous column and zero-fills the unused rows.  starts == nullptr: flat layout.
struct ColMap { const uint64_t *starts; uint32_t ncols, k; };
struct ColRange { uint64_t lo, hi, delta; };
// rare (once per column a strand enters): kept out of line - and OUT OF THE CURSOR: as a member function it took the cursor's address, which put
// the cursor of every column-layout kernel on the stack (each of its reads a flat load behind the record stores)
static HOL ColRange col_locate(const uint64_t *starts, uint32_t ncols, uint32_t k, uint64_t cell) {
    uint32_t a = 0, b = ncols;                   // largest c with starts[c] <= cell
    while (b - a > 1) { const uint32_t mid = (a + b) >> 1; if (g_load_u64(starts + mid) <= cell) a = mid; else b = mid; }
    ColRange r; r.lo = g_load_u64(starts + a); r.hi = a + 1 < ncols ? g_load_u64(starts + a + 1) : ~0ull;
    r.delta = ((uint64_t)a << k) - r.lo;
    return r;
}
struct ColCursor {
    ColMap m; uint64_t lo, hi; uint64_t delta;      // cells in [lo, hi) map to cell + delta (mod 2^64)
    HD void init(const ColMap &cm) { m = cm; lo = 0; hi = cm.starts ? 0 : ~0ull; delta = 0; }
    HD void locate(uint64_t cell) { const ColRange r = col_locate(m.starts, m.ncols, m.k, cell); lo = r.lo; hi = r.hi; delta = r.delta; }
    HD uint64_t map(uint64_t cell) { if (cell < lo || cell >= hi) locate(cell); return cell + delta; }
};

// compile-time switch of the sinks / kernels: the flat instantiation carries no cursor at all
template <bool COLS> struct ColPolicy;
template <> struct ColPolicy<false> { HD void init(const ColMap &) {} HD uint64_t map(uint64_t c) const { return c; } };
template <> struct ColPolicy<true> : ColCursor {};

// base indices
enum { B_A = 0, B_B, B_C, B_D, B_V, B_X0, B_X0P, B_X0PP, B_X1, B_X1P, B_X1PP,
       B_W,          // X0 * p + X1: what reduce's closing mul_add gate outputs (= V unless the hinted quotient wrapped mod p, i.e. V div p >= p)
       B_COUNT };
// base modes
enum { M_GLOP = 0,   // V = A*B + C ; (X0, X1) = (V div p, V mod p)
       M_LOADW = 1,  // (X0, X1) = (A, B) ; V = A*B + C
       M_WIDEV = 2   // V = A + B*2^64 ; (X0, X1) = (V div p mod p, V mod p)   (standalone reduce of an unreduced wire)
};

// fixed template ids (ids >= T_DYNAMIC are registered on demand, e.g. range_check(bits))
enum {
    T_CONST1 = 0,   // [A]
    T_CONST4,       // [A, B, C, D]
    T_REP12,        // 12 x [A]
    T_GATE,         // [C, A, B, V]                                  gate.add / mul / mul_add on u64 operands
    T_KB_GATE,      // [B] [C, A, B, V]                              GoldilocksChip::sub_no_reduce (base.rs:262-272)
    T_REDUCE,       // reduce tail of V = A + B*2^64                 GoldilocksChip::reduce (base.rs:346-368)
    T_GLOP,         // [C, A, B, V] + reduce tail                    add / mul / mul_add (base.rs:251-329)
    T_KA_GLOP,      // [A] + GLOP                                    load_constant(K) then add/mul/mul_add with K
    T_KB_GLOP,      // [B] + GLOP                                    sub (base.rs:274-283)
    T_LOADW,        // [A] + check_less_than_safe(A, p)              GoldilocksChip::load_witness (base.rs:107-119)
    T_LOADW2,       // LOADW(A) LOADW(B)                             ext load_witness (extension.rs:85-97)
    T_CLT_SAFE,     // check_less_than_safe(A, p) only
    T_LITERAL,      // B cells copied verbatim from the literal pool starting at index A
    T_DYNAMIC
};
constexpr int T_MAX = 64;

HD uint32_t slot_const(uint32_t idx) { return 0x80000000u | idx; }
HD uint32_t slot_field(uint32_t src, uint32_t shift, uint32_t width, uint32_t lshift = 0) {
    return src | (shift << 4) | (width << 11) | (lshift << 19);
}

struct tmpl_info_t { uint32_t slot_base; uint16_t ncells; uint8_t mode; uint8_t pad; };

HD uint64_t meta_pack(uint32_t tmpl, uint64_t cell_off) { return ((uint64_t)tmpl << 56) | cell_off; }
HD uint32_t meta_tmpl(uint64_t m) { return (uint32_t)(m >> 56); }
HD uint64_t meta_off(uint64_t m) { return m & 0x00FFFFFFFFFFFFFFULL; }

// Host-side template table for a given lookup_bits.
// per-cell keygen flags (host only): a vertical gate starts at this cell (halo2-base selector,
// SURVEY App. A) / the cell is registered for the range lookup (RangeChip::range_check limbs)
enum { CF_GATE = 1, CF_LOOKUP = 2 };

struct TemplateTable {
    int L = 21, rb = 84, nlimb = 4;
    std::vector<uint32_t> slots;
    std::vector<uint8_t> slot_flags;   // parallel to `slots`
    std::vector<uint8_t> fl;           // flags of the template under construction (cell index -> CF_*)
    void flag(const std::vector<uint32_t> &s, int back, uint8_t f) { const size_t i = s.size() - 1 - (size_t)back; if (fl.size() <= i) fl.resize(i + 1, 0); fl[i] |= f; }
    std::vector<tmpl_info_t> info;
    std::vector<fr_t> consts;
    std::map<int, int> rc_ids;  // range_check(bits) -> template id

    int add_const(const fr_t &c) {
        for (size_t i = 0; i < consts.size(); i++) if (fr_eq(consts[i], c)) return (int)i;
        consts.push_back(c); return (int)consts.size() - 1;
    }
    uint32_t K(const fr_t &c) { return slot_const((uint32_t)add_const(c)); }
    uint32_t Ku(uint64_t c) { return K(fr_from_u64(c)); }
    // range_check(Z, bits) cells: inner product of limbs (n>1) + last-limb fix-up
    void emit_rc(std::vector<uint32_t> &s, int src, int bits) {
        int n = (bits + L - 1) / L, rem = bits % L;
        if (n > 1) {                                       // inner_product(limbs, 2^(jL)): gates at cells 0, 3, 6, ...; every limb looked up
            s.push_back(slot_field(src, 0, L)); flag(s, 0, CF_GATE | CF_LOOKUP);
            for (int j = 1; j < n; j++) {
                s.push_back(slot_field(src, j * L, L)); flag(s, 0, CF_LOOKUP);
                s.push_back(K(fr_pow2(j * L)));
                s.push_back(slot_field(src, 0, (j + 1) * L > 128 ? 128 : (j + 1) * L)); if (j < n - 1) flag(s, 0, CF_GATE);
            }
        }
        int last_shift = (n - 1) * L;
        if (rem == 1) { s.push_back(Ku(0)); flag(s, 0, CF_GATE); for (int k = 0; k < 3; k++) s.push_back(slot_field(src, last_shift, L)); }     // assert_bit
        else if (rem > 1) {                                // mul(last, 2^(L-rem)), the product looked up
            s.push_back(Ku(0)); flag(s, 0, CF_GATE); s.push_back(slot_field(src, last_shift, L));
            s.push_back(K(fr_pow2(L - rem))); s.push_back(slot_field(src, last_shift, L, L - rem)); flag(s, 0, CF_LOOKUP);
        }
    }
    // check_less_than_safe(X, p): range_check(X, rb); [X', p, 1, X'', -2^rb, 1, X]; range_check(X', rb)
    void emit_clt(std::vector<uint32_t> &s, int x, int xp, int xpp) {
        emit_rc(s, x, rb);
        s.push_back(slot_field(xp, 0, 128)); flag(s, 0, CF_GATE); s.push_back(Ku(GL_P)); s.push_back(Ku(1));       // gates @0, @3
        s.push_back(slot_field(xpp, 0, 128)); flag(s, 0, CF_GATE); s.push_back(K(fr_neg(fr_pow2(rb)))); s.push_back(Ku(1));
        s.push_back(slot_field(x, 0, 128));
        emit_rc(s, xp, rb);
    }
    void emit_loadw(std::vector<uint32_t> &s, int x, int xp, int xpp) { s.push_back(slot_field(x, 0, 128)); emit_clt(s, x, xp, xpp); }
    void emit_gate(std::vector<uint32_t> &s) {
        s.push_back(slot_field(B_C, 0, 128)); flag(s, 0, CF_GATE); s.push_back(slot_field(B_A, 0, 128));
        s.push_back(slot_field(B_B, 0, 128)); s.push_back(slot_field(B_V, 0, 128));
    }
    void emit_tail(std::vector<uint32_t> &s) {  // base.rs:356-364
        emit_loadw(s, B_X0, B_X0P, B_X0PP);      // quotient
        emit_loadw(s, B_X1, B_X1P, B_X1PP);      // remainder
        s.push_back(Ku(GL_P));                   // load_constant(ORDER)
        s.push_back(slot_field(B_X1, 0, 128)); flag(s, 0, CF_GATE); s.push_back(slot_field(B_X0, 0, 128)); s.push_back(Ku(GL_P)); s.push_back(slot_field(B_W, 0, 128));
    }
    int finish(int id, std::vector<uint32_t> &s, int mode) {
        if ((int)info.size() <= id) info.resize(id + 1, tmpl_info_t{0, 0, 0, 0});
        info[id].slot_base = (uint32_t)slots.size(); info[id].ncells = (uint16_t)s.size(); info[id].mode = (uint8_t)mode;
        slots.insert(slots.end(), s.begin(), s.end());
        fl.resize(s.size(), 0); slot_flags.insert(slot_flags.end(), fl.begin(), fl.end()); fl.clear();
        return id;
    }
    explicit TemplateTable(int lookup_bits) {
        L = lookup_bits; nlimb = (64 + L - 1) / L; rb = nlimb * L;
        std::vector<uint32_t> s;
        s = {slot_field(B_A, 0, 128)}; finish(T_CONST1, s, M_LOADW);
        s = {slot_field(B_A, 0, 128), slot_field(B_B, 0, 128), slot_field(B_C, 0, 128), slot_field(B_D, 0, 128)}; finish(T_CONST4, s, M_LOADW);
        s.assign(12, slot_field(B_A, 0, 128)); finish(T_REP12, s, M_LOADW);
        s.clear(); emit_gate(s); finish(T_GATE, s, M_LOADW);
        s.clear(); s.push_back(slot_field(B_B, 0, 128)); emit_gate(s); finish(T_KB_GATE, s, M_LOADW);
        s.clear(); emit_tail(s); finish(T_REDUCE, s, M_WIDEV);
        s.clear(); emit_gate(s); emit_tail(s); finish(T_GLOP, s, M_GLOP);
        s.clear(); s.push_back(slot_field(B_A, 0, 128)); emit_gate(s); emit_tail(s); finish(T_KA_GLOP, s, M_GLOP);
        s.clear(); s.push_back(slot_field(B_B, 0, 128)); emit_gate(s); emit_tail(s); finish(T_KB_GLOP, s, M_GLOP);
        s.clear(); emit_loadw(s, B_X0, B_X0P, B_X0PP); finish(T_LOADW, s, M_LOADW);
        s.clear(); emit_loadw(s, B_X0, B_X0P, B_X0PP); emit_loadw(s, B_X1, B_X1P, B_X1PP); finish(T_LOADW2, s, M_LOADW);
        s.clear(); emit_clt(s, B_X0, B_X0P, B_X0PP); finish(T_CLT_SAFE, s, M_LOADW);
        s.clear(); finish(T_LITERAL, s, M_LOADW);
    }
    // range_check(A, bits) as a template (eager API, PoW check fri/mod.rs:130-145)
    int rc_template(int bits) {
        auto it = rc_ids.find(bits);
        if (it != rc_ids.end()) return it->second;
        int id = (int)info.size(); if (id < T_DYNAMIC) id = T_DYNAMIC;
        if (id >= T_MAX) return -1;
        std::vector<uint32_t> s; emit_rc(s, B_X0, bits);
        finish(id, s, M_LOADW); rc_ids[bits] = id; return id;
    }
    int ncells(int t) const { return info[t].ncells; }
};

}  // namespace h2w
