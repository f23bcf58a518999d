// batchargs.h - kernel arguments of the batched hot path, shared by its translation units (batch.hip, glue.hip).
#pragma once
#include "valbackend.h"
#include "coop.h"
#include "rowfr.h"

namespace h2w {

typedef ValBackend<DevSink> DevB;
typedef ChallengeBlock<DevB> DevCB;   // (the wire types of every backend coincide: one ChallengeBlock layout)

// (proof, query) sharding of a launch (fri/mod.rs:488-501 is the loop being dealt out): unit u = proof * num_queries + query
// belongs to rank u % world, the prologue block of proof p to rank p % world.  A rank LAUNCHES only what it owns: its i-th unit is
// u = i * world + rank, its j-th proof p = j * world + rank (world 1: everything).
// compact != 0: the rank's advice buffer holds only its own blocks, back to back in (proof, block) order - its j-th owned prologue block
// and i-th owned query block start at local cell j * pro_ncell + i * q_slot (q_slot = the larger of the two query block sizes);
// compact == 0: every block at its global offset in advice[n_proofs][num_cells].
struct ShardMap {
    int rank, world, compact; uint32_t n_own_units, n_own_proofs; uint64_t q_slot;
    uint32_t unit_slot;      // PoseidonBN254 permutation units of one (proof, query) unit: the rank's i-th unit keeps its unit states / S-box values at slot i (a rank's workspace holds its own units only)
};

struct BatchArgs {
    h2w_shape_t shape; const h2w_poseidon_consts_t *consts;
    const uint64_t *proofs; uint64_t proof_words;
    rec_t *recs; uint64_t rec_stride;
    fr_t *out; uint64_t cell_stride;
    DevCB *cbs; uint32_t *status;
    const uint16_t *ncells; const fr_t *inv_pos, *inv_neg;
    const StrandTable *st; FrParams P;      // the shape's strand table, in device memory (a by-value copy in the kernel arguments is copied to every lane's stack as soon as it is indexed)
    int nproofs;
    const fr_t *bn_tab;             // PoseidonBN254 tables of this plan: [2][BK_T] canonical / times R (coop.h bn_table_build)
    const FriTab *fri;              // the shape's FRI-gadget constants (valbackend.h FriTab)
    const uint32_t *bn_tab9;        // the times-R entries and the BK_X block in limb form (coop.h bn_table9_build): the values pass
    fr_t *unit_state;               // [own units][unit_slot][4]: output state of every PoseidonBN254 permutation unit of this rank's (proof, query) units (values phase -> emission)
    fr_t *unit_sbox;                // [own units][unit_slot][56][3]: canonical x^2, x^4, x^5 of its partial rounds' S-boxes
    uint32_t *unit_sbox9;           // the same values as the row-cooperative values pass leaves them: times R, lazy, 12 dwords of 29-bit limbs each (rowperm.h; k_sbox_canon9 -> unit_sbox)
    const rf::RowConst *rowk;       // modulus, N' = -N^-1 mod 2^261 and R^2 in 29-bit limbs (rowfr.h), device memory of the plan
    uint64_t *glp_list;             // [nproofs][st.total_glp][GLP_LIST_WORDS]: the listed Goldilocks-Poseidon permutations (values phase -> record emission)
    int glp_small_mds;
    const LoadItem *load_items; uint32_t n_load_items, n_cap_items; uint64_t load_nrec, load_ncell; uint32_t *load_flag;      // load_items: n_load_items of the load phase, then n_cap_items cap hashes
    ColMap cm;      // column-major emission (starts == nullptr: flat advice)
    ShardMap sh;
};
__device__ __forceinline__ bool own_prologue(const BatchArgs &A, int p) { return A.sh.world <= 1 || p % A.sh.world == A.sh.rank; }
// the i-th (proof, query) unit of this rank; false past the last one
__device__ __forceinline__ bool own_unit_at(const BatchArgs &A, unsigned i, int &p, int &q) {
    if (i >= A.sh.n_own_units) return false;
    const unsigned long long u = (unsigned long long)i * (unsigned)A.sh.world + (unsigned)A.sh.rank;
    p = (int)(u / (unsigned)A.shape.num_queries); q = (int)(u % (unsigned)A.shape.num_queries);
    return true;
}
// Where the cells of proof p's block (q < 0: prologue, else query q) go: the pointer that the block's GLOBAL in-proof cell offsets are
// added to.  Flat layout: out + p * cell_stride.  Compact layout: shifted so that the block lands in the rank's packed buffer.
__device__ __forceinline__ fr_t *block_out(const BatchArgs &A, int p, int q) {
    if (!A.sh.compact) return A.out + (uint64_t)p * A.cell_stride;
    const uint64_t W = (uint64_t)A.sh.world, r = (uint64_t)A.sh.rank;
    const uint64_t u0 = (uint64_t)p * (uint64_t)A.shape.num_queries;
    const uint64_t pro_before = ((uint64_t)p + W - 1 - r) / W, units_before = (u0 + W - 1 - r) / W;      // owned prologues / units of the proofs before p
    uint64_t local = pro_before * A.st->pro_ncell + units_before * A.sh.q_slot, global = 0;
    if (q >= 0) {
        if ((uint64_t)p % W == r) local += A.st->pro_ncell;
        const uint64_t u = u0 + (uint64_t)q;
        local += ((u + W - 1 - r) / W - units_before) * A.sh.q_slot;      // owned units of this proof before query q
        global = strand_q_cell(*A.st, q);
    }
    return A.out + local - global;
}

__device__ __forceinline__ ValCfg make_cfg(const BatchArgs &A, int p) {
    ValCfg c; c.proof = A.proofs + (uint64_t)p * A.proof_words; c.mode = A.shape.hash_mode; c.L = A.shape.lookup_bits; c.P = A.P;
    c.inv_pos = A.inv_pos; c.inv_neg = A.inv_neg; c.st = A.st; c.split = true; c.fri = A.fri;
    c.load_items = A.load_items; c.n_load_items = A.n_load_items; c.n_cap_items = A.n_cap_items; c.load_nrec = A.load_nrec; c.load_ncell = A.load_ncell;
    c.split_bn = false;
    return c;
}

template <bool COLS, bool VALPH, int HM> __device__ __forceinline__ void coop_sink_init(CoopSinkT<COLS, VALPH, HM> &sink, const BatchArgs &A, int p, int q) {
    sink.recs = A.recs + (uint64_t)p * A.rec_stride; sink.out = block_out(A, p, q); sink.ncells = A.ncells; sink.lane = threadIdx.x; sink.cc.init(A.cm);
    sink.glp = A.glp_list + (uint64_t)p * A.st->total_glp * GLP_LIST_WORDS; sink.small_mds = A.glp_small_mds != 0;
    sink.bind_lds();
}

// The PoseidonBN254 emission kernel: two blocks of QUAD_BLOCK threads per CU by LDS (32.9 KB of tables + 10 KB of value slots per wavefront)
#ifndef H2W_QUAD_EU
#define H2W_QUAD_EU 2
#endif
#define H2W_QUAD_ATTR __attribute__((amdgpu_waves_per_eu(H2W_QUAD_EU, H2W_QUAD_EU)))

// PoseidonBN254 Merkle chains (hash_mode 1).  A quad's strand: (owned unit, kind); its cursor, index bits and unit buffer.
template <class QuadB, class Sink> __device__ __forceinline__ void quad_strand(const BatchArgs &A, Sink &sink, unsigned own_idx, int p, int q, int kind) {
    const int sq = q == 0 ? 0 : 1;
    sink.recs = A.recs + (uint64_t)p * A.rec_stride; sink.out = block_out(A, p, q); sink.ncells = A.ncells; sink.l4 = threadIdx.x & 3; sink.cc.init(A.cm);
    sink.nrec = strand_q_rec(*A.st, q) + A.st->mk_rec_rel[sq][kind]; sink.cell_off = strand_q_cell(*A.st, q) + A.st->mk_cell_rel[sq][kind];
    const uint64_t unit0 = (uint64_t)own_idx * A.sh.unit_slot + A.st->mk_unit_rel[sq][kind];
    sink.ustate = A.unit_state + unit0 * 4; sink.sbx = A.unit_sbox + unit0 * (BN_PARTIAL_ROUNDS * 3);
    ValCfg mc = make_cfg(A, p); mc.split_bn = true;
    QuadB be(sink, mc, !(q == 0 && kind == A.st->first_zero_kind));
    const h2w_shape_t shp = A.shape;      // (a reference into the kernel arguments would put all of them on every lane's stack)
    Verifier<QuadB> V(be, shp, A.consts);
    const uint64_t x = A.cbs[p].fri_query_indices[q];
    const int lde = V.d.lde_bits; int lo = 0;
    if (kind >= 3) for (int i = 0; i <= kind - 3; i++) lo += V.d.arity[i];
    const uint64_t cap_index = (x >> (lde - A.shape.cap_height)) & ((1ull << A.shape.cap_height) - 1);
    V.merkle_strand(q, kind, PackedBits{x, lo}, lde - lo, cap_index);
    if ((threadIdx.x & 3) == 0 && be.status) atomicCAS(&A.status[p], 0u, be.status);
}
void launch_glue_strands(const BatchArgs &A, hipStream_t stream);      // glue.hip
void launch_merkle_bn_emit(const BatchArgs &A, dim3 grid, hipStream_t stream);      // glue.hip
void launch_merkle_bn_values(const BatchArgs &A, dim3 grid, hipStream_t stream);    // glue.hip
void launch_merkle_bn_values_row(const BatchArgs &A, unsigned nkinds, hipStream_t stream);    // glue.hip
void launch_prologue_values(const BatchArgs &A, hipStream_t stream);   // glue.hip
void launch_merkle_gl_values(const BatchArgs &A, unsigned nunits, unsigned nkinds, hipStream_t stream);   // glue.hip

}  // namespace h2w
