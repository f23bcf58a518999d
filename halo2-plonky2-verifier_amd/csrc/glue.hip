// glue.hip - the two strand kernels that are serial programs of Goldilocks gadget operations:
//   k_strands          the query glue: FriChip::verify_query_round minus its Merkle proofs (fri/mod.rs:338-444: index bits, subgroup_x,
//                      combine_initial, per fold step the consistency select, compute_evaluation / interpolate_coset and x^arity,
//                      final-polynomial evaluation), one lane per (proof, query);
//   k_prologue_values  the prologue on values (stark/mod.rs:497-508, challenger/mod.rs:168-222, fri/mod.rs:45-62,130-145), one wavefront per proof.
//
// Their own translation unit because the gadget stack is compiled FLATTENED here (H2W_FLATTEN_CHIPS): a strand is a serial program of
// a few thousand Goldilocks ops, each of which appends one 32-byte block record.  Out of line, every op is an AMDGPU function call:
// it starts with s_waitcnt vmcnt(0) - i.e. it waits for the record store of the PREVIOUS op to be acknowledged by memory - and
// reaches the sink's cursor (record index, cell offset) through flat loads of the backend object (profiles/: 5.3 ms for 5.5 k ops
// per lane, ~1 us per op).  Flattened, the cursor lives in registers and nothing waits for a store.
#define H2W_FLATTEN_CHIPS 1
#include <hip/hip_runtime.h>
#include "common.h"
#include "batchargs.h"

namespace h2w {

// (the backends of this unit - DevSinkT<COLS, true> and CoopSinkT<COLS, true, -1> - are instantiated nowhere else: field.h, HNI)
template <bool COLS> __global__ __launch_bounds__(64) __attribute__((flatten)) void k_strands(BatchArgs A) {
    typedef DevSinkT<COLS, true> GlueSink; typedef ValBackend<GlueSink> GlueB;
    int p, q;
    if (!own_unit_at(A, blockIdx.x * blockDim.x + threadIdx.x, p, q)) return;      // lane i of the launch = this rank's i-th (proof, query) unit
    GlueSink sink; sink.recs = A.recs + (uint64_t)p * A.rec_stride; sink.out = block_out(A, p, q); sink.ncells = A.ncells; sink.cc.init(A.cm);
    sink.nrec = strand_q_rec(A.st, q); sink.cell_off = strand_q_cell(A.st, q);
    const ChallengeBlock<GlueB> &cb = *reinterpret_cast<const ChallengeBlock<GlueB> *>(&A.cbs[p]);
    GlueB be(sink, make_cfg(A, p), true);
    Verifier<GlueB> V(be, A.shape, A.consts);
    V.query_round(q, cb);
    if (be.status) atomicCAS(&A.status[p], 0u, be.status);
}

// one wavefront per proof: wave-uniform gadget code; every rank of a sharded run computes every prologue's VALUES (it needs the
// challenges); only the proof's owner writes the block (direct cells, records, permutation list)
template <bool COLS> __global__ __launch_bounds__(64) __attribute__((flatten)) void k_prologue_values(BatchArgs A) {
    typedef CoopSinkT<COLS, true> Sink; typedef ValBackend<Sink> CoopB;
    __builtin_amdgcn_s_setprio(3);   // latency-bound serial strand: win issue arbitration against co-resident streaming waves
    stage_glp_consts(A.consts, threadIdx.x, 64);
    const int p = blockIdx.x;
    Sink sink; coop_sink_init(sink, A, p, -1); sink.nrec = 0; sink.cell_off = 0; sink.glp_slot = 0;
    sink.emit = own_prologue(A, p);
    CoopB be(sink, make_cfg(A, p), true);
    Verifier<CoopB> V(be, A.shape, A.consts);
    V.prologue(*reinterpret_cast<ChallengeBlock<CoopB> *>(&A.cbs[p]));
    if (threadIdx.x == 0) A.status[p] = be.status;
}

void launch_prologue_values(const BatchArgs &A, hipStream_t stream) {
    if (A.cm.starts) hipLaunchKernelGGL(k_prologue_values<true>, dim3((unsigned)A.nproofs), dim3(64), 0, stream, A);
    else hipLaunchKernelGGL(k_prologue_values<false>, dim3((unsigned)A.nproofs), dim3(64), 0, stream, A);
}

void launch_glue_strands(const BatchArgs &A, hipStream_t stream) {
    const unsigned nlanes = A.sh.n_own_units;
    if (A.cm.starts) hipLaunchKernelGGL(k_strands<true>, dim3((nlanes + 63) / 64, 1), dim3(64), 0, stream, A);
    else hipLaunchKernelGGL(k_strands<false>, dim3((nlanes + 63) / 64, 1), dim3(64), 0, stream, A);
}

}  // namespace h2w
