// batchargs.h - kernel arguments of the batched hot path, shared by its translation units (batch.hip, glue.hip).
#pragma once
#include "valbackend.h"

namespace h2w {

typedef ValBackend<DevSink> DevB;
typedef ChallengeBlock<DevB> DevCB;   // (the wire types of every backend coincide: one ChallengeBlock layout)

struct BatchArgs {
    h2w_shape_t shape; const h2w_poseidon_consts_t *consts;
    const uint64_t *proofs; uint64_t proof_words;
    rec_t *recs; uint64_t rec_stride;
    fr_t *out; uint64_t cell_stride;
    DevCB *cbs; uint32_t *status;
    const uint16_t *ncells; const fr_t *inv_pos, *inv_neg;
    StrandTable st; FrParams P;
    int nproofs, role_base, dbg_skip_perm, dbg_prio;
    const fr_t *bn_tab;             // PoseidonBN254 tables of this plan: [2][BK_T] canonical / times R (coop.h bn_table_build)
    const LoadItem *load_items; uint32_t n_load_items; uint64_t load_nrec, load_ncell;
    ColMap cm;      // column-major emission (starts == nullptr: flat advice)
    int shard_rank, shard_world;      // (proof, query) units are dealt round-robin to shard_world ranks (1: everything)
};
__device__ __forceinline__ bool own_prologue(const BatchArgs &A, int p) { return A.shard_world <= 1 || p % A.shard_world == A.shard_rank; }      // rank proof_id mod world
__device__ __forceinline__ bool own_unit(const BatchArgs &A, int p, int q) { return A.shard_world <= 1 || (int)(((long long)p * A.shape.num_queries + q) % A.shard_world) == A.shard_rank; }

__device__ __forceinline__ ValCfg make_cfg(const BatchArgs &A, int p) {
    ValCfg c; c.proof = A.proofs + (uint64_t)p * A.proof_words; c.mode = A.shape.hash_mode; c.L = A.shape.lookup_bits; c.P = A.P;
    c.inv_pos = A.inv_pos; c.inv_neg = A.inv_neg; c.st = &A.st; c.split = true;
    c.load_items = A.load_items; c.n_load_items = A.n_load_items; c.load_nrec = A.load_nrec; c.load_ncell = A.load_ncell;
    c.split_bn = false;
    return c;
}

void launch_glue_strands(const BatchArgs &A, unsigned nlanes, hipStream_t stream);      // glue.hip

}  // namespace h2w
