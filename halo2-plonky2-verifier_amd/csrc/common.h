// common.h — shared host-side plumbing of libh2w (error handling, device tables, kernel launch prototypes).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "records.h"

namespace h2w {

void set_error(const std::string &s);
extern thread_local std::string g_last_error;

#define H2W_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    h2w::set_error(std::string(#expr) + ": " + hipGetErrorString(_e)); return -1; } } while (0)

// Makes `device` the calling thread's current HIP device for the lifetime of the guard (a caller that drives several GPUs from
// one thread must not have to bracket every call with hipSetDevice); -1 = leave as is.  device_of: the device a pointer lives on.
struct DeviceGuard {
    int prev = -1; bool switched = false;
    explicit DeviceGuard(int device) {
        if (device < 0) return;
        if (hipGetDevice(&prev) == hipSuccess && prev != device) switched = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete; DeviceGuard &operator=(const DeviceGuard &) = delete;
};
inline int device_of(const void *ptr) {
    hipPointerAttribute_t a;
    if (!ptr || hipPointerGetAttributes(&a, ptr) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return a.type == hipMemoryTypeDevice ? a.device : -1;
}

// keygen lists of a plan's cell stream that need wires with cell identities (copy constraints, constant equalities): built once, on
// demand, by abi_backend.cpp (the eager keygen context replays the shape) and cached in the plan
struct PlanEqualities { bool ready = false; std::vector<uint64_t> pairs, const_cells; std::vector<fr_t> const_values; std::vector<int64_t> const_word; };   // const_word[i] >= 0: the constant IS that proof word
}  // namespace h2w
struct h2w_plan;
namespace h2w {
PlanEqualities &plan_equalities(h2w_plan *);                       // batch.hip
const h2w_shape_t &plan_shape(const h2w_plan *);
const h2w_poseidon_consts_t &plan_consts(const h2w_plan *);
uint64_t plan_cells(const h2w_plan *);

// arguments of the expansion kernel (expand.hip)
struct ExpandArgs {
    const uint64_t *meta;      // [nrec] template id << 56 | first cell offset (per proof, shared by the batch)
    const rec_t *recs;         // [nproofs][rec_stride]
    uint64_t nrec, rec_stride;
    fr_t *out;                 // [nproofs][cell_stride]
    uint64_t cell_stride;
    const fr_t *pool;          // literal pool (T_LITERAL), may be null
    const uint32_t *slots; uint32_t nslots;
    const tmpl_info_t *info; uint32_t ntmpl;
    const fr_t *consts; uint32_t nconsts;
    int rb, lookup_bits;
    uint32_t max_cells;        // largest template (cells per record) of the plan: sizes expand_kernel_h's chunk table
    uint32_t *tile_ctr;        // [nproofs] zeroed work counters: tiles are handed out dynamically (null: static striding)
    uint32_t ntiles = 0, q_tiles = 1;   // expand_fast: work units per proof / per query block (set by launch_expand)
    uint32_t nproofs, roam;    // expand_fast: roam != 0 -> a 1-D grid of resident blocks whose wavefronts move on to the next proof with tiles left (set by launch_expand)
    uint32_t roam_per_cu = 0;  // caller: 0 = one grid column of blocks per proof; k > 0 = roaming wavefronts on (at most) k blocks per CU.  2 fills the
                               // chip (Goldilocks caps: nothing else needs LDS); 1 leaves half of every CU's LDS to the PoseidonBN254 chain kernels of the launches in flight
    ColMap cm;                 // column-major emission (starts == nullptr: flat)
    // (proof, query) sharding (shard_world <= 1: none): the expansion kernel walks only the record ranges of the blocks this rank owns - the
    // prologue block of proof p (records [0, pro_nrec)) if p % world == rank, query block q (records [q_rec0(q), + q_nrec(q))) if
    // (p * nq + q) % world == rank; shard_compact: the blocks go to the rank's packed buffer (batchargs.h ShardMap)
    uint32_t nq, shard_rank, shard_world, shard_compact;
    uint64_t pro_nrec, q_rec0_first, q_rec0_rest, q_nrec_first, q_nrec_rest, pro_ncell, q_cell0_first, q_cell0_rest, q_ncell_rest, q_slot;
};
int launch_expand(const ExpandArgs &A, uint64_t nproofs, int grid_x, hipStream_t stream);
inline void expand_unsharded(ExpandArgs &A) {      // one block per proof: all of its records
    A.nq = 1; A.shard_rank = 0; A.shard_world = 1; A.shard_compact = 0; A.pro_nrec = A.nrec; A.q_rec0_first = A.q_rec0_rest = ~0ull; A.q_nrec_first = A.q_nrec_rest = 1;
    A.pro_ncell = A.q_cell0_first = A.q_cell0_rest = A.q_ncell_rest = A.q_slot = 0;
}

constexpr int MAX_SLOTS = 1536;   // limits of a plan's template table (the default expansion kernel sizes its LDS copy dynamically)
constexpr int MAX_CONSTS = 96;

// device copies of a TemplateTable
struct DeviceTables {
    uint32_t *slots = nullptr; tmpl_info_t *info = nullptr; fr_t *consts = nullptr;
    uint32_t nslots = 0, ntmpl = 0, nconsts = 0, max_cells = 0; int rb = 0, L = 0;
    int upload(const TemplateTable &tt) {
        free();
        rb = tt.rb; L = tt.L;
        max_cells = 0; for (const tmpl_info_t &ti : tt.info) if (ti.ncells > max_cells) max_cells = ti.ncells;
        if (tt.slots.size() > MAX_SLOTS || tt.consts.size() > MAX_CONSTS || tt.info.size() > T_MAX) { set_error("template table too large"); return -1; }
        nslots = (uint32_t)tt.slots.size(); ntmpl = (uint32_t)tt.info.size(); nconsts = (uint32_t)tt.consts.size();
        H2W_HIP(hipMalloc((void **)&slots, nslots * sizeof(uint32_t)));
        H2W_HIP(hipMalloc((void **)&info, ntmpl * sizeof(tmpl_info_t)));
        H2W_HIP(hipMalloc((void **)&consts, (nconsts ? nconsts : 1) * sizeof(fr_t)));
        H2W_HIP(hipMemcpy(slots, tt.slots.data(), nslots * sizeof(uint32_t), hipMemcpyHostToDevice));
        H2W_HIP(hipMemcpy(info, tt.info.data(), ntmpl * sizeof(tmpl_info_t), hipMemcpyHostToDevice));
        if (nconsts) H2W_HIP(hipMemcpy(consts, tt.consts.data(), nconsts * sizeof(fr_t), hipMemcpyHostToDevice));
        return 0;
    }
    void fill(ExpandArgs &A) const { A.slots = slots; A.nslots = nslots; A.info = info; A.ntmpl = ntmpl; A.consts = consts; A.nconsts = nconsts; A.max_cells = max_cells; A.rb = rb; A.lookup_bits = L; }
    void free() {
        if (slots) hipFree(slots); if (info) hipFree(info); if (consts) hipFree(consts);
        slots = nullptr; info = nullptr; consts = nullptr;
    }
};

}  // namespace h2w
