// plan.h — the plan handle of API level 3 (include/h2w.h): a compiled shape (batch.hip h2w_plan_compile) or a traced run (replay.hip h2w_plan_from_trace).
#pragma once
#include "common.h"
#include "batchargs.h"

namespace h2w { struct TracedPlan; }
using namespace h2w;      // (included by batch.hip and replay.hip only, which live in that namespace's vocabulary)
struct h2w_plan {
    h2w_shape_t shape; int device;
    TemplateTable tt; DeviceTables dt; StrandTable st; FrParams P;
    Derived d; ProofLayout pl;
    uint64_t nrec = 0, ncells = 0, rec_cells = 0;
    LoadItem *d_items = nullptr; uint32_t n_items = 0, n_cap_items = 0; uint64_t load_nrec = 0, load_ncell = 0;      // d_items: the load phase's items, then the cap hashes'
    h2w_poseidon_consts_t h_consts;                       // host copy (keygen-metadata replay)
    bool meta_ready = false; std::vector<uint8_t> sel_bits, lk_bits; uint64_t n_gates = 0, n_lookups = 0; uint32_t *d_lookup_cells = nullptr; uint8_t *d_sel_bits = nullptr;
    uint64_t *d_col_tab = nullptr; std::vector<uint64_t> h_col_tab; int col_k = -1;     // column-major emission: [starts | lens] of the last break-point set
    PlanEqualities eqs;
    fr_t *d_bn_tab = nullptr; uint32_t *d_bn_tab9 = nullptr; FriTab *d_fri = nullptr; uint64_t nunit = 0; rf::RowConst *d_rowk = nullptr;     // PoseidonBN254 tables of this plan (coop.h bn_table_build)
    StrandTable *d_st = nullptr;                      // device copy of st
    bool small_mds = false;                           // Goldilocks-Poseidon MDS entries are tiny (coop.h glp_small_mds)
    uint64_t *d_meta = nullptr; h2w_poseidon_consts_t *d_consts = nullptr; uint16_t *d_ncells = nullptr; fr_t *d_inv = nullptr;
    static constexpr int EV_RING = 64, N_EV = 13, N_SIDE = 16;
    // per call: 0 start, 9 prologue values done, 11 / 12 permutation-record kernel start / end, 1 prologue block complete, 7 / 2 glue (+ Goldilocks Merkle
    // strands) start / done, 8 / 3 expansion start / done, 4 / 10 / 5 chain kernels start / values done / end, 6 end of call
    hipEvent_t evr[EV_RING][N_EV]; int passes_of[EV_RING] = {0};
    hipStream_t side[N_SIDE]; hipStream_t side_of[N_SIDE]; int n_side = 0;   // PoseidonBN254 chain kernels run beside the glue + expansion kernels
    int chain_passes = 0;            // H2W_OPT_CHAIN_PASSES (0: by the size of the launch)
    int values_form = 0;             // H2W_OPT_VALUES_FORM (0: by the size of the launch)
    int serial_expand = 1;           // H2W_OPT_SERIAL_EXPAND: the expansion kernel of a call waits for the previous call's
    bool fork_chains = true;         // of their own batch (they share only the prologue): one side stream per caller stream seen (created on demand)
    hipEvent_t *ev = evr[0]; uint64_t n_batches = 0; bool ev_ready = false, ev_recorded = false;
    h2w::TracedPlan *traced = nullptr;      // set: the plan replays a recorded tape (replay.hip); the strand tables above are unused
    explicit h2w_plan(int L) : tt(L) {}
};

