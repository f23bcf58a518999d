// chipbatch.hip - batched chip-level ops on the device: n independent instances of ONE GoldilocksChip / GoldilocksQuadExtChip
// operation, each in a fresh Context - operands loaded as witnesses, then the op: the shape of the reference's own chip tests
// (field/goldilocks/base.rs:476-495 test_mul, extension.rs:473-493).  Same single-source chips (chips.h) as every other level,
// instantiated on the device value backend: one lane per instance computes the values and writes the block records, the
// expansion kernel materialises the cells.  Where the reference panics (GoldilocksChip::div by zero base.rs:379, extension
// inverse of zero extension.rs:327) the instance's status word is set (1 / 2) and the cells are those of the substituted
// operand, exactly like the batched verifier path (include/h2w.h "device status words").
#include <hip/hip_runtime.h>
#include <vector>
#include "common.h"
#include "valbackend.h"

namespace h2w {

struct CountSink {       // host: lays out the instance (record metas, cell count)
    static constexpr bool kCoop = false, kSplitOnly = false, kBnUnits = false, kDevSponge = false; static constexpr int kHashMode = -1;
    const TemplateTable *tt; std::vector<uint64_t> meta; uint64_t cell_off = 0;
    void rec(int t, uint64_t, uint64_t, uint64_t, uint64_t) { meta.push_back(meta_pack((uint32_t)t, cell_off)); cell_off += (uint64_t)tt->ncells(t); }
    void cell(const fr_t &) { cell_off++; }
    void gate() {}
    void lookup() {}
};

// operands: GL ops take 2 (add, sub, mul, div), 3 (mul_add) or 1 (inv) words; extension ops 2 words per element
HF int chip_op_operands(int op) {
    switch (op) {
        case H2W_OP_GL_ADD: case H2W_OP_GL_SUB: case H2W_OP_GL_MUL: case H2W_OP_GL_DIV: return 2;
        case H2W_OP_GL_MUL_ADD: return 3;
        case H2W_OP_GL_INV: return 1;
        case H2W_OP_EXT_MUL: case H2W_OP_EXT_DIV: return 4;
        case H2W_OP_EXT_INV: return 2;
        default: return -1;
    }
}
template <class B> HF void chip_program(B &be, int op, const uint64_t *w) {
    GoldilocksChip<B> gl(be); QuadExtChip<B> ex(be);
    if (op <= H2W_OP_GL_INV) {
        typename B::Gl a = gl.load_witness(w[0]);
        if (op == H2W_OP_GL_INV) { gl.inv(a); return; }
        typename B::Gl b = gl.load_witness(w[1]);
        if (op == H2W_OP_GL_ADD) gl.add(a, b);
        else if (op == H2W_OP_GL_SUB) gl.sub(a, b);
        else if (op == H2W_OP_GL_MUL) gl.mul(a, b);
        else if (op == H2W_OP_GL_DIV) gl.div(a, b);
        else { typename B::Gl c = gl.load_witness(w[2]); gl.mul_add(a, b, c); }
        return;
    }
    gle_t av; av.c[0] = w[0]; av.c[1] = w[1];
    ExtW<B> a = ex.load_witness(av);
    if (op == H2W_OP_EXT_INV) { ex.inv(a); return; }
    gle_t bv; bv.c[0] = w[2]; bv.c[1] = w[3];
    ExtW<B> b = ex.load_witness(bv);
    if (op == H2W_OP_EXT_MUL) ex.mul(a, b); else ex.div(a, b);
}

struct ChipArgs { int op, nw, L; FrParams P; const uint64_t *operands; rec_t *recs; uint64_t nrec, ncells, n; fr_t *out; const uint16_t *tmpl_cells; uint32_t *status; };
__global__ __launch_bounds__(64) void k_chip_batch(ChipArgs A) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n) return;
    DevSink sink; sink.recs = A.recs + i * A.nrec; sink.nrec = 0; sink.out = A.out + i * A.ncells; sink.cell_off = 0; sink.ncells = A.tmpl_cells; sink.cc.init(ColMap{nullptr, 0, 0});
    ValCfg cfg; cfg.proof = nullptr; cfg.mode = 0; cfg.L = A.L; cfg.P = A.P; cfg.inv_pos = cfg.inv_neg = nullptr; cfg.st = nullptr; cfg.split = false; cfg.split_bn = false;
    cfg.load_items = nullptr; cfg.n_load_items = 0; cfg.load_nrec = cfg.load_ncell = 0;
    ValBackend<DevSink> be(sink, cfg, true);
    uint64_t w[4];
    for (int k = 0; k < A.nw; k++) w[k] = g_load_u64(A.operands + i * (uint64_t)A.nw + k);
    chip_program(be, A.op, w);
    A.status[i] = be.status;
}

}  // namespace h2w

using namespace h2w;

struct h2w_chipbatch {
    int op, L, device, nw; TemplateTable tt; DeviceTables dt; FrParams P;
    uint64_t nrec = 0, ncells = 0; uint64_t *d_meta = nullptr; uint16_t *d_tmpl_cells = nullptr;
    explicit h2w_chipbatch(int L_) : tt(L_) {}
};

extern "C" {

h2w_chipbatch *h2w_chipbatch_new(int op, int lookup_bits, int device_id) {
    const int nw = chip_op_operands(op);
    if (nw < 0 || lookup_bits < 2 || lookup_bits > 28) { set_error("h2w_chipbatch_new: unknown op / lookup_bits outside [2, 28]"); return nullptr; }
    h2w_chipbatch *h = new h2w_chipbatch(lookup_bits);
    h->op = op; h->L = lookup_bits; h->device = device_id; h->nw = nw; h->P = fr_params_init();
    CountSink cs; cs.tt = &h->tt;
    {   // layout of one instance: replay on harmless operands (1: no division by zero)
        ValCfg cfg; cfg.proof = nullptr; cfg.mode = 0; cfg.L = lookup_bits; cfg.P = h->P; cfg.inv_pos = cfg.inv_neg = nullptr; cfg.st = nullptr; cfg.split = false; cfg.split_bn = false;
        cfg.load_items = nullptr; cfg.n_load_items = 0; cfg.load_nrec = cfg.load_ncell = 0;
        ValBackend<CountSink> be(cs, cfg, true);
        const uint64_t ones[4] = {1, 1, 1, 1};
        chip_program(be, op, ones);
    }
    h->nrec = cs.meta.size(); h->ncells = cs.cell_off;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { h->device = -1; return h; }      // layout queries still work; h2w_chipbatch_run fails
    if (device_id < 0 || device_id >= ndev) { set_error("h2w_chipbatch_new: device_id out of range"); delete h; return nullptr; }
    DeviceGuard dg(device_id);
    auto up = [&]() -> int {
        if (h->dt.upload(h->tt) != 0) return -1;
        H2W_HIP(hipMalloc((void **)&h->d_meta, h->nrec * sizeof(uint64_t)));
        H2W_HIP(hipMemcpy(h->d_meta, cs.meta.data(), h->nrec * sizeof(uint64_t), hipMemcpyHostToDevice));
        std::vector<uint16_t> nc(T_MAX, 0); for (size_t i = 0; i < h->tt.info.size(); i++) nc[i] = h->tt.info[i].ncells;
        H2W_HIP(hipMalloc((void **)&h->d_tmpl_cells, nc.size() * sizeof(uint16_t)));
        H2W_HIP(hipMemcpy(h->d_tmpl_cells, nc.data(), nc.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        return 0;
    };
    if (up() != 0) { h2w_chipbatch_free(h); return nullptr; }
    return h;
}
void h2w_chipbatch_free(h2w_chipbatch *h) {
    if (!h) return;
    DeviceGuard dg(h->device);
    if (h->d_meta) (void)hipFree(h->d_meta);
    if (h->d_tmpl_cells) (void)hipFree(h->d_tmpl_cells);
    h->dt.free();
    delete h;
}
uint64_t h2w_chipbatch_num_operands(const h2w_chipbatch *h) { return h ? (uint64_t)h->nw : 0; }
uint64_t h2w_chipbatch_num_cells(const h2w_chipbatch *h) { return h ? h->ncells : 0; }
int h2w_chipbatch_run(h2w_chipbatch *h, const uint64_t *operands_dev, uint64_t n, void *advice_dev, uint32_t *status_dev, void *stream_) {
    if (!h || !operands_dev || !advice_dev || !status_dev) { set_error("h2w_chipbatch_run: null argument"); return -1; }
    if (h->device < 0) { set_error("h2w_chipbatch_run: no HIP device - cells are only produced on the GPU (no CPU fallback)"); return -1; }
    if (n == 0) return 0;
    DeviceGuard dg(h->device);
    hipStream_t stream = (hipStream_t)stream_;
    const uint64_t CH = 32768;                                // instances per launch (grid.y of the expansion kernel)
    char *ws = nullptr; const size_t b_recs = (size_t)CH * h->nrec * sizeof(rec_t), b_ctr = (size_t)CH * sizeof(uint32_t);
    H2W_HIP(hipMallocAsync((void **)&ws, b_recs + b_ctr, stream));
    auto run = [&]() -> int {
        for (uint64_t first = 0; first < n; first += CH) {
            const uint64_t m = n - first < CH ? n - first : CH;
            ChipArgs A; A.op = h->op; A.nw = h->nw; A.L = h->L; A.P = h->P; A.operands = operands_dev + first * (uint64_t)h->nw; A.recs = (rec_t *)ws; A.nrec = h->nrec; A.ncells = h->ncells; A.n = m;
            A.out = (fr_t *)advice_dev + first * h->ncells; A.tmpl_cells = h->d_tmpl_cells; A.status = status_dev + first;
            hipLaunchKernelGGL(k_chip_batch, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, stream, A);
            ExpandArgs E;
            E.meta = h->d_meta; E.recs = A.recs; E.nrec = h->nrec; E.rec_stride = h->nrec; E.out = A.out; E.cell_stride = h->ncells; E.pool = nullptr;
            E.cm = ColMap{nullptr, 0, 0}; expand_unsharded(E);
            h->dt.fill(E);
            E.tile_ctr = (uint32_t *)(ws + b_recs);
            H2W_HIP(hipMemsetAsync(E.tile_ctr, 0, m * sizeof(uint32_t), stream));
            if (launch_expand(E, m, 1, stream) != 0) return -1;
            H2W_HIP(hipGetLastError());
        }
        return 0;
    };
    const int rc = run();
    (void)hipFreeAsync(ws, stream);
    return rc;
}

}  // extern "C"
