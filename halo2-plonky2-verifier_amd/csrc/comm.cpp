// comm.cpp - the ONE collective of the path: the ingest rank's flat proof block goes to every rank, one process per GPU,
// over RCCL / xGMI; optionally the ranks exchange the 32-byte digests of what they wrote.  For callers without torch.distributed (the Rust
// side): bench.py's broadcast is the same ncclBroadcast through torch.  RCCL is resolved at the first h2w_comm_* call (dlopen, preferring a
// copy the process already holds, e.g. PyTorch's), so libh2w.so loads - and every other entry point works - where no RCCL is installed.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include "common.h"

using namespace h2w;

namespace {
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;      // optional: what RCCL itself says the communicator spans
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
Rccl &rccl() {
    static Rccl r; static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so"};
        for (const char *n : names) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);      // a copy the process already holds
        for (const char *n : names) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) return;
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
        r.Broadcast = (decltype(r.Broadcast))dlsym(r.lib, "ncclBroadcast");
        r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
        r.CommCount = (decltype(r.CommCount))dlsym(r.lib, "ncclCommCount");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
        r.ok = r.GetUniqueId && r.CommInitRank && r.Broadcast && r.AllGather && r.CommDestroy && r.GetErrorString;
    });
    return r;
}
bool need_rccl(const char *who) {
    if (rccl().ok) return true;
    set_error(std::string(who) + ": RCCL (librccl.so.1) could not be loaded");
    return false;
}
int nccl_fail(const char *who, ncclResult_t rc) { set_error(std::string(who) + ": " + rccl().GetErrorString(rc)); return -1; }
}  // namespace

struct h2w_comm { ncclComm_t comm = nullptr; int rank = 0, world = 1, device = -1; };

extern "C" {

int h2w_comm_unique_id(void *id128) {
    if (!id128) { set_error("h2w_comm_unique_id: null argument"); return -1; }
    static_assert(sizeof(ncclUniqueId) == H2W_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (!need_rccl("h2w_comm_unique_id")) return -1;
    ncclUniqueId id; const ncclResult_t rc = rccl().GetUniqueId(&id);
    if (rc != ncclSuccess) return nccl_fail("h2w_comm_unique_id", rc);
    std::memcpy(id128, &id, sizeof id);
    return 0;
}
h2w_comm *h2w_comm_init(const void *id128, int rank, int world, int device_id) {
    if (!id128 || world < 1 || rank < 0 || rank >= world) { set_error("h2w_comm_init: bad id / rank / world"); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("h2w_comm_init: no HIP device"); return nullptr; }
    if (device_id < 0 || device_id >= ndev) { set_error("h2w_comm_init: device_id out of range"); return nullptr; }
    if (!need_rccl("h2w_comm_init")) return nullptr;
    DeviceGuard dg(device_id);
    ncclUniqueId id; std::memcpy(&id, id128, sizeof id);
    h2w_comm *c = new h2w_comm; c->rank = rank; c->world = world; c->device = device_id;
    const ncclResult_t rc = rccl().CommInitRank(&c->comm, world, id, rank);
    if (rc != ncclSuccess) { nccl_fail("h2w_comm_init", rc); delete c; return nullptr; }
    return c;
}
void h2w_comm_free(h2w_comm *c) {
    if (!c) return;
    if (c->comm && rccl().ok) { DeviceGuard dg(c->device); (void)rccl().CommDestroy(c->comm); }
    delete c;
}
int h2w_comm_rank(const h2w_comm *c) { return c ? c->rank : -1; }
int h2w_comm_world(const h2w_comm *c) {      // the ranks RCCL counts in the communicator (ncclCommCount), not the number it was asked for
    if (!c) return 0;
    int n = c->world;
    if (c->comm && rccl().ok && rccl().CommCount && rccl().CommCount(c->comm, &n) != ncclSuccess) n = -1;
    return n;
}
int h2w_comm_broadcast_proofs(h2w_comm *c, uint64_t *proofs_dev, uint64_t n_words, int root, void *stream) {
    if (!c || !proofs_dev || root < 0 || root >= c->world) { set_error("h2w_comm_broadcast_proofs: bad argument"); return -1; }
    if (n_words == 0) return 0;
    DeviceGuard dg(c->device);
    const ncclResult_t rc = rccl().Broadcast(proofs_dev, proofs_dev, (size_t)n_words, ncclUint64, root, c->comm, (hipStream_t)stream);
    return rc == ncclSuccess ? 0 : nccl_fail("h2w_comm_broadcast_proofs", rc);
}
int h2w_comm_allgather_digests(h2w_comm *c, const uint64_t *digest4_dev, uint64_t *all_dev, void *stream) {
    if (!c || !digest4_dev || !all_dev) { set_error("h2w_comm_allgather_digests: null argument"); return -1; }
    DeviceGuard dg(c->device);
    const ncclResult_t rc = rccl().AllGather(digest4_dev, all_dev, 4, ncclUint64, c->comm, (hipStream_t)stream);
    return rc == ncclSuccess ? 0 : nccl_fail("h2w_comm_allgather_digests", rc);
}

}  // extern "C"
