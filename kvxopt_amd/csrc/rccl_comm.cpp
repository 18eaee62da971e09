// RCCL bound directly (include/kvxhip.h, "kvx_rccl_*"): the collectives of the sharded factor without torch.distributed.
//
// Why: a process runs on the HIP runtime it loads first.  Once torch is imported that is the runtime inside the PyTorch wheel
// (7.0.51831), under which this library's one-enqueue launch graph cannot be used and every collective
// is a C -> Python callback that takes the GIL.  A rank process that talks to librccl.so itself stays on the system runtime
// (ROCm 7.2), enqueues its collectives from C in stream order and never creates a Python-side process group.
//
// librccl.so is opened with dlopen at the first kvx_rccl_* call: single-GPU users of libkvxhip.so do not need it.
// One communicator over all ranks plus one per distinct rank range of the factor's map (ncclCommSplit, the same list in the
// same order on every rank: kvx_chol_dist_groups).  The reference is single-process (src/C/cholmod.c:85): nothing to cite there.
#include "chol_internal.hpp"

#include <dlfcn.h>
#include <cstring>
#include <rccl/rccl.h>

#include <map>
#include <mutex>

using namespace kvx;

namespace {

struct Api {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t *, ncclConfig_t *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    std::string err;
};

Api &api()
{
    static Api A;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {getenv("KVX_RCCL_LIB"), "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so", "librccl.so.1", "librccl.so"};
        for (const char *nm : names) {
            if (!nm || !nm[0]) continue;
            A.h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (A.h) break;
        }
        if (!A.h) { A.err = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : ""); return; }
        auto sym = [&](const char *s) { void *p = dlsym(A.h, s); if (!p && A.err.empty()) A.err = std::string("librccl.so lacks ") + s; return p; };
        A.GetUniqueId = (decltype(A.GetUniqueId))sym("ncclGetUniqueId");
        A.CommInitRank = (decltype(A.CommInitRank))sym("ncclCommInitRank");
        A.CommSplit = (decltype(A.CommSplit))sym("ncclCommSplit");
        A.CommDestroy = (decltype(A.CommDestroy))sym("ncclCommDestroy");
        A.Broadcast = (decltype(A.Broadcast))sym("ncclBroadcast");
        A.AllReduce = (decltype(A.AllReduce))sym("ncclAllReduce");
        A.AllGather = (decltype(A.AllGather))sym("ncclAllGather");
        A.GetErrorString = (decltype(A.GetErrorString))sym("ncclGetErrorString");
        A.GetVersion = (decltype(A.GetVersion))sym("ncclGetVersion");
    });
    return A;
}

}  // namespace

struct kvx_rccl {
    int rank = 0, nranks = 1;
    ncclComm_t world = nullptr;
    std::map<std::pair<int, int>, ncclComm_t> sub;     // [lo, hi) -> communicator (rank lo is its rank 0); only ranges this rank is in
    double *scratch = nullptr;                         // device: small host-value collectives
    int64_t scratch_cap = 0;
    int64_t ncoll = 0, bytes = 0;
};

#define NCCLCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { \
    set_last_error(std::string("RCCL: ") + (api().GetErrorString ? api().GetErrorString(r_) : "error") + " (" #x ")"); return KVX_ECOMM; } } while (0)

static int need_api()
{
    Api &A = api();
    if (!A.h || !A.err.empty()) { set_last_error(A.err.empty() ? "librccl.so unavailable" : A.err); return KVX_ECOMM; }
    return KVX_OK;
}

extern "C" {

int kvx_rccl_version(int *version)
{
    return guarded([&] {
        int rc = need_api();
        if (rc) return rc;
        NCCLCHK(api().GetVersion(version));
        return (int)KVX_OK;
    });
}

int kvx_rccl_unique_id(char id[128])
{
    return guarded([&] {
        int rc = need_api();
        if (rc) return rc;
        static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
        ncclUniqueId u;
        NCCLCHK(api().GetUniqueId(&u));
        std::memcpy(id, &u, sizeof(u));
        return (int)KVX_OK;
    });
}

int kvx_rccl_init(int rank, int nranks, const char id[128], kvx_rccl **out)
{
    return guarded([&] {
        if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks) { set_last_error("kvx_rccl_init: bad arguments"); return (int)KVX_EINVAL; }
        int rc = need_api();
        if (rc) return rc;
        ncclUniqueId u;
        std::memcpy(&u, id, sizeof(u));
        kvx_rccl *W = new kvx_rccl();
        W->rank = rank; W->nranks = nranks;
        ncclResult_t r = api().CommInitRank(&W->world, nranks, u, rank);
        if (r != ncclSuccess) { set_last_error(std::string("RCCL: ncclCommInitRank: ") + api().GetErrorString(r)); delete W; return (int)KVX_ECOMM; }
        W->scratch_cap = 1024;
        if (hipMalloc((void **)&W->scratch, (size_t)W->scratch_cap * (size_t)std::max(nranks, 1) * sizeof(double)) != hipSuccess) {
            (void)hipGetLastError(); (void)api().CommDestroy(W->world); delete W; set_last_error("kvx_rccl_init: hipMalloc failed"); return (int)KVX_ENOMEM;
        }
        *out = W;
        return (int)KVX_OK;
    });
}

// Collective over ALL ranks of W: creates the communicator of the range [lo, hi) (ranks outside it take no colour).
int kvx_rccl_split(kvx_rccl *W, int lo, int hi)
{
    return guarded([&] {
        if (!W || lo < 0 || hi > W->nranks || hi - lo < 1) { set_last_error("kvx_rccl_split: bad range"); return (int)KVX_EINVAL; }
        if (lo == 0 && hi == W->nranks) return (int)KVX_OK;
        if (W->sub.count({lo, hi})) return (int)KVX_OK;
        const bool in = W->rank >= lo && W->rank < hi;
        ncclComm_t c = nullptr;
        NCCLCHK(api().CommSplit(W->world, in ? 1 : NCCL_SPLIT_NOCOLOR, W->rank - lo, &c, nullptr));
        if (in) W->sub[{lo, hi}] = c;
        return (int)KVX_OK;
    });
}

// kvx_dist_comm_fn: ctx = kvx_rccl*.  In null-stream order, as the callback's contract says (the library has put the null
// stream behind the factor's stream and puts the factor's stream behind the null stream afterwards).
int kvx_rccl_comm(void *ctx, const kvx_dist_op *op)
{
    return guarded([&] {
        kvx_rccl *W = (kvx_rccl *)ctx;
        if (!W || !op) { set_last_error("kvx_rccl_comm: no communicator"); return (int)KVX_EINVAL; }
        ncclComm_t c = W->world;
        int lo = 0;
        if (!(op->lo == 0 && op->hi == W->nranks)) {
            auto it = W->sub.find({op->lo, op->hi});
            if (it == W->sub.end()) { set_last_error("kvx_rccl_comm: no communicator for this rank range (kvx_rccl_split)"); return (int)KVX_EINVAL; }
            c = it->second;
            lo = op->lo;
        }
        if (op->count <= 0) return (int)KVX_OK;
        hipStream_t st = nullptr;
        if (op->kind == KVX_DIST_BCAST) NCCLCHK(api().Broadcast(op->buf_dev, op->buf_dev, (size_t)op->count, ncclDouble, op->root - lo, c, st));
        else if (op->kind == KVX_DIST_ALLREDUCE) NCCLCHK(api().AllReduce(op->buf_dev, op->buf_dev, (size_t)op->count, ncclDouble, ncclSum, c, st));
        else if (op->kind == KVX_DIST_ALLREDUCE_MIN) NCCLCHK(api().AllReduce(op->buf_dev, op->buf_dev, (size_t)op->count, ncclDouble, ncclMin, c, st));
        else { set_last_error("kvx_rccl_comm: unknown collective kind"); return (int)KVX_EINVAL; }
        W->ncoll++;
        W->bytes += 8 * op->count;
        return (int)KVX_OK;
    });
}

// Small host-value collectives over all ranks (timing rule of bench.py, barriers): vals[n] in place; op 0 = SUM, 1 = MAX, 2 = MIN.
int kvx_rccl_allreduce_host(kvx_rccl *W, double *vals, int n, int op)
{
    return guarded([&] {
        if (!W || !vals || n < 0 || n > W->scratch_cap) { set_last_error("kvx_rccl_allreduce_host: bad arguments"); return (int)KVX_EINVAL; }
        if (n == 0) return (int)KVX_OK;
        HIPCHK(hipMemcpy(W->scratch, vals, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
        NCCLCHK(api().AllReduce(W->scratch, W->scratch, (size_t)n, ncclDouble, op == 1 ? ncclMax : (op == 2 ? ncclMin : ncclSum), W->world, nullptr));
        HIPCHK(hipMemcpy(vals, W->scratch, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));      // (null stream: ordered behind the collective)
        return (int)KVX_OK;
    });
}

int kvx_rccl_allgather_host(kvx_rccl *W, const double *mine, int n, double *all)
{
    return guarded([&] {
        if (!W || !mine || !all || n < 0 || n > W->scratch_cap) { set_last_error("kvx_rccl_allgather_host: bad arguments"); return (int)KVX_EINVAL; }
        if (n == 0) return (int)KVX_OK;
        double *mine_d = W->scratch + (size_t)W->rank * (size_t)n;       // in-place form: the rank's block inside the receive buffer
        HIPCHK(hipMemcpy(mine_d, mine, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
        NCCLCHK(api().AllGather(mine_d, W->scratch, (size_t)n, ncclDouble, W->world, nullptr));
        HIPCHK(hipMemcpy(all, W->scratch, (size_t)n * (size_t)W->nranks * sizeof(double), hipMemcpyDeviceToHost));
        return (int)KVX_OK;
    });
}

// every rank has reached this call and the device work each enqueued before it is complete
int kvx_rccl_barrier(kvx_rccl *W)
{
    return guarded([&] {
        if (!W) return (int)KVX_EINVAL;
        HIPCHK(hipDeviceSynchronize());
        double one = 1.0;
        HIPCHK(hipMemcpy(W->scratch, &one, sizeof(double), hipMemcpyHostToDevice));
        NCCLCHK(api().AllReduce(W->scratch, W->scratch, 1, ncclDouble, ncclSum, W->world, nullptr));
        HIPCHK(hipStreamSynchronize(nullptr));
        return (int)KVX_OK;
    });
}

int kvx_rccl_stats(kvx_rccl *W, int64_t out[2])
{
    if (!W || !out) return KVX_EINVAL;
    out[0] = W->ncoll; out[1] = W->bytes;
    return KVX_OK;
}

void kvx_rccl_free(kvx_rccl *W)
{
    if (!W) return;
    (void)hipDeviceSynchronize();
    for (auto &kv : W->sub) (void)api().CommDestroy(kv.second);
    if (W->world) (void)api().CommDestroy(W->world);
    if (W->scratch) (void)hipFree(W->scratch);
    delete W;
}

}  // extern "C"
