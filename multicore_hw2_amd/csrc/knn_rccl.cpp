// knn_rccl.cpp — the one exchange step of the path: min-reduce the per-GPU packed keys with RCCL.
//
// The reference gathers every GPU's winners to the host under `omp critical` and re-ranks them on the
// CPU (sources/src/core.cu:925-957, wrong for m > 1: core.cu:941-943).  Here every GPU holds
// key[m] = (float_bits(d2) << 32) | global_index for its shard, and ONE
// ncclAllReduce(ncclUint64, ncclMin) per GPU inside ncclGroupStart/End leaves the global answer on all of
// them: unsigned order of the key = lexicographic (distance, index) = v0's first strict minimum.
// Single process, one communicator per device (ncclCommInitAll), created ONCE per process.
//
// RCCL is opened with dlopen at first use, not linked: librccl.so is 570 MB, a single-GPU caller never
// needs it, and a host program that already carries an RCCL (PyTorch bundles one under the same SONAME)
// keeps exactly one copy in the process.
#include "knn_common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>
#include <vector>

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    std::string why;   // set when loading failed
};

std::mutex g_mu;
RcclApi g_api;
bool g_tried = false;
// ONE communicator set per process: ncclCommInitAll costs seconds on an 8-GPU node, and a caller that varied its device
// list from call to call (the round-3 drop-in did: {0,1} for n = 2, {0..7} for larger sets) would pay that every time.
// The first reduction fixes the device list; a later call with another list is refused (the drop-in then merges on the
// host, index-API callers get the error).
std::vector<int> g_comm_devs;
std::vector<ncclComm_t> g_comms;
int g_comm_sets = 0;   // communicator sets created so far (0 or 1): knn_get_option("rccl_comm_sets")

bool load_api_locked()
{
    if (g_tried)
        return g_api.handle != nullptr;
    g_tried = true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *name : names) {
        g_api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (g_api.handle)
            break;
    }
    if (!g_api.handle) {
        const char *e = dlerror();
        g_api.why = std::string("dlopen(librccl): ") + (e ? e : "not found");
        return false;
    }
    bool ok = true;
    auto sym = [&](const char *name) -> void * {
        void *p = dlsym(g_api.handle, name);
        if (!p) {
            ok = false;
            g_api.why = std::string("librccl has no symbol ") + name;
        }
        return p;
    };
    g_api.CommInitAll = (decltype(g_api.CommInitAll))sym("ncclCommInitAll");
    g_api.CommDestroy = (decltype(g_api.CommDestroy))sym("ncclCommDestroy");
    g_api.AllReduce = (decltype(g_api.AllReduce))sym("ncclAllReduce");
    g_api.GroupStart = (decltype(g_api.GroupStart))sym("ncclGroupStart");
    g_api.GroupEnd = (decltype(g_api.GroupEnd))sym("ncclGroupEnd");
    g_api.GetErrorString = (decltype(g_api.GetErrorString))sym("ncclGetErrorString");
    g_api.GetVersion = (decltype(g_api.GetVersion))sym("ncclGetVersion");
    if (!ok) {
        dlclose(g_api.handle);
        g_api.handle = nullptr;
    }
    return ok;
}

}  // namespace

// 1 if RCCL could be opened (tries once), else 0 with the reason in *why.
int knn_rccl_available(std::string *why)
{
    std::lock_guard<std::mutex> lock(g_mu);
    const bool ok = load_api_locked();
    if (!ok && why)
        *why = g_api.why;
    return ok ? 1 : 0;
}

int knn_rccl_version()
{
    std::lock_guard<std::mutex> lock(g_mu);
    int v = 0;
    if (load_api_locked() && g_api.GetVersion(&v) == ncclSuccess)
        return v;
    return 0;
}

int knn_rccl_comm_sets()
{
    std::lock_guard<std::mutex> lock(g_mu);
    return g_comm_sets;
}

// recv[g] (null array: keys[g] itself, in place) <- elementwise unsigned minimum over g of keys[g] (m packed keys on
// devices[g]), enqueued on streams[g] (null = the device's default stream).  Returns 0, or -1 with a message in err;
// with a separate recv the keys are never written, so a caller can still merge them another way after a failure.
int knn_rccl_allreduce_min(int ndev, const int *devices, u64 *const *keys, int m, const hipStream_t *streams,
                           std::string &err, u64 *const *recv)
{
    if (ndev < 1 || !devices || !keys || m < 0) {
        err = "knn_rccl_allreduce_min: bad arguments";
        return -1;
    }
    if (m == 0)
        return 0;
    std::lock_guard<std::mutex> lock(g_mu);   // also serialises group calls of one process
    if (!load_api_locked()) {
        err = g_api.why;
        return -1;
    }
    const std::vector<int> devs(devices, devices + ndev);
    if (g_comms.empty()) {
        std::vector<ncclComm_t> comms((size_t)ndev);
        const ncclResult_t r = g_api.CommInitAll(comms.data(), ndev, devs.data());
        if (r != ncclSuccess) {
            err = std::string("ncclCommInitAll: ") + g_api.GetErrorString(r);
            return -1;
        }
        g_comms = std::move(comms);
        g_comm_devs = devs;
        ++g_comm_sets;
    } else if (devs != g_comm_devs) {
        err = "this process already holds an RCCL communicator set for another device list (one set per process: "
              "reduce over the same GPUs every time, or merge the keys on the host)";
        return -1;
    }
    const std::vector<ncclComm_t> &comms = g_comms;
    int prev = -1;
    (void)hipGetDevice(&prev);
    ncclResult_t r = g_api.GroupStart();
    for (int g = 0; g < ndev && r == ncclSuccess; ++g) {
        if (hipSetDevice(devs[(size_t)g]) != hipSuccess) {
            r = ncclUnhandledCudaError;
            break;
        }
        r = g_api.AllReduce(keys[g], recv ? recv[g] : keys[g], (size_t)m, ncclUint64, ncclMin, comms[(size_t)g],
                            streams ? streams[g] : (hipStream_t) nullptr);
    }
    const ncclResult_t rend = g_api.GroupEnd();
    if (prev >= 0)
        (void)hipSetDevice(prev);
    if (r == ncclSuccess)
        r = rend;
    if (r != ncclSuccess) {
        err = std::string("ncclAllReduce(uint64, min): ") + g_api.GetErrorString(r);
        return -1;
    }
    return 0;
}
