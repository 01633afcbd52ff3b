// knn_api.cpp — host side of libknn_mi355x.so: the C-ABI of include/knn_mi355x.h.
//
// Mirrors the orchestration of the reference's v8::cudaCallback (sources/src/core.cu:856-958)
// for ONE purpose — same inputs, same outputs, same error behaviour — without its structure:
// one host thread per GPU like core.cu:873, but shard offsets are folded into packed keys on
// the GPU (instead of the host-side `+= offset`, core.cu:932-933) and the final reduction is an
// unsigned min over m keys per GPU (instead of the distance-recomputing CPU loop,
// core.cu:935-957, whose indexing is wrong for m > 1 — SURVEY.md §8 a2').
#include "../../include/knn_mi355x.h"
#include "knn_common.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <unordered_map>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const char *what, const char *detail = nullptr)
{
    g_err = what;
    if (detail) {
        g_err += ": ";
        g_err += detail;
    }
    return code;
}

#define HIP_TRY(call)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            char buf_[256];                                                                \
            snprintf(buf_, sizeof buf_, "%s:%d, code:%d, reason: %s", __FILE__, __LINE__,  \
                     (int)e_, hipGetErrorString(e_));                                      \
            g_err = buf_;                                                                  \
            return KNN_EHIP;                                                               \
        }                                                                                  \
    } while (0)

// Abort-on-error for the void drop-in entry (reference core.h:77-87: print and exit(1)).
[[noreturn]] void die(const char *file, int line, int code, const char *reason)
{
    printf("Error: %s:%d, code:%d, reason: %s \n", file, line, code, reason);
    fflush(stdout);
    exit(1);
}
#define DIE_IF(rc)                                                \
    do {                                                          \
        int rc_ = (rc);                                           \
        if (rc_ != KNN_OK)                                        \
            die(__FILE__, __LINE__, rc_, g_err.c_str());          \
    } while (0)

std::atomic<long long> g_opt_path{0};
std::atomic<long long> g_opt_shards{0};
std::atomic<long long> g_opt_filter_qt{0};
std::atomic<long long> g_opt_filter_rounds{0};
std::atomic<long long> g_opt_filter_chain{0};
std::atomic<long long> g_opt_stream{0};
std::atomic<long long> g_opt_scan_blocks{0};     // pruned scan, blocks per CU: 0 auto, 1, 2
std::atomic<long long> g_last_cells{0};          // read-only: shards of the most recent cudaCallback that were served by the pruned scan
std::atomic<long long> g_opt_sample_stride{0};   // deep-K scans: stride of the sample pass in tiles, 0 = policy
std::atomic<long long> g_opt_run_thresholds{0};  // deep-K scan: 0 / 1 running thresholds, 2 off (A/B)
std::atomic<long long> g_opt_cells_lists{0};     // pruned scan, who lists a cell's queries: 0 auto, 1 the match launch, 2 the scan's own waves
std::atomic<long long> g_opt_scan_deal{0};       // pruned scan, how waves get their items: 0 auto, 1 fixed deal, 2 block counter
std::atomic<long long> g_opt_cells_build{0};     // cell-sorted layout: 0 fast two-pass build (counted one if a bucket overflows), 1 the one-pass placement, 2 the counted two-pass build (A/B, tests)
std::atomic<long long> g_opt_cells{0};       // cell-sorted layouts (k <= 16): 0 resident indexes large enough to prune (index_create_impl), 1 from 2^17 rows, 2 never
std::atomic<long long> g_opt_ingest{0};      // indexes created from host rows: 0 layouts built under the copy, 1 copy then build
std::atomic<long long> g_opt_rccl{0};        // 0 auto (several GPUs, one shard each), 1 always, 2 never
std::atomic<long long> g_rccl_reductions{0}; // cudaCallback merges done by the RCCL all-reduce
std::atomic<long long> g_last_shards{0};     // shards of the most recent cudaCallback (tests, KNN_MI355X_TRACE_CALL)

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess)
            prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (prev >= 0)
            (void)hipSetDevice(prev);
    }
};

// Staging buffers of the host-input entry points (cudaCallback, knn_index_query_host, indexes created
// from host rows).  A one-shot call otherwise spends 0.3-0.5 ms in hipMalloc + hipFree (hipFree
// alone 0.22 ms) around a 0.1-0.4 ms scan at TA scale; the pool keeps up to kPoolSlots freed buffers
// per device (at most kPoolBytes in total) and hands one back when it is at least as large as, and
// at most twice, the size asked for.  Buffers are only returned to it after the stream work that
// used them has been waited for.
constexpr int kPoolDevices = 64;
constexpr size_t kPoolSlots = 64;
constexpr size_t kPoolBytes = 4ull << 30;
struct PoolEntry {
    void *p;
    size_t bytes;
};
std::mutex g_pool_mu;
std::vector<PoolEntry> g_pool[kPoolDevices];

// device `dev` must be current
hipError_t pool_get(int dev, size_t bytes, void **out)
{
    *out = nullptr;
    if (bytes == 0)
        bytes = 1;
    if (dev >= 0 && dev < kPoolDevices) {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        std::vector<PoolEntry> &v = g_pool[dev];
        size_t best = v.size();
        for (size_t i = 0; i < v.size(); ++i)
            if (v[i].bytes >= bytes && v[i].bytes / 2 <= bytes && (best == v.size() || v[i].bytes < v[best].bytes))
                best = i;
        if (best != v.size()) {
            *out = v[best].p;
            v.erase(v.begin() + (long)best);
            return hipSuccess;
        }
    }
    return hipMalloc(out, bytes);
}

// `bytes` = the size the buffer was asked for with (its real size may be larger: that is fine, the
// pool then under-estimates what it holds by at most 2x)
void pool_put(int dev, void *p, size_t bytes)
{
    if (!p)
        return;
    if (bytes == 0)
        bytes = 1;
    std::vector<void *> drop;
    if (dev >= 0 && dev < kPoolDevices && bytes <= kPoolBytes) {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        std::vector<PoolEntry> &v = g_pool[dev];
        size_t held = bytes;
        for (const PoolEntry &e : v)
            held += e.bytes;
        while (!v.empty() && (v.size() >= kPoolSlots || held > kPoolBytes)) {  // oldest first
            held -= v.front().bytes;
            drop.push_back(v.front().p);
            v.erase(v.begin());
        }
        v.push_back(PoolEntry{p, bytes});
    } else {
        drop.push_back(p);
    }
    for (void *d : drop)
        (void)hipFree(d);
}

std::unordered_map<void *, std::pair<int, size_t>> g_live;  // pooled buffers handed to the filter code

}  // namespace

hipError_t knn_dev_alloc(void **p, size_t bytes)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess)
        return e;
    e = pool_get(dev, bytes, p);
    if (e == hipSuccess && *p) {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        g_live[*p] = std::make_pair(dev, bytes);
    }
    return e;
}

// One device-wide wait covers any number of frees that follow it on this thread (knn_index_destroy gives ~15
// buffers back; a wait per buffer was 15 device-wide syncs).
static thread_local bool g_free_synced = false;
void knn_dev_free_begin_synced()
{
    (void)hipDeviceSynchronize();
    g_free_synced = true;
}
void knn_dev_free_end_synced() { g_free_synced = false; }

hipError_t knn_dev_free(void *p)
{
    if (!p)
        return hipSuccess;
    std::pair<int, size_t> info(-1, 0);
    {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        auto it = g_live.find(p);
        if (it != g_live.end()) {
            info = it->second;
            g_live.erase(it);
        }
    }
    if (info.first < 0)
        return hipFree(p);
    if (!g_free_synced)
        (void)hipDeviceSynchronize();  // hipFree's implicit wait: nothing may still be using the buffer
    pool_put(info.first, p, info.second);
    return hipSuccess;
}

namespace {

// Two non-blocking streams per device for the streamed one-shot path (created once: a stream costs
// ~0.1 ms to create, more than a TA-scale call).
struct DeviceStreams {
    hipStream_t copy = nullptr, compute = nullptr;
};
DeviceStreams g_streams[kPoolDevices];

// device `dev` must be current
hipError_t streams_get(int dev, DeviceStreams *out)
{
    if (dev < 0 || dev >= kPoolDevices)
        return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_pool_mu);
    DeviceStreams &d = g_streams[dev];
    if (!d.copy) {
        hipError_t e = hipStreamCreateWithFlags(&d.copy, hipStreamNonBlocking);
        if (e != hipSuccess)
            return e;
    }
    if (!d.compute) {
        hipError_t e = hipStreamCreateWithFlags(&d.compute, hipStreamNonBlocking);
        if (e != hipSuccess)
            return e;
    }
    *out = d;
    return hipSuccess;
}

}  // namespace

struct knn_index {
    int device = 0;
    int k = 0;
    long long n = 0;
    long long base = 0;
    int num_cu = 256;
    const float *refs = nullptr;  // device, AoS [n][k]
    float *owned_refs = nullptr;  // set when the index copied the references itself (pooled)
    size_t owned_bytes = 0;
    long long stats[4] = {0, 0, 0, 0};
    FilterState filter;           // MFMA filter layouts + workspace (usable == false: exact only)
    GridState *grid = nullptr;    // k <= 4: uniform-grid spatial index (null: not built / ruled out)
    int timing = 0;            // 0 off, N > 0: bracket every N-th dominant-kernel launch with events
    unsigned long long timing_seq = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;  // one pair per timed launch
    size_t events_used = 0;
    int last_slot = 0;
    bool filter_wanted = false;  // the creator asked for the filter layouts explicitly (one-shot cost model)
    unsigned long long recent_slots = 0;   // the workspace slots of the last eight calls, one byte each (newest lowest)
    int recent_calls = 0;
    bool sharded = false;        // a cell-range shard (knn_index_create_sharded): always served by the cell-pruned path
    ShardGeom geom;              // its copy of the global grid (filter.cells->geom points here)
    int rank = 0;
    // Calls on one index from several host threads are serialised (enqueueing a batch is ~20 us of host work; the GPU
    // work of different slots still overlaps): the workspaces' lazily grown buffers, the event list, the statistics and
    // the chain events are plain members.  Recursive: knn_index_query_host calls the keyed entry points.
    std::recursive_mutex mu;
};

extern "C" {

const char *knn_last_error(void) { return g_err.c_str(); }

const char *knn_version(void) { return "knn_mi355x 0.1 (gfx950)"; }

long long knn_trim(void)
{
    long long released = 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess)
        ndev = 0;
    for (int dev = 0; dev < kPoolDevices && dev < ndev; ++dev) {
        std::vector<PoolEntry> take;
        {
            std::lock_guard<std::mutex> lock(g_pool_mu);
            take.swap(g_pool[dev]);
        }
        if (take.empty())
            continue;
        DeviceGuard guard(dev);
        for (const PoolEntry &e : take) {
            (void)hipFree(e.p);
            released += (long long)e.bytes;
        }
    }
    return released;
}

int knn_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

int knn_set_option(const char *name, long long value)
{
    if (!name)
        return fail(KNN_EINVAL, "knn_set_option: null name");
    if (!strcmp(name, "path")) {
        if (value < 0 || value > 3)
            return fail(KNN_EINVAL, "knn_set_option: path must be 0, 1, 2 or 3");
        g_opt_path = value;
        return KNN_OK;
    }
    if (!strcmp(name, "shards")) {
        if (value < 0 || value > 1024)
            return fail(KNN_EINVAL, "knn_set_option: shards must be in [0, 1024]");
        g_opt_shards = value;
        return KNN_OK;
    }
    if (!strcmp(name, "filter_qt")) {
        if (value != 0 && value != 2 && value != 8 && value != 16 && value != 32)
            return fail(KNN_EINVAL, "knn_set_option: filter_qt must be 0, 2, 8, 16 or 32");
        g_opt_filter_qt = value;
        return KNN_OK;
    }
    if (!strcmp(name, "filter_chain")) {
        if (value < 0 || value > 2)
            return fail(KNN_EINVAL, "knn_set_option: filter_chain must be 0 (auto), 1 (chained) or 2 (free)");
        g_opt_filter_chain = value;
        return KNN_OK;
    }
    if (!strcmp(name, "stream")) {
        if (value < 0 || value > 2)
            return fail(KNN_EINVAL, "knn_set_option: stream must be 0 (auto), 1 (never) or 2 (whenever the shard is >= 32 MiB)");
        g_opt_stream = value;
        return KNN_OK;
    }
    if (!strcmp(name, "scan_blocks")) {
        if (value < 0 || value > 2)
            return fail(KNN_EINVAL, "knn_set_option: scan_blocks must be 0 (auto), 1 or 2");
        g_opt_scan_blocks = value;
        return KNN_OK;
    }
    if (!strcmp(name, "cells_build")) {
        if (value < 0 || value > 2)
            return fail(KNN_EINVAL, "knn_set_option: cells_build must be 0 (fast two-pass), 1 (one-pass placement) or 2 (counted two-pass)");
        g_opt_cells_build = value;
        return KNN_OK;
    }
    if (!strcmp(name, "sample_stride")) {
        if (value < 0 || value > 1024)
            return fail(KNN_EINVAL, "knn_set_option: sample_stride must be in [0, 1024]");
        g_opt_sample_stride = value;
        return KNN_OK;
    }
    if (!strcmp(name, "run_thresholds")) {
        if (value < 0 || value > 2)
            return fail(KNN_EINVAL, "knn_set_option: run_thresholds must be 0 (auto), 1 (on) or 2 (off)");
        g_opt_run_thresholds = value;
        return KNN_OK;
    }
    if (!strcmp(name, "cells_centre")) {
        if (value < 0 || value > 2)
            return fail(KNN_EINVAL, "knn_set_option: cells_centre must be 0 (auto: per-cell frames for clustered data), 1 (always) or 2 (never)");
        g_knn_cells_centre = (int)value;
        return KNN_OK;
    }
    if (!strcmp(name, "cells_lists")) {
        if (value < 0 || value > 2)
            return fail(KNN_EINVAL, "knn_set_option: cells_lists must be 0 (auto), 1 (match launch) or 2 (the scan lists its own items)");
        g_opt_cells_lists = value;
        return KNN_OK;
    }
    if (!strcmp(name, "scan_deal")) {
        if (value < 0 || value > 2)
            return fail(KNN_EINVAL, "knn_set_option: scan_deal must be 0 (auto), 1 (fixed deal) or 2 (block counter)");
        g_opt_scan_deal = value;
        return KNN_OK;
    }
    if (!strcmp(name, "cells")) {
        if (value < 0 || value > 2)
            return fail(KNN_EINVAL, "knn_set_option: cells must be 0 (library policy), 1 (always) or 2 (never)");
        g_opt_cells = value;
        return KNN_OK;
    }
    if (!strcmp(name, "ingest")) {
        if (value < 0 || value > 1)
            return fail(KNN_EINVAL, "knn_set_option: ingest must be 0 (layouts built under the copy) or 1 (copy, then build)");
        g_opt_ingest = value;
        return KNN_OK;
    }
    if (!strcmp(name, "rccl")) {
        if (value < 0 || value > 2)
            return fail(KNN_EINVAL, "knn_set_option: rccl must be 0 (auto), 1 (always) or 2 (never: host merge)");
        g_opt_rccl = value;
        return KNN_OK;
    }
    if (!strcmp(name, "filter_rounds")) {
        if (value < 0 || value > 64)
            return fail(KNN_EINVAL, "knn_set_option: filter_rounds must be in [0, 64]");
        g_opt_filter_rounds = value;
        return KNN_OK;
    }
    return fail(KNN_EINVAL, "knn_set_option: unknown option", name);
}

long long knn_get_option(const char *name)
{
    if (name && !strcmp(name, "path"))
        return g_opt_path;
    if (name && !strcmp(name, "shards"))
        return g_opt_shards;
    if (name && !strcmp(name, "filter_qt"))
        return g_opt_filter_qt;
    if (name && !strcmp(name, "filter_rounds"))
        return g_opt_filter_rounds;
    if (name && !strcmp(name, "filter_chain"))
        return g_opt_filter_chain;
    if (name && !strcmp(name, "stream"))
        return g_opt_stream;
    if (name && !strcmp(name, "rccl"))
        return g_opt_rccl;
    if (name && !strcmp(name, "ingest"))
        return g_opt_ingest;
    if (name && !strcmp(name, "cells"))
        return g_opt_cells;
    if (name && !strcmp(name, "scan_blocks"))
        return g_opt_scan_blocks;
    if (name && !strcmp(name, "scan_deal"))
        return g_opt_scan_deal;
    if (name && !strcmp(name, "cells_lists"))
        return g_opt_cells_lists;
    if (name && !strcmp(name, "cells_centre"))
        return (long long)g_knn_cells_centre.load();
    if (name && !strcmp(name, "cells_centred_builds"))   // read-only: cell-sorted layouts moved into per-cell frames so far
        return g_knn_cells_centred_builds.load();
    if (name && !strcmp(name, "run_thresholds"))
        return g_opt_run_thresholds;
    if (name && !strcmp(name, "sample_stride"))
        return g_opt_sample_stride;
    if (name && !strcmp(name, "cells_build"))
        return g_opt_cells_build;
    if (name && !strcmp(name, "rccl_reductions"))   // read-only: cudaCallback merges done by RCCL so far
        return g_rccl_reductions;
    if (name && !strcmp(name, "rccl_comm_sets"))    // read-only: RCCL communicator sets created by this process (at most one)
        return knn_rccl_comm_sets();
    if (name && !strcmp(name, "last_shards"))       // read-only: GPUs / shards the most recent cudaCallback was split over
        return g_last_shards;
    if (name && !strcmp(name, "last_cells"))        // read-only: how many of them the pruned scan served
        return g_last_cells;
    if (name && !strcmp(name, "rccl_version"))      // read-only: NCCL_VERSION_CODE of the loaded RCCL, 0 if none
        return knn_rccl_version();
    return -1;
}

}  // extern "C"

namespace {
// build_filter: 1 build the MFMA filter layouts, 2 build them cell-sorted (the pruned scan), 0 do not, -1 library policy
// build_grid (k <= 4): 1 build the grid index, 0 do not, -1 library policy
int index_create_impl(knn_index **out, int device, int k, long long n_local, const float *refs,
                      int refs_on_device, long long base_index, void *stream, int build_filter, int build_grid = -1);
}

extern "C" {

int knn_index_create(knn_index **out, int device, int k, long long n_local, const float *refs,
                     int refs_on_device, long long base_index, void *stream)
{
    return index_create_impl(out, device, k, n_local, refs, refs_on_device, base_index, stream, -1);
}

}  // extern "C"

namespace {
int index_create_impl(knn_index **out, int device, int k, long long n_local, const float *refs,
                      int refs_on_device, long long base_index, void *stream, int build_filter, int build_grid)
{
    if (!out)
        return fail(KNN_EINVAL, "knn_index_create: null out");
    *out = nullptr;
    if (k < 1 || n_local < 0 || base_index < 0 || (n_local > 0 && !refs))
        return fail(KNN_EINVAL, "knn_index_create: bad k, n_local, base_index or refs");
    if (base_index + n_local > 0x7FFFFFFFll + 1)
        return fail(KNN_EINVAL, "knn_index_create: global index exceeds int32 (results are int)");
    int ndev = knn_device_count();
    if (ndev < 1)
        return fail(KNN_ENODEV, "knn_index_create: no HIP device visible");
    if (device < 0 || device >= ndev)
        return fail(KNN_EINVAL, "knn_index_create: device out of range");
    DeviceGuard guard(device);
    if (!guard.ok)
        return fail(KNN_EHIP, "knn_index_create: hipSetDevice failed");

    knn_index *idx = new (std::nothrow) knn_index();
    if (!idx)
        return fail(KNN_ENOMEM, "knn_index_create: out of host memory");
    idx->device = device;
    idx->k = k;
    idx->n = n_local;
    idx->base = base_index;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        idx->num_cu = prop.multiProcessorCount;

    hipStream_t s = (hipStream_t)stream;
    // MFMA filter layouts (skipped for small shards and when the exact path is forced)
    idx->filter_wanted = build_filter > 0;
    // cell-sorted layout for the pruned scan: indexes the caller keeps (library policy) or on request; the
    // one-shot cudaCallback asks for explicit layouts and answers one batch, which does not repay the sort.
    // Pruning needs enough cells for the dimension: measured on uniform data at m = 1024 (profiles/r02_cells_policy.txt)
    // the pruned scan wins from 2^19 rows for k <= 12 and from 2^20 rows for k = 13..16; below that the lists
    // hold most of the batch and the full scan's register-resident loop is the faster way to score them.
    // (16 < k <= 32, round 5: the cells cut the first 16 dimensions; policy from the measurements in profiles/r05_cells_k17_32.txt:
    // uniform data, m = 1024, ms per step pruned / full scan on one box — n = 2^24: k 17 0.211 / 0.94, 20 0.265 / 0.96, 22 0.424 / 0.962,
    // 24 0.625 / 0.976, 25 0.716 / 0.980, 26 1.06 / 0.99 (loses); n = 2^23: k 21 0.238 / 0.465, 23 0.354 / 0.476, 24 0.445 / 0.477 (a wash),
    // 25 0.579 / 0.480 (loses); n = 2^22: k 20 0.143 / 0.25, 21 0.174 / 0.236, 22 0.220 / 0.240 and 23 0.219 / 0.243 (a wash), 24 loses)
    const long long cells_from = k <= 12 ? (1ll << 19) : k <= 16 ? (1ll << 20) : k <= 21 ? (1ll << 22) : k <= 23 ? (1ll << 23)
                                 : k <= KNN_CELLS_AUTO_MAX_K ? (1ll << 24) : (1ll << 62);
    const bool want_cells = k <= 32 && n_local >= (1ll << 17) &&
                            (g_opt_cells == 1 || (g_opt_cells == 0 && build_filter == 2) ||
                             (g_opt_cells == 0 && build_filter < 0 && n_local >= cells_from));
    if (build_filter < 0) {
        // library policy: shards of >= 65536 rows; for 32 < k <= 4096 (3k+3 exact lane-ops per pair,
        // one query per lane above k = 64, row-per-lane kernels above 128) the MFMA filter pays off from 4096 rows already
        build_filter = g_opt_path == 2 || n_local >= 65536 || (k > 32 && k <= KNN_FILTER_MAX_K && n_local >= 4096);
        idx->filter_wanted = build_filter && n_local < 65536 && g_opt_path != 2;
    }
    const bool grid_planned = k <= 4 && (g_opt_path == 3 || (g_opt_path == 0 && build_grid != 0 && (build_grid > 0 || n_local >= 16384)));
    // (a shard the grid index will serve gets no MFMA layouts unless the grid turns out to be ruled out)
    bool want_layouts = n_local > 0 && g_opt_path != 1 && g_opt_path != 3 && build_filter;
    bool layouts_done = false;
    if (n_local > 0) {
        if (refs_on_device) {
            idx->refs = refs;
        } else {
            const size_t bytes = (size_t)n_local * (size_t)k * sizeof(float);
            hipError_t e = pool_get(device, bytes, (void **)&idx->owned_refs);
            idx->owned_bytes = bytes;
            if (e != hipSuccess) {
                delete idx;
                return fail(KNN_ENOMEM, "knn_index_create: hipMalloc(refs)", hipGetErrorString(e));
            }
            DeviceStreams ds;
            if (want_layouts && !grid_planned && !want_cells && g_opt_ingest != 1 && streams_get(device, &ds) == hipSuccess) {
                // ingest: rows and filter layouts in one pass over PCIe (knn_filter_build_from_host)
                e = hipStreamSynchronize(s);   // work the caller queued ahead of this call
                if (e == hipSuccess)
                    e = knn_filter_build_from_host(idx->filter, k, n_local, idx->owned_refs, refs, ds.copy, ds.compute);
                layouts_done = e == hipSuccess;
            } else if (want_layouts && !grid_planned && want_cells && g_opt_ingest != 1 && g_opt_cells_build == 0 &&
                       streams_get(device, &ds) == hipSuccess) {
                // ingest into a cell-sorted index: the fast build's bucket pass under the copy (round 5)
                e = hipStreamSynchronize(s);   // work the caller queued ahead of this call
                if (e == hipSuccess)
                    e = knn_filter_build_cells_from_host(idx->filter, k, n_local, idx->owned_refs, refs, ds.copy, ds.compute);
                layouts_done = e == hipSuccess && idx->filter.usable;   // (else: the rows are there, the build below sorts them)
                if (e == hipSuccess && !layouts_done)
                    knn_filter_free(idx->filter);
            } else {
                e = hipMemcpyAsync(idx->owned_refs, refs, bytes, hipMemcpyHostToDevice, s);
                if (e == hipSuccess)
                    e = hipStreamSynchronize(s);  // the host buffer may be freed right after return
            }
            if (e != hipSuccess) {
                (void)hipDeviceSynchronize();
                knn_filter_free(idx->filter);
                pool_put(device, idx->owned_refs, bytes);
                delete idx;
                return fail(KNN_EHIP, "knn_index_create: H2D copy of refs", hipGetErrorString(e));
            }
            idx->refs = idx->owned_refs;
        }
    }
    // k <= 4: the uniform-grid index — resident shards of >= 16384 rows, or whenever forced
    if (n_local > 0 && grid_planned) {
        const hipError_t e = knn_grid_build(&idx->grid, k, n_local, idx->refs, s);
        if (e != hipSuccess) {
            knn_index_destroy(idx);
            return fail(KNN_EHIP, "knn_index_create: building the grid index", hipGetErrorString(e));
        }
        if (idx->grid)
            want_layouts = false;
    }
    if (want_layouts && !layouts_done) {
        hipError_t e = knn_filter_build(idx->filter, k, n_local, idx->refs, s, want_cells ? (g_opt_cells_build == 1 ? 2 : g_opt_cells_build == 2 ? 3 : 1) : 0);
        if (e == hipErrorOutOfMemory) {
            // no room for the fp16 layouts beside the rows: the index still works, exact kernels only
            (void)hipGetLastError();
            knn_filter_free(idx->filter);
            e = hipSuccess;
        }
        if (e != hipSuccess) {
            knn_index_destroy(idx);
            return fail(KNN_EHIP, "knn_index_create: building the filter layouts", hipGetErrorString(e));
        }
    }
    *out = idx;
    return KNN_OK;
}
}  // namespace

// ---------------------------------------------------------------------------------------------
// Cell-range shards (round 4): one global grid, every rank holds a contiguous range of its cell codes.
// ---------------------------------------------------------------------------------------------
struct knn_geom {
    ShardGeom g;
};

extern "C" {

int knn_geom_create(knn_geom **out, int k, long long n_global, int nranks, const float *sample_host, long long samples,
                    int seed_tiles)
{
    if (!out)
        return fail(KNN_EINVAL, "knn_geom_create: null out");
    *out = nullptr;
    if (k < 1 || n_global < 1 || nranks < 1 || !sample_host || samples < 1 || seed_tiles < 0)
        return fail(KNN_EINVAL, "knn_geom_create: bad arguments");
    knn_geom *g = new (std::nothrow) knn_geom();
    if (!g)
        return fail(KNN_ENOMEM, "knn_geom_create: out of host memory");
    if (!knn_geom_from_sample(g->g, k, n_global, nranks, sample_host, samples, seed_tiles == 0 ? 2 : seed_tiles)) {
        delete g;
        return fail(KNN_EINVAL, "knn_geom_create: the set does not suit cell-range shards (k > 16, fewer than 512 cells of >= 144 "
                                "rows per rank, fewer than 64 sample rows, or a sample that is not finite / has no extent): "
                                "shard by index range instead (knn_index_create with base_index)");
    }
    *out = g;
    return KNN_OK;
}

void knn_geom_destroy(knn_geom *g) { delete g; }

int knn_geom_info(const knn_geom *g, long long out[8])
{
    if (!g || !out)
        return fail(KNN_EINVAL, "knn_geom_info: bad arguments");
    out[0] = g->g.bits;
    out[1] = g->g.ncells;
    out[2] = g->g.cells_per_rank;
    out[3] = g->g.seed_tiles;
    out[4] = (long long)g->g.part_bytes();
    out[5] = (long long)g->g.part_bytes() * g->g.nranks;
    out[6] = g->g.sa;
    out[7] = g->g.nranks;
    return KNN_OK;
}

long long knn_geom_first_cell(const knn_geom *g, int rank)
{
    if (!g || rank < 0)
        return fail(KNN_EINVAL, "knn_geom_first_cell: bad arguments");
    return (long long)g->g.first_cell(rank);   // (rank >= ranks: the number of cells)
}

int knn_geom_assign(const knn_geom *g, int device, const float *rows_dev, long long n, int *owner_dev, void *stream)
{
    if (!g || n < 0 || (n > 0 && (!rows_dev || !owner_dev)))
        return fail(KNN_EINVAL, "knn_geom_assign: bad arguments");
    DeviceGuard guard(device);
    if (!guard.ok)
        return fail(KNN_EHIP, "knn_geom_assign: hipSetDevice failed");
    HIP_TRY(knn_geom_assign_launch(g->g, rows_dev, n, owner_dev, (hipStream_t)stream));
    return KNN_OK;
}

int knn_index_create_sharded(knn_index **out, int device, const knn_geom *g, int rank, long long n_local, const float *refs_dev,
                             const unsigned *gids_dev, void *stream)
{
    if (!out)
        return fail(KNN_EINVAL, "knn_index_create_sharded: null out");
    *out = nullptr;
    if (!g || rank < 0 || rank >= g->g.nranks || n_local < 0 || (n_local > 0 && (!refs_dev || !gids_dev)))
        return fail(KNN_EINVAL, "knn_index_create_sharded: bad geometry, rank, n_local or pointers");
    const int ndev = knn_device_count();
    if (ndev < 1)
        return fail(KNN_ENODEV, "knn_index_create_sharded: no HIP device visible");
    if (device < 0 || device >= ndev)
        return fail(KNN_EINVAL, "knn_index_create_sharded: device out of range");
    DeviceGuard guard(device);
    if (!guard.ok)
        return fail(KNN_EHIP, "knn_index_create_sharded: hipSetDevice failed");
    hipStream_t s = (hipStream_t)stream;
    if (n_local > 0) {
        // the rows' global numbers: strictly ascending (ties between equal distances are decided by LOCAL row order inside
        // the shard, which then is global order) and inside int32 (results are int)
        unsigned bad = 0u, last = 0u;
        HIP_TRY(knn_gids_check(gids_dev, n_local, &bad, s));
        HIP_TRY(hipMemcpy(&last, gids_dev + (n_local - 1), sizeof last, hipMemcpyDeviceToHost));
        if (bad != 0u || last > 0x7FFFFFFFu)
            return fail(KNN_EINVAL, "knn_index_create_sharded: gids must be strictly ascending and below 2^31");
    }
    knn_index *idx = new (std::nothrow) knn_index();
    if (!idx)
        return fail(KNN_ENOMEM, "knn_index_create_sharded: out of host memory");
    idx->device = device;
    idx->k = g->g.k;
    idx->n = n_local;
    idx->base = 0;   // keys carry LOCAL rows until the batch's last kernel translates them through gids
    idx->sharded = true;
    idx->geom = g->g;
    idx->rank = rank;
    idx->refs = refs_dev;
    idx->filter_wanted = true;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        idx->num_cu = prop.multiProcessorCount;
    if (n_local > 0) {
        unsigned bad_rows = 0u;
        const hipError_t e = knn_filter_build(idx->filter, idx->k, n_local, refs_dev, s, g_opt_cells_build == 1 ? 2 : g_opt_cells_build == 2 ? 3 : 1, &idx->geom,
                                              rank, &bad_rows);
        if (e != hipSuccess || !idx->filter.usable || !idx->filter.cells) {
            char why[160];
            if (bad_rows != 0u)
                snprintf(why, sizeof why, "%u row(s) lie outside rank %d's cell range of this geometry (knn_geom_assign says where rows belong)",
                         bad_rows, rank);
            else
                snprintf(why, sizeof why, "%s", e != hipSuccess ? hipGetErrorString(e) : "the rows do not suit the layouts (mostly outside the global box, or not finite)");
            knn_index_destroy(idx);
            return fail(e != hipSuccess ? KNN_EHIP : KNN_EINVAL, "knn_index_create_sharded", why);
        }
        idx->filter.cells->gids = gids_dev;
        idx->filter.cells->geom = &idx->geom;
    }
    *out = idx;
    return KNN_OK;
}

int knn_index_seed_export(knn_index *idx, void *layer_dev, void *stream)
{
    if (!idx || !idx->sharded || !layer_dev)
        return fail(KNN_EINVAL, "knn_index_seed_export: needs a cell-range shard and the layer buffer");
    std::lock_guard<std::recursive_mutex> lock(idx->mu);
    DeviceGuard guard(idx->device);
    if (idx->n == 0) {   // a rank without rows: its part is all padding (zero fragments never score: +INF norms)
        const size_t pb = idx->geom.part_bytes();
        unsigned char *part = (unsigned char *)layer_dev + (size_t)idx->rank * pb;
        const size_t frag_bytes = (size_t)idx->geom.cells_per_rank * idx->geom.seed_tiles * 1024u;
        HIP_TRY(hipMemsetAsync(part, 0, KNN_SEED_HEADER_BYTES + frag_bytes, (hipStream_t)stream));
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)(part + KNN_SEED_HEADER_BYTES + frag_bytes), 0x00007C00,
                                  (size_t)idx->geom.cells_per_rank * idx->geom.seed_tiles * 32u, (hipStream_t)stream));
        return KNN_OK;
    }
    HIP_TRY(knn_cells_seed_export(idx->filter, idx->rank, (unsigned char *)layer_dev, (hipStream_t)stream));
    return KNN_OK;
}

int knn_index_seed_attach(knn_index *idx, const void *layer_dev)
{
    if (!idx || !idx->sharded || !layer_dev)
        return fail(KNN_EINVAL, "knn_index_seed_attach: needs a cell-range shard and the complete layer");
    std::lock_guard<std::recursive_mutex> lock(idx->mu);
    if (idx->n == 0)
        return KNN_OK;
    DeviceGuard guard(idx->device);
    // the bound constants must cover every rank's fragments: largest coordinate and norm over all parts' headers
    float bmax = idx->filter.bmax, nmax = idx->filter.nmax;
    for (int r = 0; r < idx->geom.nranks; ++r) {
        float hdr[2] = {0.0f, 0.0f};
        HIP_TRY(hipMemcpy(hdr, (const unsigned char *)layer_dev + (size_t)r * idx->geom.part_bytes(), sizeof hdr, hipMemcpyDeviceToHost));
        if (!(hdr[0] >= 0.0f) || !(hdr[1] >= 0.0f) || !(hdr[0] < INFINITY) || !(hdr[1] < INFINITY))
            return fail(KNN_EINVAL, "knn_index_seed_attach: a part of the layer has no valid header (was every rank's part exported and gathered?)");
        bmax = fmaxf(bmax, hdr[0]);
        nmax = fmaxf(nmax, hdr[1]);
    }
    idx->filter.bmax = bmax;
    idx->filter.nmax = nmax;
    idx->filter.cells->seed_layer = (const unsigned char *)layer_dev;
    return KNN_OK;
}

}  // extern "C"

extern "C" {

void knn_index_destroy(knn_index *idx)
{
    if (!idx)
        return;
    {
        DeviceGuard guard(idx->device);
        // the pool must never hand out memory a kernel may still be reading: ONE device-wide wait, then
        // every buffer of the index goes back
        knn_dev_free_begin_synced();
        if (idx->owned_refs)
            pool_put(idx->device, idx->owned_refs, idx->owned_bytes);
        knn_filter_free(idx->filter);
        knn_grid_free(idx->grid);
        knn_dev_free_end_synced();
        for (auto &ev : idx->events) {
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
    }
    delete idx;
}

int knn_keys_init(int device, unsigned long long *keys_dev, int m, void *stream)
{
    if (m < 0 || (m > 0 && !keys_dev))
        return fail(KNN_EINVAL, "knn_keys_init: bad arguments");
    DeviceGuard guard(device);
    if (!guard.ok)
        return fail(KNN_EHIP, "knn_keys_init: hipSetDevice failed");
    HIP_TRY(knn_keys_fill_launch((u64 *)keys_dev, m, (hipStream_t)stream));
    return KNN_OK;
}

int knn_index_query_keys(knn_index *idx, int m, const float *queries_dev,
                         unsigned long long *keys_dev, void *stream)
{
    return knn_index_query_keys_slot(idx, 0, m, queries_dev, keys_dev, stream);
}

int knn_index_query_keys_slot(knn_index *idx, int slot, int m, const float *queries_dev,
                              unsigned long long *keys_dev, void *stream)
{
    return knn_index_query_keys_ex(idx, slot, m, queries_dev, keys_dev, stream, 0u);
}

int knn_index_query_keys_ex(knn_index *idx, int slot, int m, const float *queries_dev,
                            unsigned long long *keys_dev, void *stream, unsigned flags)
{
    return knn_index_query(idx, slot, m, queries_dev, keys_dev, nullptr, stream, flags);
}

int knn_index_query(knn_index *idx, int slot, int m, const float *queries_dev, unsigned long long *keys_dev,
                    int *indices_dev, void *stream, unsigned flags)
{
    if (!idx || m < 0 || slot < 0 || slot >= KNN_SLOTS || (m > 0 && (!queries_dev || !keys_dev)) ||
        (flags & ~(unsigned)KNN_QUERY_INIT_KEYS) != 0u)
        return fail(KNN_EINVAL, "knn_index_query_keys: bad arguments");
    if (m == 0)
        return KNN_OK;
    std::lock_guard<std::recursive_mutex> lock(idx->mu);
    DeviceGuard guard(idx->device);
    if (!guard.ok)
        return fail(KNN_EHIP, "knn_index_query_keys: hipSetDevice failed");
    const bool init_keys = (flags & KNN_QUERY_INIT_KEYS) != 0u;
    if (idx->sharded && !init_keys)
        return fail(KNN_EINVAL, "knn_index_query: a cell-range shard WRITES its keys (pass KNN_QUERY_INIT_KEYS): they carry local "
                                "rows until the batch's last kernel, so they cannot fold into keys other shards have written");
    if (idx->n == 0) {   // an empty shard leaves (+INF, 0)
        if (init_keys)
            HIP_TRY(knn_keys_fill_launch((u64 *)keys_dev, m, (hipStream_t)stream));
        if (indices_dev)
            HIP_TRY(knn_keys_unpack_launch((const u64 *)keys_dev, m, indices_dev, (hipStream_t)stream));
        return KNN_OK;
    }
    idx->stats[0] = 1;
    idx->stats[1] = 0;
    idx->stats[2] = 0;
    hipStream_t s = (hipStream_t)stream;
    std::pair<hipEvent_t, hipEvent_t> *ev = nullptr;
    if (idx->timing > 0 && idx->timing_seq++ % (unsigned long long)idx->timing == 0) {
        if (idx->events_used == idx->events.size()) {
            std::pair<hipEvent_t, hipEvent_t> fresh;
            HIP_TRY(hipEventCreate(&fresh.first));
            HIP_TRY(hipEventCreate(&fresh.second));
            idx->events.push_back(fresh);
        }
        ev = &idx->events[idx->events_used++];
    }
    const long long path = g_opt_path;
    const bool cells_live = idx->filter.cells && (g_opt_cells != 2 || idx->sharded);
    const bool use_filter = idx->sharded ||
                            (idx->filter.usable && !(idx->grid && (path == 0 || path == 3)) &&
                             (path == 2 || (path == 0 && (m >= 5 || cells_live) && (idx->n >= 65536 || idx->filter_wanted))));
    // the filter paths start the keys themselves when asked to (the cell-pruned one inside its first kernel)
    if (init_keys && !use_filter)
        HIP_TRY(knn_keys_fill_launch((u64 *)keys_dev, m, s));
    if (idx->grid && (path == 0 || path == 3)) {
        // k <= 4 spatial index: one wave per query walks the grid rings; the brute-force scan behind it is
        // gated on the "some query gave up" word (far-outside queries, empty regions)
        idx->stats[0] = 3;
        idx->last_slot = slot;
        if (ev)
            HIP_TRY(hipEventRecord(ev->first, s));
        const unsigned *gate = nullptr;
        HIP_TRY(knn_grid_query(idx->grid, slot, m, queries_dev, idx->base, (u64 *)keys_dev, &gate, s));
        if (ev)
            HIP_TRY(hipEventRecord(ev->second, s));
        HIP_TRY(knn_exact_launch(idx->k, m, idx->n, idx->base, queries_dev, idx->refs, (u64 *)keys_dev, idx->num_cu, gate, s));
        if (indices_dev)
            HIP_TRY(knn_keys_unpack_launch((const u64 *)keys_dev, m, indices_dev, s));
        return KNN_OK;
    }
    // (with a cell-sorted layout even one query is served faster by the pruned scan than by reading the shard)
    if (use_filter) {
        // the event pair brackets the MFMA filter kernel alone (the dominant kernel)
        idx->stats[0] = 2;
        idx->filter.force_qt = (int)g_opt_filter_qt;
        idx->filter.force_rounds = (int)g_opt_filter_rounds;
        idx->filter.chain_policy = (int)g_opt_filter_chain;
        idx->filter.cells_policy = idx->sharded ? 0 : (int)g_opt_cells;   // (a cell-range shard has no other layout)
        // batches in flight on several workspace slots = a caller after throughput: the pruned scan of a small shard then
        // takes ONE block per CU, so that the next batch's preparation kernels find registers beside it (knn_cells_query).
        // Derived from the last eight calls (the flag used to stick for the life of the index once any call
        // had named a slot other than 0 — a caller that went back to one batch at a time kept the throughput shapes).
        idx->recent_slots = (idx->recent_slots << 8) | (unsigned long long)(unsigned)slot;
        if (idx->recent_calls < 8)
            ++idx->recent_calls;
        {
            bool several = false;
            for (int i = 1; i < idx->recent_calls; ++i)
                several = several || ((idx->recent_slots >> (8 * i)) & 0xFFull) != (unsigned long long)(unsigned)slot;
            idx->filter.several_slots = several;
        }
        idx->filter.scan_blocks = (int)g_opt_scan_blocks;
        idx->filter.scan_deal = (int)g_opt_scan_deal;
        idx->filter.cells_lists = (int)g_opt_cells_lists;
        idx->filter.run_thresholds = (int)g_opt_run_thresholds;
        idx->filter.sample_stride = (int)g_opt_sample_stride;
        idx->last_slot = slot;
        HIP_TRY(knn_filter_query(idx->filter, slot, m, queries_dev, idx->refs, idx->base, (u64 *)keys_dev,
                                 idx->num_cu, s, ev ? ev->first : nullptr, ev ? ev->second : nullptr, init_keys, indices_dev));
        if (idx->filter.ws[slot].last_used_cells)
            idx->stats[0] = 4;
        return KNN_OK;
    }
    if (ev)
        HIP_TRY(hipEventRecord(ev->first, s));
    HIP_TRY(knn_exact_launch(idx->k, m, idx->n, idx->base, queries_dev, idx->refs, (u64 *)keys_dev,
                             idx->num_cu, nullptr, s));
    if (ev)
        HIP_TRY(hipEventRecord(ev->second, s));
    if (indices_dev)
        HIP_TRY(knn_keys_unpack_launch((const u64 *)keys_dev, m, indices_dev, s));
    return KNN_OK;
}

int knn_debug_filter_scores(knn_index *idx, int m, const float *queries_dev, float *scores_dev,
                            float *qnorm_dev, double consts[8])
{
    if (!idx || m < 1 || !queries_dev || !scores_dev || !qnorm_dev || !consts)
        return fail(KNN_EINVAL, "knn_debug_filter_scores: bad arguments");
    std::lock_guard<std::recursive_mutex> lock(idx->mu);
    if (!idx->filter.usable)
        return fail(KNN_EINVAL, "knn_debug_filter_scores: this index has no filter layouts");
    DeviceGuard guard(idx->device);
    HIP_TRY(knn_filter_debug(idx->filter, m, queries_dev, idx->refs, scores_dev, nullptr, qnorm_dev, consts,
                             nullptr));
    return KNN_OK;
}

int knn_index_timing(knn_index *idx, int enable)
{
    if (!idx)
        return fail(KNN_EINVAL, "knn_index_timing: null index");
    std::lock_guard<std::recursive_mutex> lock(idx->mu);
    idx->timing = enable > 0 ? enable : 0;
    idx->timing_seq = 0;
    idx->events_used = 0;
    if (enable > 0) {
        // The event pairs are made HERE, not at the launches they bracket: the first bracketed launch of a run used to create
        // its pair on the spot — 0.2-0.3 ms of hipEventCreate inside the caller's timed region (a 20-step bench run read 0.141 ms
        // per step where the same 20 batches take 0.126 without the instrumentation; tools/fill_drain.py).
        DeviceGuard guard(idx->device);
        while (idx->events.size() < 64) {
            std::pair<hipEvent_t, hipEvent_t> fresh;
            HIP_TRY(hipEventCreate(&fresh.first));
            if (hipEventCreate(&fresh.second) != hipSuccess) {
                (void)hipEventDestroy(fresh.first);
                return fail(KNN_EHIP, "knn_index_timing: hipEventCreate failed");
            }
            idx->events.push_back(fresh);
        }
    }
    return KNN_OK;
}

int knn_index_timing_read(knn_index *idx, int *launches, double *total_ms)
{
    if (!idx || !launches || !total_ms)
        return fail(KNN_EINVAL, "knn_index_timing_read: bad arguments");
    std::lock_guard<std::recursive_mutex> lock(idx->mu);
    DeviceGuard guard(idx->device);
    double sum = 0.0;
    for (size_t i = 0; i < idx->events_used; ++i) {
        float ms = 0.0f;
        HIP_TRY(hipEventSynchronize(idx->events[i].second));
        HIP_TRY(hipEventElapsedTime(&ms, idx->events[i].first, idx->events[i].second));
        sum += ms;
    }
    *launches = (int)idx->events_used;
    *total_ms = sum;
    idx->events_used = 0;
    return KNN_OK;
}

int knn_index_last_stats(knn_index *idx, long long stats[4])
{
    if (!idx || !stats)
        return fail(KNN_EINVAL, "knn_index_last_stats: bad arguments");
    std::lock_guard<std::recursive_mutex> lock(idx->mu);
    const FilterWorkspace &w = idx->filter.ws[idx->last_slot];
    if ((idx->stats[0] == 2 || idx->stats[0] == 4) && w.ctl) {
        DeviceGuard guard(idx->device);
        unsigned ctl[KNN_CTL_WORDS];
        HIP_TRY(hipMemcpy(ctl, w.ctl_cur ? w.ctl_cur : w.ctl, sizeof ctl, hipMemcpyDeviceToHost));
        std::vector<unsigned> counts(w.nlists);
        if (!counts.empty())
            HIP_TRY(hipMemcpy(counts.data(), w.counts, counts.size() * sizeof(unsigned),
                              hipMemcpyDeviceToHost));
        long long records = 0;
        for (unsigned c : counts)
            records += c < w.slice ? c : w.slice;
        if (idx->stats[0] == 4)   // + the shared overflow area
            records += ctl[KNN_CTL_RECORDS] < w.ovf_cap ? ctl[KNN_CTL_RECORDS] : w.ovf_cap;
        idx->stats[1] = records;
        idx->stats[2] = ctl[KNN_CTL_FALLBACK] ? 1 : ctl[KNN_CTL_EXACT_CELLS] ? 2 : 0;   // 2: the batch's listed (cell, query) pairs were evaluated exactly
        idx->stats[3] = idx->filter.n_outliers;
    }
    memcpy(stats, idx->stats, sizeof idx->stats);
    return KNN_OK;
}

int knn_index_debug_counters(knn_index *idx, long long out[4])
{
    if (!idx || !out)
        return fail(KNN_EINVAL, "knn_index_debug_counters: bad arguments");
    std::lock_guard<std::recursive_mutex> lock(idx->mu);
    out[0] = out[1] = out[2] = out[3] = 0;
    const FilterWorkspace &w = idx->filter.ws[idx->last_slot];
    if (idx->stats[0] == 4 && w.ctl_cur) {
        DeviceGuard guard(idx->device);
        unsigned ctl[KNN_CTL_WORDS];
        HIP_TRY(hipMemcpy(ctl, w.ctl_cur, sizeof ctl, hipMemcpyDeviceToHost));
        out[0] = ctl[KNN_CTL_WIDE_SEEDS];
        out[1] = ctl[KNN_CTL_DENSE_CELLS];
        out[2] = idx->filter.cells ? idx->filter.cells->ncells : 0;
        out[3] = idx->filter.cells ? idx->filter.cells->max_cell_rows : 0;
    }
    return KNN_OK;
}

int knn_debug_scan_plan(int num_cu, int blocks_per_cu, unsigned nitems, int m, long long out[8])
{
    return knn_debug_scan_plan_ex(num_cu, blocks_per_cu, nitems, m, 0, out);
}

int knn_debug_scan_plan_ex(int num_cu, int blocks_per_cu, unsigned nitems, int m, int self_lists, long long out[8])
{
    if (!out || num_cu < 1 || blocks_per_cu < 1 || blocks_per_cu > 2 || m < 1 || m > KNN_CELL_BATCH)
        return fail(KNN_EINVAL, "knn_debug_scan_plan: bad arguments");
    const int m_padded = (m + 31) / 32 * 32;
    const CellScanPlan p = knn_cells_scan_plan(num_cu, blocks_per_cu, nitems, KNN_RECORD_CAPACITY, m_padded, self_lists != 0);
    out[0] = p.blocks;
    out[1] = p.nlists;
    out[2] = p.slice;
    out[3] = p.ovf_base;
    out[4] = p.ovf_cap;
    out[5] = (long long)p.lds_bytes;
    out[6] = KNN_RECORD_CAPACITY;
    out[7] = KNN_MAX_LISTS;
    return KNN_OK;
}

int knn_keys_to_indices(int device, const unsigned long long *keys_dev, int m, int *out_dev,
                        void *stream)
{
    if (m < 0 || (m > 0 && (!keys_dev || !out_dev)))
        return fail(KNN_EINVAL, "knn_keys_to_indices: bad arguments");
    DeviceGuard guard(device);
    if (!guard.ok)
        return fail(KNN_EHIP, "knn_keys_to_indices: hipSetDevice failed");
    HIP_TRY(knn_keys_unpack_launch((const u64 *)keys_dev, m, out_dev, (hipStream_t)stream));
    return KNN_OK;
}

int knn_keys_allreduce_min(int ndev, const int *devices, unsigned long long *const *keys_dev, int m,
                           void *const *streams)
{
    if (ndev < 1 || !devices || !keys_dev || m < 0)
        return fail(KNN_EINVAL, "knn_keys_allreduce_min: bad arguments");
    const int have = knn_device_count();
    for (int g = 0; g < ndev; ++g) {
        if (devices[g] < 0 || devices[g] >= have || (m > 0 && !keys_dev[g]))
            return fail(KNN_EINVAL, "knn_keys_allreduce_min: bad device or null key array");
        for (int h = 0; h < g; ++h)
            if (devices[h] == devices[g])
                return fail(KNN_EINVAL, "knn_keys_allreduce_min: a device appears twice (one key array per GPU)");
    }
    if (have < 1)
        return fail(KNN_ENODEV, "knn_keys_allreduce_min: no HIP device visible");
    std::string err;
    static_assert(sizeof(unsigned long long) == sizeof(u64), "key type");
    if (knn_rccl_allreduce_min(ndev, devices, (u64 *const *)keys_dev, m, (const hipStream_t *)streams, err) != 0)
        return fail(KNN_EHIP, "knn_keys_allreduce_min", err.c_str());
    return KNN_OK;
}

int knn_synth_fill_device(int device, float *dst_dev, long long count, unsigned long long seed,
                          long long first, void *stream)
{
    if (count < 0 || (count > 0 && !dst_dev))
        return fail(KNN_EINVAL, "knn_synth_fill_device: bad arguments");
    DeviceGuard guard(device);
    if (!guard.ok)
        return fail(KNN_EHIP, "knn_synth_fill_device: hipSetDevice failed");
    HIP_TRY(knn_synth_fill_launch(dst_dev, count, seed, first, (hipStream_t)stream));
    return KNN_OK;
}

}  // extern "C"

namespace {

// Queries on the host, packed keys back on the host; everything else stays on `idx->device`.
// keep_dev != nullptr: the keys stay on the device instead (*keep_dev = pooled buffer of m keys, complete
// on return; the caller gives it back with pool_put) — the RCCL merge of cudaCallback reads them there.
int query_keys_host(knn_index *idx, int m, const float *queries_host, u64 *keys_host, u64 **keep_dev = nullptr)
{
    DeviceGuard guard(idx->device);
    if (!guard.ok)
        return fail(KNN_EHIP, "query: hipSetDevice failed");
    float *q_dev = nullptr;
    u64 *keys_dev = nullptr;
    const size_t qbytes = (size_t)m * (size_t)idx->k * sizeof(float);
    const size_t kbytes = (size_t)m * sizeof(u64);
    int rc = KNN_OK;
    hipError_t e = pool_get(idx->device, qbytes, (void **)&q_dev);
    if (e == hipSuccess)
        e = pool_get(idx->device, kbytes, (void **)&keys_dev);
    if (e == hipSuccess)
        e = hipMemcpy(q_dev, queries_host, qbytes, hipMemcpyHostToDevice);
    if (e != hipSuccess)
        rc = fail(KNN_EHIP, "query: staging queries", hipGetErrorString(e));
    if (rc == KNN_OK)
        rc = knn_keys_init(idx->device, keys_dev, m, nullptr);
    if (rc == KNN_OK)
        rc = knn_index_query_keys(idx, m, q_dev, keys_dev, nullptr);
    if (rc == KNN_OK) {
        e = keep_dev ? hipStreamSynchronize(nullptr) : hipMemcpy(keys_host, keys_dev, kbytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess)
            rc = fail(KNN_EHIP, "query: D2H keys", hipGetErrorString(e));
    }
    if (rc != KNN_OK)
        (void)hipDeviceSynchronize();  // nothing in flight may still use the buffers going back to the pool
    pool_put(idx->device, q_dev, qbytes);
    if (keep_dev && rc == KNN_OK)
        *keep_dev = keys_dev;
    else
        pool_put(idx->device, keys_dev, kbytes);
    return rc;
}

}  // namespace

extern "C" int knn_index_query_host(knn_index *idx, int m, const float *queries_host, int *out_host)
{
    if (!idx || m < 0 || (m > 0 && (!queries_host || !out_host)))
        return fail(KNN_EINVAL, "knn_index_query_host: bad arguments");
    if (m == 0)
        return KNN_OK;
    std::vector<u64> keys((size_t)m, kKeyInit);
    if (idx->n > 0) {
        int rc = query_keys_host(idx, m, queries_host, keys.data());
        if (rc != KNN_OK)
            return rc;
    }
    for (int j = 0; j < m; ++j)
        out_host[j] = (int)(unsigned)(keys[(size_t)j] & 0xFFFFFFFFull);
    return KNN_OK;
}

namespace {

// One shard of a one-shot call, exact kernels only, with the scan hidden under the host-to-device
// copy: the rows go over in chunks on a copy stream and every chunk is scanned (min-folded into the
// same keys) on a compute stream as soon as it has landed.  The reference copies everything first
// (core.cu:885-891) and its own timings are dominated by that copy (README.md:291-292); for
// k = 16, m <= ~1300 the exact scan of a chunk takes less than its PCIe transfer, so the call
// costs the transfer plus one chunk's scan.
int run_shard_streamed(int device, int k, int m, long long rows, long long base, const float *queries_host,
                       const float *refs_host, u64 *keys_host, u64 **keep_dev = nullptr, int nchunks_hint = 8)
{
    DeviceGuard guard(device);
    if (!guard.ok)
        return fail(KNN_EHIP, "cudaCallback: hipSetDevice failed");
    hipDeviceProp_t prop;
    int num_cu = 256;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        num_cu = prop.multiProcessorCount;
    DeviceStreams st;
    HIP_TRY(streams_get(device, &st));
    const size_t row_bytes = (size_t)k * sizeof(float);
    const size_t rbytes = (size_t)rows * row_bytes, qbytes = (size_t)m * row_bytes, kbytes = (size_t)m * sizeof(u64);
    // Chunks of at least 32 MiB, multiples of 1024 rows.  Every pageable copy call costs ~0.25 ms of
    // pipeline fill on top of its bytes (profiles/r02_ingest_timing.txt) and the call ends one chunk's
    // scan after the last byte, so the count that minimises  0.25 ms * c + t_scan / c  is sqrt(t_scan / 0.25 ms)
    // (`nchunks_hint`, from the caller's cost model), kept within 2 .. 16.
    long long want_chunks = nchunks_hint < 2 ? 2 : nchunks_hint > 16 ? 16 : nchunks_hint;
    long long chunk_rows = (rows + want_chunks - 1) / want_chunks;
    const long long lo_rows = (long long)((32u << 20) / row_bytes) + 1;
    chunk_rows = std::max(lo_rows, chunk_rows);
    chunk_rows = (chunk_rows + 1023) / 1024 * 1024;
    const long long nchunks = (rows + chunk_rows - 1) / chunk_rows;

    float *r_dev = nullptr, *q_dev = nullptr;
    u64 *keys_dev = nullptr;
    std::vector<hipEvent_t> events;
    hipError_t e = pool_get(device, rbytes, (void **)&r_dev);
    if (e == hipSuccess)
        e = pool_get(device, qbytes, (void **)&q_dev);
    if (e == hipSuccess)
        e = pool_get(device, kbytes, (void **)&keys_dev);
    if (e == hipSuccess)
        e = hipMemcpyAsync(q_dev, queries_host, qbytes, hipMemcpyHostToDevice, st.compute);
    if (e == hipSuccess)
        e = knn_keys_fill_launch(keys_dev, m, st.compute);
    for (long long c = 0; c < nchunks && e == hipSuccess; ++c) {
        const long long r0 = c * chunk_rows, r1 = std::min(rows, r0 + chunk_rows);
        hipEvent_t ev = nullptr;
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess)
            break;
        events.push_back(ev);
        // pageable source: the call returns once the chunk is staged, while the GPU still scans the
        // previous one
        e = hipMemcpyAsync(r_dev + (size_t)r0 * k, refs_host + (size_t)r0 * k, (size_t)(r1 - r0) * row_bytes,
                           hipMemcpyHostToDevice, st.copy);
        if (e == hipSuccess)
            e = hipEventRecord(ev, st.copy);
        if (e == hipSuccess)
            e = hipStreamWaitEvent(st.compute, ev, 0);
        if (e == hipSuccess)
            e = knn_exact_launch(k, m, r1 - r0, base + r0, q_dev, r_dev + (size_t)r0 * k, keys_dev, num_cu, nullptr,
                                 st.compute);
    }
    if (e == hipSuccess && !keep_dev)
        e = hipMemcpyAsync(keys_host, keys_dev, kbytes, hipMemcpyDeviceToHost, st.compute);
    const hipError_t e_sync1 = hipStreamSynchronize(st.copy);
    const hipError_t e_sync2 = hipStreamSynchronize(st.compute);
    if (e == hipSuccess)
        e = e_sync1 != hipSuccess ? e_sync1 : e_sync2;
    for (hipEvent_t ev : events)
        (void)hipEventDestroy(ev);
    if (e != hipSuccess)
        (void)hipDeviceSynchronize();  // nothing in flight may still use the buffers going back to the pool
    pool_put(device, r_dev, rbytes);
    pool_put(device, q_dev, qbytes);
    if (keep_dev && e == hipSuccess)
        *keep_dev = keys_dev;
    else
        pool_put(device, keys_dev, kbytes);
    if (e != hipSuccess)
        return fail(KNN_EHIP, "cudaCallback: streamed shard", hipGetErrorString(e));
    return KNN_OK;
}

}  // namespace

namespace {

// Cost model of ONE shard of a one-shot call (host rows in, keys out) on one MI355X: which of the three ways serves it
// and what it is expected to take.  Constants measured on one box: pageable H2D 50-55 GB/s with ~0.25 ms of
// fixed cost per copy call; exact kernels 58e12 lane-ops/s with a compile-time K (or m < 48), 45e12 / 22e12 with a run-time
// k up to 64 / 128 (knn_exact_qregn), 5e12 on the row-per-lane kernels beyond; the MFMA filter ~3.4e-14 s per pair and
// K-step plus its layouts (built under the copy unless option `ingest` = 1); the grid index (k <= 4) four passes over the
// rows to build (~0.1 ms + 0.15 ns per row) and one wave per query.
struct ShardPlan {
    bool streamed = false;   // exact scan chunk by chunk under the copy (run_shard_streamed)
    int want_filter = 0;     // staged: build the MFMA layouts
    bool want_grid = false;  // staged: build the grid index (k <= 4)
    int nchunks = 2;         // streamed: copy calls
    double t_streamed = 0.0, t_staged = 0.0;
    double seconds = 0.0;    // the chosen way
};

ShardPlan plan_shard(int k, int m, long long rows)
{
    ShardPlan p;
    const double pairs = (double)m * (double)rows;
    const bool fast_k = k == 1 || k == 2 || k == 3 || k == 4 || k == 8 || k == 16;
    const double exact_rate = (fast_k || m < 48) ? 58e12 : k <= 64 ? 45e12 : k <= 128 ? 22e12 : 5e12;
    const double t_exact = (3.0 * k + 3.0) * pairs / exact_rate;
    const int kt = knn_kt_of(k);   // 0: no fp16 layouts for this k (k > 4096)
    const double t_filter = kt == 0 ? 1e30 : 1.0e-3 + 2.5 * 4.0 * k * (double)rows / 4e12 + 3.4e-14 * kt * pairs + 1e-4;
    const double t_filter_under_copy = 6e-4 + 3.4e-14 * kt * pairs + 1e-4;   // host rows: layouts built under the copy
    p.want_filter = g_opt_path == 2 ||
                    (g_opt_path == 0 && m >= 5 && kt != 0 && (g_opt_ingest == 1 ? t_filter : t_filter_under_copy) < t_exact);
    // The pruned scan for a one-shot call (round 5: the cell sort's bucket pass runs under the copy, knn_filter_build_cells_from_host):
    // ~1.0 ms per 2^24 rows stay behind the last byte (cell prefix + placement) where the plain layouts leave 0.07, and a
    // batch of 1024 queries then costs 0.04 ms + 6.5 ps per row instead of 3.4e-14 s per pair — it repays the sort from about
    // 2500 queries on at C3's size (measured: one batch 0.145 ms against 0.58).  Where the library's own policy puts
    // resident indexes on the cells (k <= 16, >= 2^19 / 2^20 rows), and only with the default build options.
    // (same fixed costs as the plain layouts' line above, so that the two differ by what differs)
    const double t_cells_query = 6e-4 + 1.0e-3 * (double)rows / 16777216.0 + ceil(m / 1024.0) * (4e-5 + 6.5e-12 * (double)rows) + 1e-4;
    if (p.want_filter && g_opt_path == 0 && g_opt_cells == 0 && g_opt_ingest == 0 && g_opt_cells_build == 0 && k <= 16 &&
        rows >= (k <= 12 ? (1ll << 19) : (1ll << 20)) && (double)rows * 96.0 <= 2147483648.0 && t_cells_query < t_filter_under_copy)
        p.want_filter = 2;
    // Third option: the exact scan chunk by chunk under the copy: costs the longer of the two plus one chunk's scan.  The
    // staged alternatives pay the copy in one piece; the filter layouts are then built under the copy's tail
    // (knn_filter_build_from_host), leaving the query itself.
    const double bytes = 4.0 * k * (double)rows;
    const double t_h2d = bytes / 52e9;
    const double kCopyCall = 2.5e-4;
    double nchunks = floor(sqrt(t_exact / kCopyCall) + 0.5);
    nchunks = std::max(2.0, std::min(16.0, std::min(nchunks, floor(bytes / (double)(32u << 20)))));
    p.nchunks = (int)nchunks;
    p.t_streamed = std::max(t_h2d, t_exact) + t_exact / nchunks + kCopyCall * nchunks + 1e-4;
    const double t_filter_query = 6e-4 + 3.4e-14 * kt * pairs + 1e-4;   // layouts under the copy + query
    const double t_query = p.want_filter == 2 ? t_cells_query : p.want_filter ? (g_opt_ingest == 1 ? t_filter : t_filter_query) : t_exact;
    const double t_grid = k <= 4 ? 1.5e-4 + 1.5e-10 * (double)rows + 2e-8 * m : 1e30;
    p.want_grid = k <= 4 && (g_opt_path == 3 || (g_opt_path == 0 && rows >= 65536 && t_grid < t_query));
    p.t_staged = t_h2d + kCopyCall + (p.want_grid ? t_grid : t_query);
    p.streamed = g_opt_path != 2 && !p.want_grid && g_opt_stream != 1 && bytes >= (double)(64u << 20) &&
                 (g_opt_stream == 2 || p.t_streamed < p.t_staged);
    p.seconds = p.streamed ? p.t_streamed : p.t_staged;
    return p;
}

// How many GPUs a one-shot call is split over (reference core.cu:865-872: every GPU, never more than n, ONE when
// n <= min(2^18, m << 10) — which is every case of the TA harness).  Here the reference's small-n rule stands,
// and beyond it the cost model decides: a call fans out over G GPUs when the shards' expected time (each GPU has its own
// PCIe link and copies its own range) plus what fanning out costs — a host thread, an index create / query / destroy and
// a share of the key exchange per GPU — is lowest; of the counts within 10 % of the best the smallest is taken.  The fan-out constants (0.2 ms + 0.02 ms per GPU)
// are estimates: no multi-GPU node was available to measure them.
int shard_policy(int k, int m, long long n, int ndev)
{
    long long cap = std::min<long long>(ndev, n);                           // core.cu:867-868
    if (cap <= 1 || n <= std::min<long long>(1ll << 18, (long long)m << 10))   // core.cu:871-872
        return 1;
    // candidates: 1, 2, 4, ... and every visible GPU; of those within 10 % of the best, the fewest
    std::vector<int> cand;
    for (int g = 1; g < (int)cap; g *= 2)
        cand.push_back(g);
    cand.push_back((int)cap);
    std::vector<double> t(cand.size(), 0.0);
    double t_min = 1e30;
    for (size_t i = 0; i < cand.size(); ++i) {
        const int g = cand[i];
        t[i] = plan_shard(k, m, (n + g - 1) / g).seconds + (g > 1 ? 2e-4 + 2e-5 * g : 0.0);
        t_min = std::min(t_min, t[i]);
    }
    size_t pick = 0;
    while (t[pick] > 1.1 * t_min)
        ++pick;
    const int best = cand[pick];
    return best;
}

}  // namespace

// Test hook (host arithmetic): how ONE shard of a one-shot call would be served — out = {filter layouts: 0 none, 1 plain, 2 cell-sorted
// (the pruned scan), exact scan streamed under the copy, grid index, copy calls of the streamed form}.
extern "C" int knn_debug_plan_shard(int k, int m, long long rows, long long out[4])
{
    if (k < 1 || m < 1 || rows < 1 || !out)
        return fail(KNN_EINVAL, "knn_debug_plan_shard: bad arguments");
    const ShardPlan p = plan_shard(k, m, rows);
    out[0] = p.want_filter;
    out[1] = p.streamed ? 1 : 0;
    out[2] = p.want_grid ? 1 : 0;
    out[3] = p.nchunks;
    return KNN_OK;
}

extern "C" int knn_debug_shard_policy(int k, int m, long long n, int ndev)
{
    if (k < 1 || m < 1 || n < 1 || ndev < 1)
        return fail(KNN_EINVAL, "knn_debug_shard_policy: bad arguments");
    return shard_policy(k, m, n, ndev);
}

// ---------------------------------------------------------------------------------------------
// The drop-in entry point (reference core.h:71, core.cu:1282-1297 -> v8, core.cu:856-958).
// ---------------------------------------------------------------------------------------------
extern "C" void cudaCallback(int k, int m, int n, float *searchPoints, float *referencePoints,
                             int **results)
{
    if (k < 1 || m < 1 || n < 1 || !searchPoints || !referencePoints || !results) {
        fail(KNN_EINVAL, "cudaCallback: k, m, n must be >= 1 and pointers non-null");
        die(__FILE__, __LINE__, KNN_EINVAL, g_err.c_str());
    }
    const int ndev = knn_device_count();
    if (ndev < 1) {
        // The reference computes on the CPU here (core.cu:869-870); this library has no CPU path.
        fail(KNN_ENODEV, "cudaCallback: no HIP device visible (this library has no CPU fallback)");
        die(__FILE__, __LINE__, KNN_ENODEV, g_err.c_str());
    }
    // core.cu:865-872: how many GPUs take part (never more shards than points; one GPU for the small sets of the TA
    // harness; beyond that by the cost model: shard_policy).  Option `shards` forces a count (test hook).
    long long shards = g_opt_shards > 0 ? (long long)g_opt_shards : (long long)shard_policy(k, m, n, ndev);
    if (shards > n)
        shards = n;
    g_last_shards = shards;
    g_last_cells = 0;
    static const bool trace = getenv("KNN_MI355X_TRACE_CALL") != nullptr;
    if (trace)
        fprintf(stderr, "[knn call] k %d m %d n %d: %d GPU(s) visible -> %lld shard(s)%s\n", k, m, n, ndev, shards,
                g_opt_shards > 0 ? " (option shards)" : "");
    // core.cu:875: thread_n = divup(n, num_gpus); the last shard takes what is left.  A shard
    // past the end is empty here (the reference patches it to one overlapping point,
    // core.cu:881-882; an empty shard gives the same minimum).
    const long long per = (n + shards - 1) / shards;

    std::vector<std::vector<u64>> shard_keys((size_t)shards);
    std::vector<int> shard_rc((size_t)shards, KNN_OK);
    std::vector<std::string> shard_err((size_t)shards);
    // The exchange step.  Several GPUs with one shard each (the reference's configuration, core.cu:873):
    // every shard leaves its m packed keys ON ITS DEVICE and one RCCL all-reduce(uint64, min) merges them
    // (knn_rccl.cpp); device 0's copy comes back to the host.  More shards than devices (the single-GPU
    // test hook), a single device, `rccl` = 2, or no usable librccl: the keys come back per shard and are
    // min-merged on the host — the same unsigned minimum.
    // ONE communicator set per process, for all visible devices (ncclCommInitAll takes seconds on an 8-GPU node): a call
    // that uses fewer GPUs than are visible merges on the host instead of creating a second set.
    std::string rccl_why;
    const bool rccl_wanted = g_opt_rccl != 2 && shards == ndev && (g_opt_rccl == 1 || shards > 1);
    const bool use_rccl = rccl_wanted && knn_rccl_available(&rccl_why) != 0;
    if (g_opt_rccl == 1 && shards > 1 && shards != ndev) {
        // (`rccl` = 1 means "always, and die when RCCL cannot serve" — a call that the policy or option `shards`
        // splits over another number of GPUs than are visible has no communicator set, and merging on the host in silence
        // would break that contract)
        char why[160];
        snprintf(why, sizeof why, "%lld shards on %d visible GPU(s): the one communicator set of the process spans all visible devices", shards, ndev);
        fail(KNN_EINVAL, "cudaCallback: option rccl = 1 but the call's shards are not the visible devices", why);
        die(__FILE__, __LINE__, KNN_EINVAL, g_err.c_str());
    }
    if (trace && shards > 1 && !use_rccl)
        fprintf(stderr, "[knn call] keys merged on the host: %s\n",
                g_opt_rccl == 2 ? "option rccl = 2" : shards != ndev ? "the shards are not the visible devices" : rccl_why.c_str());
    if (rccl_wanted && !use_rccl && g_opt_rccl == 1) {
        fail(KNN_EHIP, "cudaCallback: option rccl = 1 but RCCL cannot be used", rccl_why.c_str());
        die(__FILE__, __LINE__, KNN_EHIP, g_err.c_str());
    }
    std::vector<u64 *> shard_dev_keys((size_t)shards, nullptr);   // use_rccl: pooled, on device g % ndev

    auto run_shard = [&](long long g) {
        const long long lo = std::min<long long>(g * per, n);
        const long long hi = std::min<long long>(lo + per, n);
        std::vector<u64> &keys = shard_keys[(size_t)g];
        u64 **keep_dev = use_rccl ? &shard_dev_keys[(size_t)g] : nullptr;
        if (!use_rccl)
            keys.assign((size_t)m, kKeyInit);
        if (hi <= lo) {
            if (use_rccl) {   // an empty shard still takes part in the all-reduce: m keys of (+INF, 0)
                DeviceGuard guard((int)(g % ndev));
                u64 *kd = nullptr;
                hipError_t e = guard.ok ? pool_get((int)(g % ndev), (size_t)m * sizeof(u64), (void **)&kd) : hipErrorInvalidDevice;
                if (e == hipSuccess)
                    e = knn_keys_fill_launch(kd, m, nullptr);
                if (e == hipSuccess)
                    e = hipStreamSynchronize(nullptr);
                if (e != hipSuccess) {
                    shard_rc[(size_t)g] = fail(KNN_EHIP, "cudaCallback: empty shard keys", hipGetErrorString(e));
                    shard_err[(size_t)g] = g_err;
                    pool_put((int)(g % ndev), kd, (size_t)m * sizeof(u64));
                } else {
                    *keep_dev = kd;
                }
            }
            return;
        }
        knn_index *idx = nullptr;
        // which of the three ways serves this shard: plan_shard (cost model)
        const ShardPlan plan = plan_shard(k, m, hi - lo);
        const int want_filter = plan.want_filter;
        const bool want_grid = plan.want_grid, streamed = plan.streamed;
        const double t_streamed = plan.t_streamed, t_staged = plan.t_staged;
        const double nchunks = plan.nchunks;
        const auto t0 = std::chrono::steady_clock::now();
        if (streamed) {
            const int rc = run_shard_streamed((int)(g % ndev), k, m, hi - lo, lo, searchPoints,
                                              referencePoints + (size_t)lo * (size_t)k, keys.data(), keep_dev, (int)nchunks);
            shard_rc[(size_t)g] = rc;
            if (rc != KNN_OK)
                shard_err[(size_t)g] = g_err;
            if (trace)
                fprintf(stderr, "[knn call] shard %lld rows %lld streamed exact scan: %.3f ms (model: streamed %.3f, staged %.3f)\n",
                        g, hi - lo, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(),
                        t_streamed * 1e3, t_staged * 1e3);
            return;
        }
        int rc = index_create_impl(&idx, (int)(g % ndev), k, hi - lo,
                                   referencePoints + (size_t)lo * (size_t)k, 0, lo, nullptr, want_grid ? 0 : want_filter,
                                   want_grid ? 1 : 0);
        const auto t1 = std::chrono::steady_clock::now();
        if (rc == KNN_OK)
            rc = query_keys_host(idx, m, searchPoints, keys.data(), keep_dev);
        if (rc == KNN_OK && idx->stats[0] == 4)
            ++g_last_cells;
        const auto t2 = std::chrono::steady_clock::now();
        knn_index_destroy(idx);
        if (trace) {
            const auto t3 = std::chrono::steady_clock::now();
            auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
                return std::chrono::duration<double, std::milli>(b - a).count();
            };
            fprintf(stderr, "[knn call] shard %lld rows %lld filter %d: create (alloc + H2D + layouts) %.3f ms, "
                            "query (stage + scan + D2H) %.3f ms, destroy %.3f ms\n",
                    g, hi - lo, want_filter, ms(t0, t1), ms(t1, t2), ms(t2, t3));
        }
        shard_rc[(size_t)g] = rc;
        if (rc != KNN_OK)
            shard_err[(size_t)g] = g_err;
    };

    // One host thread per GPU (core.cu:873); shards that share a device run back to back.
    const int nthreads = (int)std::min<long long>(shards, ndev);
    if (nthreads <= 1) {
        for (long long g = 0; g < shards; ++g)
            run_shard(g);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; ++t)
            pool.emplace_back([&, t] {
                for (long long g = t; g < shards; g += nthreads)
                    run_shard(g);
            });
        for (auto &th : pool)
            th.join();
    }
    auto release_dev_keys = [&]() {
        for (long long g = 0; g < shards; ++g)
            if (shard_dev_keys[(size_t)g]) {
                DeviceGuard guard((int)(g % ndev));
                pool_put((int)(g % ndev), shard_dev_keys[(size_t)g], (size_t)m * sizeof(u64));
                shard_dev_keys[(size_t)g] = nullptr;
            }
    };
    for (long long g = 0; g < shards; ++g)
        if (shard_rc[(size_t)g] != KNN_OK) {
            release_dev_keys();
            die(__FILE__, __LINE__, shard_rc[(size_t)g], shard_err[(size_t)g].c_str());
        }

    int *out = (int *)malloc(sizeof(int) * (size_t)m);  // caller free()s (main.cu:98,175)
    if (!out) {
        release_dev_keys();
        die(__FILE__, __LINE__, KNN_ENOMEM, "malloc(results)");
    }
    bool merged_on_gpus = false;
    if (use_rccl) {
        // Final reduce on the GPUs: ncclAllReduce(uint64, min) over the shards' keys, one D2H from shard 0's device.
        // The reduction writes to scratch buffers, never to the keys: if RCCL fails at run time (communicator
        // creation on a node it does not like, a failed collective) and the option is 0 = "RCCL where possible", the
        // shards' keys are still what the scans left and the host merge below takes over; `rccl` = 1 insists.
        std::vector<int> devs((size_t)shards);
        for (long long g = 0; g < shards; ++g)
            devs[(size_t)g] = (int)(g % ndev);
        std::string err;
        std::vector<u64> merged((size_t)m);
        std::vector<u64 *> scratch((size_t)shards, nullptr);
        int rc = 0;
        for (long long g = 0; g < shards && rc == 0; ++g) {
            DeviceGuard guard(devs[(size_t)g]);
            const hipError_t e = guard.ok ? pool_get(devs[(size_t)g], (size_t)m * sizeof(u64), (void **)&scratch[(size_t)g])
                                          : hipErrorInvalidDevice;
            if (e != hipSuccess) {
                err = std::string("scratch keys for the reduction: ") + hipGetErrorString(e);
                rc = -1;
            }
        }
        // (test hook: a run-time RCCL failure, and — a one-GPU box can only reach this code with `rccl` = 1 — the
        // fallback of the automatic mode with it)
        const bool test_fail = getenv("KNN_MI355X_TEST_RCCL_FAIL") != nullptr;
        if (rc == 0 && test_fail) {
            err = "KNN_MI355X_TEST_RCCL_FAIL is set";
            rc = -1;
        }
        if (rc == 0)
            rc = knn_rccl_allreduce_min((int)shards, devs.data(), shard_dev_keys.data(), m, nullptr, err, scratch.data());
        if (rc == 0) {
            DeviceGuard guard(devs[0]);
            const hipError_t e = hipMemcpy(merged.data(), scratch[0], (size_t)m * sizeof(u64), hipMemcpyDeviceToHost);
            if (e != hipSuccess) {
                err = std::string("D2H of the reduced keys: ") + hipGetErrorString(e);
                rc = -1;
            }
        }
        for (long long g = 0; g < shards; ++g) {   // nothing may still be reading the buffers going back to the pool
            DeviceGuard guard(devs[(size_t)g]);
            (void)hipStreamSynchronize(nullptr);
            if (scratch[(size_t)g])
                pool_put(devs[(size_t)g], scratch[(size_t)g], (size_t)m * sizeof(u64));
        }
        if (rc != 0 && (g_opt_rccl == 0 || test_fail)) {
            // host merge of the (untouched) per-shard keys
            static bool warned = false;
            if (!warned) {
                warned = true;
                fprintf(stderr, "knn_mi355x: RCCL key reduction failed (%s); merging the shards' keys on the host\n", err.c_str());
            }
            (void)hipGetLastError();
            for (long long g = 0; g < shards && rc != 0; ++g) {
                DeviceGuard guard(devs[(size_t)g]);
                shard_keys[(size_t)g].assign((size_t)m, kKeyInit);
                const hipError_t e = hipMemcpy(shard_keys[(size_t)g].data(), shard_dev_keys[(size_t)g], (size_t)m * sizeof(u64),
                                               hipMemcpyDeviceToHost);
                if (e != hipSuccess) {
                    err += std::string("; D2H of shard keys: ") + hipGetErrorString(e);
                    rc = -2;
                }
            }
            if (rc == -1)
                rc = 1;   // keys are on the host: merge below
        }
        release_dev_keys();
        if (rc < 0) {
            free(out);
            fail(KNN_EHIP, "cudaCallback: RCCL key reduction", err.c_str());
            die(__FILE__, __LINE__, KNN_EHIP, g_err.c_str());
        }
        if (rc == 0) {
            merged_on_gpus = true;
            ++g_rccl_reductions;
            for (int j = 0; j < m; ++j)
                out[j] = (int)(unsigned)(merged[(size_t)j] & 0xFFFFFFFFull);
        }
    }
    if (!merged_on_gpus) {
        // Host merge: unsigned min of packed keys == lexicographic (distance, global index).
        for (int j = 0; j < m; ++j) {
            u64 best = kKeyInit;
            for (long long g = 0; g < shards; ++g)
                best = std::min(best, shard_keys[(size_t)g][(size_t)j]);
            out[j] = (int)(unsigned)(best & 0xFFFFFFFFull);
        }
    }
    *results = out;
}
