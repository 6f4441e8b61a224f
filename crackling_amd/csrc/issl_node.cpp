// Several GPUs of one node in one process: image replication (RCCL broadcast over xGMI, peer-copy fallback) and
// guide sharding with one host thread per device.  See include/issl_hip.h, "one process, several GPUs".
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <dlfcn.h>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "issl_device.hpp"

using namespace issl;

namespace {

double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// RCCL is loaded on demand so that libissl_hip.so has no link-time dependency on it (a process that already
// holds an RCCL -- e.g. PyTorch's -- is reused through the common SONAME).
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;

    bool load()
    {
        if (handle) return ok;
        const char *names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char *n : names) {
            handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
        }
        if (!handle) return false;
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(handle, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(handle, "ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(handle, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(handle, "ncclGroupEnd"));
        Broadcast = reinterpret_cast<decltype(Broadcast)>(dlsym(handle, "ncclBroadcast"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(handle, "ncclGetErrorString"));
        ok = CommInitAll && CommDestroy && GroupStart && GroupEnd && Broadcast && GetErrorString;
        return ok;
    }
};

Rccl g_rccl;

} // namespace

struct issl_node {
    issl_index *root = nullptr;            // caller's index, uploaded on devices[0]
    std::vector<int> devices;
    std::vector<void *> images;            // images[0] belongs to root; the others are owned here
    std::vector<issl_index *> replicas;    // replicas[0] == root
    std::vector<double> busy_ms;           // per device: time spent scoring in the last issl_node_score call
    std::vector<uint64_t> guides_done;     // per device: guides it scored in that call
    bool force_rccl = false, no_rccl = false; // ISSL_FORCE_RCCL / ISSL_NO_RCCL, read once at creation
    issl_node_info info{};
};

#define NODE_HIP_TRY(expr)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            set_error(std::string("HIP error: ") + hipGetErrorString(e_) + " at " #expr);          \
            return ISSL_E_DEVICE;                                                                  \
        }                                                                                          \
    } while (0)

// Move the image of devices[0] to every other device.  Returns ISSL_OK and sets *used_rccl.
static int broadcast_image(issl_node *nd, size_t bytes, int *used_rccl)
{
    const int n = static_cast<int>(nd->devices.size());
    *used_rccl = 0;
    if (n <= 1 && !nd->force_rccl) return ISSL_OK; // a single device has nothing to receive; ISSL_FORCE_RCCL=1 still runs the collective (test aid)
    std::set<int> distinct(nd->devices.begin(), nd->devices.end());
    const bool want_rccl = distinct.size() == nd->devices.size() && !nd->no_rccl;
    if (want_rccl && g_rccl.load()) {
        std::vector<ncclComm_t> comms(n);
        std::vector<hipStream_t> streams(n);
        ncclResult_t r = g_rccl.CommInitAll(comms.data(), n, nd->devices.data());
        if (r == ncclSuccess) {
            bool ok = true;
            int n_streams = 0;
            for (int i = 0; i < n && ok; ++i) {
                ok = hipSetDevice(nd->devices[i]) == hipSuccess && hipStreamCreate(&streams[i]) == hipSuccess;
                if (ok) n_streams = i + 1;
            }
            // in pieces of 1 GiB (a 300 M-site image is 61 GB): no collective takes a count beyond 2^31
            const size_t piece = size_t(1) << 30;
            for (size_t at = 0; at < bytes && ok; at += piece) {
                const size_t len = bytes - at < piece ? bytes - at : piece;
                g_rccl.GroupStart();
                for (int i = 0; i < n; ++i) {
                    (void)hipSetDevice(nd->devices[i]);
                    r = g_rccl.Broadcast(static_cast<const uint8_t *>(nd->images[0]) + at, static_cast<uint8_t *>(nd->images[i]) + at, len,
                                         ncclUint8, 0, comms[i], streams[i]);
                    if (r != ncclSuccess) ok = false;
                }
                r = g_rccl.GroupEnd();
                if (r != ncclSuccess) ok = false;
            }
            for (int i = 0; i < n_streams; ++i) { // (also after a failure: nothing stays in flight, no stream leaks)
                (void)hipSetDevice(nd->devices[i]);
                if (hipStreamSynchronize(streams[i]) != hipSuccess) ok = false;
                (void)hipStreamDestroy(streams[i]);
            }
            for (int i = 0; i < n; ++i) g_rccl.CommDestroy(comms[i]);
            if (ok) {
                *used_rccl = 1;
                return ISSL_OK;
            }
        }
        // fall through to peer copies; the message is informational only
    }
    for (int i = 1; i < n; ++i) {
        NODE_HIP_TRY(hipMemcpyPeer(nd->images[i], nd->devices[i], nd->images[0], nd->devices[0], bytes));
    }
    NODE_HIP_TRY(hipSetDevice(nd->devices[0]));
    NODE_HIP_TRY(hipDeviceSynchronize());
    return ISSL_OK;
}

extern "C" {

int issl_node_close(issl_node *nd)
{
    if (!nd) return ISSL_OK;
    for (size_t i = 1; i < nd->replicas.size(); ++i)
        if (nd->replicas[i]) issl_index_close(nd->replicas[i]);
    for (size_t i = 1; i < nd->images.size(); ++i) {
        if (nd->images[i]) {
            (void)hipSetDevice(nd->devices[i]);
            (void)hipFree(nd->images[i]);
        }
    }
    delete nd;
    return ISSL_OK;
}

int issl_node_create(issl_index *idx, const int *devices, int n_devices, issl_node **out)
{
    if (!idx || !out) { set_error("null argument"); return ISSL_E_ARG; }
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) {
        set_error("no HIP device available: the ISSL scorer has no CPU fallback");
        return ISSL_E_DEVICE;
    }
    issl_node *nd = new (std::nothrow) issl_node();
    if (!nd) { set_error("out of memory"); return ISSL_E_NOMEM; }
    if (!devices || n_devices <= 0) {
        for (int d = 0; d < visible; ++d) nd->devices.push_back(d);
    } else {
        for (int i = 0; i < n_devices; ++i) {
            if (devices[i] < 0 || devices[i] >= visible) {
                delete nd;
                set_error("device " + std::to_string(devices[i]) + " out of range (" + std::to_string(visible) + " visible)");
                return ISSL_E_ARG;
            }
            nd->devices.push_back(devices[i]);
        }
    }
    const int n = static_cast<int>(nd->devices.size());
    nd->root = idx;
    nd->info.n_devices = n;
    const char *force = std::getenv("ISSL_FORCE_RCCL"), *no = std::getenv("ISSL_NO_RCCL");
    nd->force_rccl = force && force[0] == '1';
    nd->no_rccl = no && no[0] == '1';
    nd->busy_ms.assign(n, 0.0);
    nd->guides_done.assign(n, 0);
    double t0 = now_ms();
    int rc = issl_index_upload(idx, nd->devices[0]);
    if (rc) { delete nd; return rc; }
    nd->info.ms_upload = now_ms() - t0;
    void *img0 = nullptr;
    size_t bytes = 0;
    rc = issl_index_image(idx, &img0, &bytes);
    if (rc) { delete nd; return rc; }
    nd->images.assign(n, nullptr);
    nd->replicas.assign(n, nullptr);
    nd->images[0] = img0;
    nd->replicas[0] = idx;
    for (int i = 1; i < n; ++i) {
        if (hipSetDevice(nd->devices[i]) != hipSuccess || hipMalloc(&nd->images[i], bytes) != hipSuccess) {
            set_error("cannot allocate the index image on device " + std::to_string(nd->devices[i]));
            issl_node_close(nd);
            return ISSL_E_DEVICE;
        }
    }
    t0 = now_ms();
    int used = 0;
    rc = broadcast_image(nd, bytes, &used);
    if (rc) { issl_node_close(nd); return rc; }
    nd->info.ms_broadcast = now_ms() - t0;
    nd->info.used_rccl = used;
    // An image whose cold sections (site table, slice lists) live in pinned host memory is replicated hot part only:
    // every device reads the ONE host copy (BASELINE configs[4]: index larger than the HBM).
    void *cold = nullptr;
    size_t cold_bytes = 0;
    rc = issl_index_cold(idx, &cold, &cold_bytes);
    if (rc) { issl_node_close(nd); return rc; }
    for (int i = 1; i < n; ++i) {
        rc = cold ? issl_index_attach_image_cold(nd->devices[i], nd->images[i], bytes, cold, cold_bytes, &nd->replicas[i])
                  : issl_index_attach_image(nd->devices[i], nd->images[i], bytes, &nd->replicas[i]);
        if (rc) { issl_node_close(nd); return rc; }
    }
    *out = nd;
    return ISSL_OK;
}

int issl_node_score(issl_node *nd, const uint64_t *guides, size_t n, int max_dist, double threshold, int method,
                    double *mit, double *cfd)
{
    if (!nd || (n && (!guides || !mit || !cfd))) { set_error("null argument"); return ISSL_E_ARG; }
    const size_t world = nd->replicas.size();
    const double t0 = now_ms();
    std::vector<int> rcs(world, ISSL_OK);
    std::vector<std::string> errs(world);
    // The reference cuts its guide loop statically (OpenMP, isslScoreOfftargets.cpp:316); Crackling emits guides in
    // genome order, so contiguous eighths would give the GPU with the repeat-dense region the longest shard.  The
    // batch is a queue of chunks instead: every device thread takes the next one when it is done with its last
    // (scores land at the chunk's place in the caller's arrays, so input order is kept).
    const size_t chunk = std::min<size_t>(262144, std::max<size_t>(16384, (n + world * 8 - 1) / (world * 8)));
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};
    auto work = [&](size_t r) {
        nd->busy_ms[r] = 0.0;
        nd->guides_done[r] = 0;
        while (!failed.load(std::memory_order_relaxed)) {
            const size_t lo = next.fetch_add(chunk);
            if (lo >= n) break;
            const size_t cnt = std::min(chunk, n - lo);
            const double t1 = now_ms();
            rcs[r] = issl_score(nd->replicas[r], guides + lo, cnt, max_dist, threshold, method, mit + lo, cfd + lo);
            nd->busy_ms[r] += now_ms() - t1;
            nd->guides_done[r] += cnt;
            if (rcs[r]) {
                errs[r] = issl_last_error();
                failed.store(true);
                break;
            }
        }
    };
    std::vector<std::thread> pool;
    for (size_t r = 1; r < world; ++r) pool.emplace_back(work, r);
    work(0);
    for (auto &t : pool) t.join();
    nd->info.ms_last_score = now_ms() - t0;
    for (size_t r = 0; r < world; ++r) {
        if (rcs[r]) {
            set_error("device " + std::to_string(nd->devices[r]) + ": " + errs[r]);
            return rcs[r];
        }
    }
    return ISSL_OK;
}

int issl_node_shard_times(const issl_node *nd, double *busy_ms, uint64_t *guides, int n)
{
    if (!nd || !busy_ms || !guides) { set_error("null argument"); return ISSL_E_ARG; }
    if (n < static_cast<int>(nd->replicas.size())) { set_error("buffer too small"); return ISSL_E_ARG; }
    for (size_t r = 0; r < nd->replicas.size(); ++r) { busy_ms[r] = nd->busy_ms[r]; guides[r] = nd->guides_done[r]; }
    return ISSL_OK;
}

int issl_node_get_info(const issl_node *nd, issl_node_info *out)
{
    if (!nd || !out) { set_error("null argument"); return ISSL_E_ARG; }
    *out = nd->info;
    return ISSL_OK;
}

} // extern "C"
