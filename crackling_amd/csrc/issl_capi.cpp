// extern "C" surface of libissl_hip.so (include/issl_hip.h): index handles, HBM image management,
// scoring pipeline orchestration.  Compiled by hipcc as host code; kernels are in issl_kernels.hip.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

#include "issl_device.hpp"

using namespace issl;

namespace issl {

static uint64_t align256(uint64_t x) { return (x + 255ull) & ~255ull; }

void layout_image(ImageHeader &h, const Geometry &g, uint64_t n_scores_unique, uint64_t n_tiles, bool dense_mit,
                  const LayoutSpec &spec)
{
    std::memset(&h, 0, sizeof h);
    h.magic = kImageMagic;
    h.version = kImageVersion;
    h.kind = 0;
    h.n_sites = g.n_sites;
    h.seq_len = g.seq_len;
    h.n_lines = g.n_lines;
    h.slice_width = g.slice_width;
    h.n_slices = g.n_slices;
    h.n_scores_file = g.n_scores;
    h.n_buckets = g.n_buckets();
    h.n_scores_unique = n_scores_unique;
    h.n_tiles = n_tiles;
    h.tile_cands = kTileCands;
    const bool no_lists = spec.no_lists && spec.sorted != 0 && spec.cold == 0;
    const uint64_t sites_b = align256(8 * g.n_sites), lists_b = no_lists ? 0 : align256(8 * g.n_sites * g.n_slices);
    h.lists_absent = no_lists ? 1 : 0;
    const bool esig = spec.inline_sigs && spec.cold == 0 && spec.sorted == 0; // in-list signatures: list-order layouts in HBM
    uint64_t off = kHeaderBytes, cold_off = 0;
    h.off_bucket_start = off; off = align256(off + 8 * (h.n_buckets + 1));
    h.off_tile_first = off;   off = align256(off + 4 * (h.n_buckets + 1));
    h.off_score_mask = off;   off = align256(off + 8 * n_scores_unique);
    h.off_score_val = off;    off = align256(off + 8 * n_scores_unique);
    if (dense_mit) { h.off_mit_dense = off; off = align256(off + 8ull * (1u << 20)); }
    h.cold_on_host = spec.cold;
    // the cold sections form a buffer of their own (pinned host memory): site table first, then the slice lists
    if (spec.cold & 2u) { h.off_sites = cold_off; cold_off += sites_b; } else { h.off_sites = off; off += sites_b; }
    if (spec.cold & 1u) { h.off_entries = cold_off; cold_off += lists_b; } else { h.off_entries = off; off += lists_b; }
    h.cold_bytes = spec.cold ? cold_off : sites_b + lists_b + (esig ? lists_b : 0);
    h.off_scan = off;         off = align256(off + 4ull * kTileCands * n_tiles);
    if (esig) { h.off_esig = off; off += lists_b; }
    if (spec.cold == 3u) { h.off_occ8 = off; off = align256(off + g.n_sites * g.n_slices); }
    if (spec.sorted) {
        h.off_sub_start = off; off = align256(off + 4 * h.n_buckets * 257);
        if (spec.sorted == 1) { h.off_srec = off; off = align256(off + sizeof(StreamRec) * kTileCands * n_tiles); }
        else                  { h.off_sid = off;  off = align256(off + 4ull * kTileCands * n_tiles); }
        h.off_site_occ = off; off = align256(off + 4 * g.n_sites);
    }
    h.total_bytes = off;
}

ImageView make_view(const ImageHeader &h, void *base, void *cold)
{
    uint8_t *p = static_cast<uint8_t *>(base);
    uint8_t *c = static_cast<uint8_t *>(cold);
    ImageView v;
    v.bucket_start = reinterpret_cast<const uint64_t *>(p + h.off_bucket_start);
    v.tile_first = reinterpret_cast<const uint32_t *>(p + h.off_tile_first);
    v.score_mask = reinterpret_cast<const uint64_t *>(p + h.off_score_mask);
    v.score_val = reinterpret_cast<const double *>(p + h.off_score_val);
    v.mit_dense = h.off_mit_dense ? reinterpret_cast<const double *>(p + h.off_mit_dense) : nullptr;
    v.sites = reinterpret_cast<const uint64_t *>(((h.cold_on_host & 2u) ? c : p) + h.off_sites);
    v.entries = h.lists_absent ? nullptr : reinterpret_cast<const uint64_t *>(((h.cold_on_host & 1u) ? c : p) + h.off_entries);
    v.esig = h.off_esig ? reinterpret_cast<const uint64_t *>(p + h.off_esig) : nullptr;
    v.occ8 = h.off_occ8 ? reinterpret_cast<const uint8_t *>(p + h.off_occ8) : nullptr;
    v.sub_start = h.off_sub_start ? reinterpret_cast<const uint32_t *>(p + h.off_sub_start) : nullptr;
    v.srec = h.off_srec ? reinterpret_cast<const StreamRec *>(p + h.off_srec) : nullptr;
    v.sid = h.off_sid ? reinterpret_cast<const uint32_t *>(p + h.off_sid) : nullptr;
    v.site_occ = h.off_site_occ ? reinterpret_cast<const uint32_t *>(p + h.off_site_occ) : nullptr;
    v.scan = reinterpret_cast<const uint32_t *>(p + h.off_scan);
    v.n_sites = h.n_sites;
    v.n_buckets = static_cast<uint32_t>(h.n_buckets);
    v.n_scores = static_cast<uint32_t>(h.n_scores_unique);
    v.slice_width = static_cast<uint32_t>(h.slice_width);
    v.n_slices = static_cast<uint32_t>(h.n_slices);
    v.n_tiles = static_cast<uint32_t>(h.n_tiles);
    return v;
}

static bool env_flag(const char *name) { const char *e = std::getenv(name); return e && e[0] == '1'; }

Tuning Tuning::from_env()
{
    Tuning t;
    t.scan_blocks = kScanGridBlocks;
    t.scan_threads = 1024;
    t.upload_chunk_kib = 16384;
    t.upload_ring_min_kib = 65536;
    t.upload_threads = 8;
    t.item_guides = kItemGuides;
    t.scan_generic = false;
    t.stage_timing = false;
    t.scan_events = 2;
    t.upload_timing = env_flag("ISSL_UPLOAD_TIMING");
    t.raw_chunks = 0;
    t.inline_sigs = -1;
    t.host_cold = -1;
    t.keep_lists = -1;
    t.sorted_layout = -1;
    t.compact = -1;
    t.prune = -1;
    t.tail_shapes = 1;
    t.hit_slots = 1;
    t.lean_tail = 1;
    t.small_bin = 1;
    t.expect_guides = 0;
    t.fine_items = 0;
    t.lanes = 1;
    static const char *const keys[][2] = {
        {"ISSL_SCAN_BLOCKS", "scan_blocks"}, {"ISSL_SCAN_THREADS", "scan_threads"}, {"ISSL_UPLOAD_CHUNK_KIB", "upload_chunk_kib"}, {"ISSL_UPLOAD_RING_MIN_KIB", "upload_ring_min_kib"}, {"ISSL_UPLOAD_THREADS", "upload_threads"}, {"ISSL_ITEM_GUIDES", "item_guides"},
        {"ISSL_SCAN_GENERIC", "scan_generic"}, {"ISSL_STAGE_TIMING", "stage_timing"}, {"ISSL_SCAN_EVENTS", "scan_events"}, {"ISSL_RAW_CHUNKS", "raw_chunks"},
        {"ISSL_INLINE_SIGS", "inline_sigs"}, {"ISSL_FORCE_HOST_COLD", "host_cold"}, {"ISSL_SCAN_STAMPS", "scan_stamps"},
        {"ISSL_SORTED_LAYOUT", "sorted_layout"}, {"ISSL_PRUNE", "prune"}, {"ISSL_LANES", "lanes"},
        {"ISSL_COMPACT", "compact"}, {"ISSL_TAIL_SHAPES", "tail_shapes"}, {"ISSL_HIT_SLOTS", "hit_slots"}, {"ISSL_LEAN_TAIL", "lean_tail"}, {"ISSL_SMALL_BIN", "small_bin"}, {"ISSL_EXPECT_GUIDES", "expect_guides"}, {"ISSL_FINE_ITEMS", "fine_items"},
        {"ISSL_KEEP_LISTS", "keep_lists"},
    };
    for (const auto &k : keys)
        if (const char *e = std::getenv(k[0])) (void)t.set(k[1], e); // values out of range leave the default
    return t;
}

bool Tuning::set(const char *key, const char *value)
{
    if (!key || !value) return false;
    const std::string k(key);
    char *end = nullptr;
    const long long n = std::strtoll(value, &end, 10);
    const bool is_int = end != value && *end == 0;
    if (k == "upload_chunk_kib") { if (!is_int || n < 4 || n > (1 << 20)) return false; upload_chunk_kib = static_cast<size_t>(n); }
    else if (k == "upload_ring_min_kib") { if (!is_int || n < 0) return false; upload_ring_min_kib = static_cast<size_t>(n); }
    else if (k == "upload_threads") { if (!is_int || n < 1 || n > 32) return false; upload_threads = static_cast<int>(n); }
    else if (k == "scan_threads") { if (!is_int || n < 64 || n > 1024 || n % 64) return false; scan_threads = static_cast<uint32_t>(n); }
    else if (k == "scan_blocks") { if (!is_int || n < 1 || n > static_cast<long long>(kScanMaxBlocks)) return false; scan_blocks = static_cast<uint32_t>(n); }
    else if (k == "item_guides") { if (!is_int || n < 8 || n > static_cast<long long>(kItemGuides)) return false; item_guides = static_cast<uint32_t>(n) & ~7u; }
    else if (k == "scan_generic") { if (!is_int || (n != 0 && n != 1)) return false; scan_generic = n == 1; }
    else if (k == "stage_timing") { if (!is_int || (n != 0 && n != 1)) return false; stage_timing = n == 1; }
    else if (k == "scan_events") { if (!is_int || n < 0 || n > 2) return false; scan_events = static_cast<int>(n); }
    else if (k == "raw_chunks") { if (!is_int || n < 0) return false; raw_chunks = static_cast<size_t>(n); }
    else if (k == "inline_sigs") { if (!is_int || n < -1 || n > 1) return false; inline_sigs = static_cast<int>(n); }
    else if (k == "host_cold") { if (!is_int || n < -1 || n > 1) return false; host_cold = static_cast<int>(n); }
    else if (k == "keep_lists") { if (!is_int || n < -1 || n > 1) return false; keep_lists = static_cast<int>(n); }
    else if (k == "sorted_layout") { if (!is_int || n < -1 || n > 1) return false; sorted_layout = static_cast<int>(n); }
    else if (k == "compact") { if (!is_int || n < -1 || n > 1) return false; compact = static_cast<int>(n); }
    else if (k == "prune") { if (!is_int || n < -1 || n > 1) return false; prune = static_cast<int>(n); }
    else if (k == "lanes") { if (!is_int || n < 1 || n > 3) return false; lanes = static_cast<int>(n); }
    else if (k == "tail_shapes") { if (!is_int || n < 0 || n > 1) return false; tail_shapes = static_cast<int>(n); }
    else if (k == "lean_tail") { if (!is_int || n < 0 || n > 1) return false; lean_tail = static_cast<int>(n); }
    else if (k == "small_bin") { if (!is_int || n < 0 || n > 1) return false; small_bin = static_cast<int>(n); }
    else if (k == "expect_guides") { if (!is_int || n < 0) return false; expect_guides = static_cast<size_t>(n); }
    else if (k == "fine_items") { if (!is_int || n < 0) return false; fine_items = static_cast<size_t>(n); }
    else if (k == "hit_slots") { if (!is_int || n < 0 || n > 2) return false; hit_slots = static_cast<int>(n); }
    else if (k == "scan_stamps") stamps_path = value;
    else return false;
    return true;
}

} // namespace issl

constexpr uint32_t kRing = 64;
constexpr size_t kMaxBatch = size_t(1) << 24; // guides per pipeline launch
constexpr size_t kSlotBytesMax = size_t(8) << 30; // hit slots of a workspace: up to 8 GiB (512 k guides per batch)

// One complete workspace + the internal stream that asynchronous batches run on.  (Rotating consecutive batches through
// several of these so that the short kernels of one batch run in the shadow of the next scan was measured in round 1
// -- 0.54 ms per step against 0.50 -- and removed: a scan fills every wave slot of the chip)
struct Lane {
    Workspace ws;
    hipEvent_t ev[6] = {};      // stage boundaries of the last batch
    hipEvent_t done = nullptr;  // end of the last batch
    hipStream_t stream = nullptr; // internal stream of asynchronous batches
    hipStream_t tail_stream = nullptr; // lanes = 2: the batch's verify / group / replay run here (high priority), beside the
                                // scan of the next batch on the other lane's stream
    bool ready = false;         // events and stream created
    uint32_t last_n = 0;
    uint32_t pending = 0;       // batches enqueued since the last finish
    bool staged = true;         // the last batch recorded its stage events
    bool done_recorded = false; // `done` stands behind the lane's last batch
    hipStream_t last_tail = nullptr; // the stream that batch's last kernels went to
    int last_max_dist = 0;
    uint32_t last_prune = 0;
    bool lean = false;          // the lane's finished batches had no guide beyond its hit slots: the next ones are enqueued
                                // without the grouping pass and the many-hit replays (Workspace::lean_tail)
};

struct issl_index {
    uint64_t worst_per_guide = 0;    // sum over the slices of their longest bucket: what one guide can be compared with at most (issl_score)
    std::unique_ptr<HostIndex> host; // absent for attached images
    Geometry geo;
    std::vector<uint64_t> bucket_sizes;
    Tuning tuning = Tuning::from_env(); // the environment is read here, once per handle
    // device state
    int device = -1;
    void *d_image = nullptr;
    bool owns_image = false;
    void *h_cold = nullptr;   // pinned host buffer of the cold sections (hdr.cold_on_host), else null
    void *d_cold = nullptr;   // the same buffer as the device addresses it
    bool owns_cold = false;
    ImageHeader hdr{};
    ImageView view{};
    Lane lane;                // workspace + stream of the synchronous entry points and of every other asynchronous batch
    Lane lane2;               // ... and of the batches in between (lanes option = 2)
    uint32_t n_async = 0;     // asynchronous batches enqueued so far: picks the lane
    Lane *last_lane = nullptr; // lane of the most recent batch (whose counters issl_last_stats reports)
    hipEvent_t ring[2 * kRing] = {}; // scan begin/end of the batches enqueued since the last finish (of those that recorded them: n_ring)
    uint32_t n_ring = 0;
    bool have_events = false;
    issl_stats stats{};
    uint32_t n_pending = 0;  // batches enqueued and not yet finished
    hipEvent_t prev_scan_end = nullptr; // lanes = 2: end of the previous batch's scan (scans run one after the other)
    hipEvent_t prev_batch_end = nullptr; // lanes = 3: end of the previous batch (its scan starts when that batch is through)
    bool list_order_only = false; // the lists of this index cannot be re-ordered (kSortNeedsListOrder)
    // issl_score: the largest piece that went through at once (no grow-and-rerun round) with record buffers of at least
    // proven_chunks chunks at a max_dist of at least proven_dist: pieces within that skip the per-guide estimate (0.9 ms per 500 k guides)
    size_t proven_guides = 0, proven_chunks = 0;
    int proven_dist = -1;
};

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            set_error(std::string("HIP error: ") + hipGetErrorString(e_) + " at " #expr);          \
            return ISSL_E_DEVICE;                                                                  \
        }                                                                                          \
    } while (0)

// 20 bp sequences cut into slices of 8, 4 or 2 bits (5, 10 or 20 slices): what isslCreateIndex can write correctly (it
// keeps slice values in a uint8_t, isslCreateIndex.cpp:228, and 40 bits only divide into whole positions for these
// widths).  Width 8 -- the README's recommendation, Crackling's default -- gets the sorted layouts and the pruned scan;
// the narrower ones the list-order layouts in HBM and the scan of whole buckets (the reference's own loop, :330-344).
static int supported_geometry(const Geometry &g)
{
    if (g.seq_len == 20 && (g.slice_width == 8 || g.slice_width == 4 || g.slice_width == 2) && g.n_slices * g.slice_width == 40) return ISSL_OK;
    set_error("unsupported index geometry: the gfx950 scan kernels implement 20 bp sequences in slices of 8, 4 or 2 bits "
              "(got seq_len=" + std::to_string(g.seq_len) + " slice_width=" +
              std::to_string(g.slice_width) + " slices=" + std::to_string(g.n_slices) + ")");
    return ISSL_E_UNSUPPORTED;
}

static int select_device(int device)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error("no HIP device available: the ISSL scorer has no CPU fallback");
        return ISSL_E_DEVICE;
    }
    if (device < 0 || device >= count) {
        set_error("device " + std::to_string(device) + " out of range (" + std::to_string(count) + " visible)");
        return ISSL_E_ARG;
    }
    HIP_TRY(hipSetDevice(device));
    return ISSL_OK;
}

static void free_workspace(Workspace &w)
{
    void *ptrs[] = {w.ng, w.gfill, w.gstart, w.gword, w.gidx, w.gbucket, w.items, w.plan, w.range_start, w.counters, w.scan_count, w.scan_span, w.sticky, w.stamps, w.gcur_big, w.gcur_big2, w.terms, w.sorted, w.gcount,
                    w.goff, w.blocksum, w.d_guides, w.d_mit, w.d_cfd, w.d_kept, w.d_hitrec, w.pay, w.rank, w.fword, w.fmeta,
                    w.fitems, w.fcount, w.fcount0, w.fsum, w.slots};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (w.raw) (void)hipFree(w.raw);
    if (w.raw_used) (void)hipFree(w.raw_used);
    if (w.h_stage) (void)hipHostFree(w.h_stage);
    w = Workspace{};
}

// Pinned host memory for the host-pointer entry point: a copy from pageable memory is staged by the runtime piece by piece
// (three of them cost 0.5 ms per 100 k guides); from here it is one DMA each.  No pinned memory: the plain copies do.
static bool ensure_stage(Workspace &w, size_t bytes)
{
    if (w.h_stage_bytes >= bytes) return true;
    if (w.h_stage) (void)hipHostFree(w.h_stage);
    w.h_stage = nullptr;
    w.h_stage_bytes = 0;
    const size_t want = bytes + bytes / 4;
    if (hipHostMalloc(&w.h_stage, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); w.h_stage = nullptr; return false; }
    w.h_stage_bytes = want;
    return true;
}

template <typename T> static int dev_alloc(T *&p, size_t count)
{
    if (p) {
        (void)hipFree(p);
        p = nullptr;
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p), std::max<size_t>(count, 1) * sizeof(T)));
    return ISSL_OK;
}

static int ensure_hit_capacity(Workspace &w, size_t want)
{
    if (want <= w.cap_hits) return ISSL_OK;
    int rc;
    if ((rc = dev_alloc(w.sorted, want))) return rc;
    if ((rc = dev_alloc(w.terms, 2 * want))) return rc;
    if ((rc = dev_alloc(w.pay, 2 * want))) return rc;
    if ((rc = dev_alloc(w.rank, want))) return rc;
    w.cap_hits = want;
    return ISSL_OK;
}

static int ensure_raw_capacity(Workspace &w, size_t chunks)
{
    if (chunks <= w.cap_chunks) return ISSL_OK;
    int rc = dev_alloc(w.raw, (chunks + 1) * kChunkRecs); // +1: spare chunk that absorbs writes after exhaustion
    if (rc) return rc;
    if ((rc = dev_alloc(w.raw_used, chunks + 1))) return rc;
    w.cap_chunks = chunks;
    return ISSL_OK;
}

static uint32_t scan_waves(const Tuning &tn) { return tn.scan_blocks * 16u; }

static int ensure_workspace(issl_index *ix, size_t n, Lane &lane, uint32_t fine_ways = kFineWays)
{
    Workspace &w = lane.ws;
    const Tuning &tn = ix->tuning;
    const size_t nb = ix->hdr.n_buckets;
    int rc;
    if (w.n_buckets != nb) {
        if ((rc = dev_alloc(w.ng, nb))) return rc;
        if ((rc = dev_alloc(w.gfill, nb))) return rc;
        HIP_TRY(hipMemset(w.ng, 0, 4 * nb)); // k_plan leaves both zeroed for the next batch
        HIP_TRY(hipMemset(w.gfill, 0, 4 * nb));
        if ((rc = dev_alloc(w.gstart, nb + 1))) return rc;
        if ((rc = dev_alloc(w.counters, 1))) return rc;
        if ((rc = dev_alloc(w.plan, 1))) return rc;
        if ((rc = dev_alloc(w.range_start, kMaxRanges + 2))) return rc;
        if ((rc = dev_alloc(w.scan_count, kScanMaxBlocks))) return rc;
        if ((rc = dev_alloc(w.scan_span, 2 * kSpanRing))) return rc;
        HIP_TRY(hipMemset(w.scan_span, 0, 16 * kSpanRing));
        HIP_TRY(hipMemset(w.scan_count, 0, 8 * kScanMaxBlocks));
        w.n_buckets = static_cast<uint32_t>(nb);
    }
    bool grew = false;
    // hit_slots = 2 (tests, A/B): wide hit slots from the first batch on, not only once a batch has shown that it needs them
    const bool force_wide = tn.hit_slots == 2 && w.slot_width != kSlotHitsWide &&
                            std::max<size_t>(std::max<size_t>(n, w.cap_guides), 1024) * kSlotHitsWide * sizeof(SlotRec) <= kSlotBytesMax;
    if (force_wide) w.slot_width = kSlotHitsWide;
    if (n > w.cap_guides || (force_wide && w.slots)) {
        grew = true;
        const size_t cap = std::max<size_t>(std::max<size_t>(n, w.cap_guides), 1024);
        const size_t slots = cap * ix->hdr.n_slices + kGuideGroup * nb;
        const size_t items = nb + cap * ix->hdr.n_slices / 8 + 2; // item sizes down to 8 guides (item_guides knob)
        if ((rc = dev_alloc(w.gword, slots))) return rc;
        if ((rc = dev_alloc(w.gidx, slots))) return rc;
        if ((rc = dev_alloc(w.gbucket, slots))) return rc;
        if ((rc = dev_alloc(w.items, items + 1))) return rc;
        if ((rc = dev_alloc(w.gcount, cap + 1))) return rc;
        if ((rc = dev_alloc(w.goff, cap + 1))) return rc;
        if ((rc = dev_alloc(w.gcur_big, cap + 1))) return rc;
        if ((rc = dev_alloc(w.gcur_big2, cap + 1))) return rc;
        if ((rc = dev_alloc(w.blocksum, (cap + 1) / 2048 + 2))) return rc;
        if ((rc = dev_alloc(w.d_guides, cap))) return rc;
        if ((rc = dev_alloc(w.d_mit, cap))) return rc;
        if ((rc = dev_alloc(w.d_cfd, cap))) return rc;
        if ((rc = dev_alloc(w.d_kept, cap))) return rc;
        // hit slots (Workspace): kSlotHits x 32 bytes per guide -- 1.6 GB for 100 k guides; batches beyond kSlotBytesMax (or a
        // device short of memory) go without, every hit then passes through the grouping pass
        if (w.slots) { (void)hipFree(w.slots); w.slots = nullptr; }
        w.cap_slot_guides = 0;
        if (cap * w.slot_width * sizeof(SlotRec) > kSlotBytesMax) w.slot_width = kSlotHits; // (a larger batch: back to narrow slots)
        if (cap * w.slot_width * sizeof(SlotRec) <= kSlotBytesMax) {
            if (hipMalloc(reinterpret_cast<void **>(&w.slots), cap * w.slot_width * sizeof(SlotRec)) == hipSuccess) w.cap_slot_guides = cap;
            else { (void)hipGetLastError(); w.slots = nullptr; }
        }
        w.cap_guides = cap;
        w.cap_gslots = slots;
        w.cap_items = items;
    }
    // pruned scan: every guide sits in up to 13 (max_dist 5: 67) successor-byte groups of each of its 5 buckets.  These arrays
    // grow with the batch AND with the number of groups per bucket -- by themselves: the staging buffers above are in use by
    // the caller when a batch's max_dist asks for more groups.
    if (ix->hdr.off_sub_start && (grew || fine_ways > w.fine_ways)) {
        const size_t cap = w.cap_guides;
        const uint32_t ways = std::max(fine_ways, w.fine_ways);
        const size_t m = std::min<size_t>(cap, prune_max_guides(ways > kFineWays ? 3u : 2u, ix->hdr.n_slices));
        const size_t places = m * ix->hdr.n_slices * ways;
        const size_t groups = std::min<size_t>(nb * 256, places);
        const size_t fslots = places + kGuideGroup * groups;
        // one item per tile of a group (and per 512 guides of it): sized from the mean group length (uniform data has
        // sites / 65536 candidates per group -- sites / 4096 with 4-bit slices --, +1.2 tiles for the ends); a batch that needs more scans whole buckets
        // and reports it (sticky[3]), finish_batches() then enlarges the list for the next one
        const size_t tiles_per_group = static_cast<size_t>(ix->hdr.n_sites * ix->hdr.n_slices / (static_cast<uint64_t>(nb) * 256ull * kTileCands)) + 4;
        // (fine_items knob: start with a short list -- tests of the two ways out of a list that is too short)
        const size_t fitems = std::max<size_t>(tn.fine_items ? tn.fine_items : tiles_per_group * (groups + places / 64) + 2, w.cap_fitems);
        if ((rc = dev_alloc(w.fword, fslots + 64))) return rc; // (+ slack: short_unit_masks reads whole groups of 32 slots)
        if ((rc = dev_alloc(w.fmeta, fslots))) return rc;
        if ((rc = dev_alloc(w.fitems, fitems + 1))) return rc;
        if ((rc = dev_alloc(w.fcount, nb * 256))) return rc;
        if ((rc = dev_alloc(w.fcount0, nb * 256))) return rc;
        if ((rc = dev_alloc(w.fsum, nb))) return rc;
        w.cap_fslots = fslots;
        w.cap_fitems = fitems;
        w.fine_ways = ways;
    }

    if (w.cap_chunks == 0) {
        // every scan wave may hold one partly filled chunk; beyond that ~1 record per 50k comparisons.
        // raw_chunks knob: start with a small raw buffer (tests of the grow-and-rerun path)
        const size_t want = tn.raw_chunks ? tn.raw_chunks : std::max<size_t>(size_t(scan_waves(tn)) * 6, n);
        if ((rc = ensure_raw_capacity(w, want))) return rc;
    }
    if (!ix->have_events) {
        for (auto &e : ix->ring) HIP_TRY(hipEventCreate(&e));
        ix->have_events = true;
    }
    if (!lane.ready) {
        for (auto &e : lane.ev) HIP_TRY(hipEventCreate(&e));
        HIP_TRY(hipEventCreate(&lane.done));
        int prio_low = 0, prio_high = 0; // (numerically lowest = most urgent)
        HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
        HIP_TRY(hipStreamCreateWithPriority(&lane.stream, hipStreamNonBlocking, prio_low));
        HIP_TRY(hipStreamCreateWithPriority(&lane.tail_stream, hipStreamNonBlocking, prio_high));
        lane.ready = true;
    }
    if (!w.stamps && !tn.stamps_path.empty()) { // diagnostics: per-wave start/end times of the scan
        if ((rc = dev_alloc(w.stamps, kStampsWords))) return rc;
        HIP_TRY(hipMemset(w.stamps, 0, 8ull * kStampsWords));
    }
    if (!w.sticky) {
        if ((rc = dev_alloc(w.sticky, 4))) return rc;
        HIP_TRY(hipMemset(w.sticky, 0, 16));
    }
    return ISSL_OK;
}

// The local MIT table can be indexed directly by the 20 mismatch flags when every mask keeps to the even bits
// below bit 40 (always true for tables written by isslCreateIndex.cpp:239-252).
static bool masks_are_dense(const std::vector<uint64_t> &masks)
{
    for (uint64_t m : masks)
        if (m & ~0x5555555555ull) return false;
    return true;
}

static uint32_t dense_index(uint64_t mask)
{
    uint32_t idx = 0;
    for (uint32_t p = 0; p < 20; ++p) idx |= static_cast<uint32_t>((mask >> (2 * p)) & 1ull) << p;
    return idx;
}

static double wall_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// upload_timing knob (ISSL_UPLOAD_TIMING=1): one diagnostic line per upload stage on stderr
static void upload_note(const issl_index *ix, const char *what, double t0)
{
    if (ix->tuning.upload_timing) std::fprintf(stderr, "[issl upload] %s %.1f ms\n", what, wall_ms() - t0);
}

// Non-null when the slice lists are to be built on the device (the host index then has no arrays).
struct DeviceBuildInput {
    const uint64_t *sigs;
    const uint32_t *occ;
    bool on_device; // the two arrays are device memory of the upload's device (issl_index_build_from_device_sites)
};

struct DevTemp { // device allocation freed on every path out of a function
    void *p = nullptr;
    ~DevTemp() { if (p) (void)hipFree(p); }
};

// Sections of a file-mapped index into device memory.  hipMemcpy from a FRESH private file mapping moves 11 GB/s on an
// MI355X host (every page of the mapping is faulted in on the way; 56 GB/s once they are), so the 14 GB of a human-scale
// .issl took 0.6 - 0.8 s of a one-shot scorer's second.  Here a few threads pread() the file into a ring of pinned chunks
// and every chunk goes out with its own asynchronous copy: 48 - 53 GB/s, the link's rate (tools/ubench_h2d.cpp,
// profiles/r05_ubench_h2d.txt).  Anything that is not file-backed, or small, takes the plain copy.
// Sections are QUEUED (begin) and waited for one by one (wait): the readers go from the last chunk of one section
// straight to the first of the next while the caller launches the kernels that consume the section that has landed.
// Every reader pins its own two slots when it first needs them (pinning costs ~0.3 ms per MiB: 80 ms for the whole ring
// in one go, before the first byte moved), and the ring is given back on a thread of its own (release_async: another
// 40 ms nobody has to wait for).
class FileUploader {
  public:
    // chunk_kib: bytes per pinned slot (default 16 MiB); min_kib: sections smaller than this take the plain copy (default 64 MiB).
    // Both from the upload_chunk_kib / upload_ring_min_kib knobs: tests send a 10 MB golden index through a ring of 64 KiB slots.
    FileUploader(size_t chunk_kib, size_t min_kib, int threads)
        : chunk_(std::max<size_t>(chunk_kib, 4) << 10), min_bytes_(min_kib << 10), n_threads_(static_cast<uint32_t>(std::min(std::max(threads, 1), static_cast<int>(kMaxThreads)))) {}
    ~FileUploader() { abandon_ = true; release(); } // (an upload that failed half way: what is still queued is dropped)
    FileUploader(const FileUploader &) = delete;
    FileUploader &operator=(const FileUploader &) = delete;

    // Queue a section; *ticket names it for wait().  Plain copies are done before this returns.
    int begin(const HostIndex &h, void *dst, const void *src, size_t bytes, int *ticket)
    {
        int fd = -1;
        uint64_t off = 0;
        std::unique_ptr<Job> job(new (std::nothrow) Job());
        if (!job) { set_error("out of memory"); return ISSL_E_NOMEM; }
        if (bytes < std::max<size_t>(min_bytes_, 1) || !h.file_range(src, bytes, &fd, &off) || !ensure()) {
            HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
            job->recorded = true; // (nothing to wait for)
            std::lock_guard<std::mutex> lock(mu_);
            jobs_.push_back(std::move(job));
            *ticket = static_cast<int>(jobs_.size() - 1);
            return ISSL_OK;
        }
        job->fd = fd;
        job->off = off;
        job->dst = static_cast<char *>(dst);
        job->src = static_cast<const char *>(src);
        job->bytes = bytes;
        job->n_chunks = (bytes + chunk_ - 1) / chunk_;
        if (hipEventCreateWithFlags(&job->landed, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            set_error("HIP error: cannot create an event for a section of the index");
            return ISSL_E_DEVICE;
        }
        {
            std::lock_guard<std::mutex> lock(mu_);
            jobs_.push_back(std::move(job));
            *ticket = static_cast<int>(jobs_.size() - 1);
            if (pool_.empty()) {
                (void)hipGetDevice(&device_);
                for (uint32_t w = 0; w < n_threads_; ++w) pool_.emplace_back([this, w] { work(w); });
            }
        }
        cv_work_.notify_all();
        return ISSL_OK;
    }
    // Returns when the section has landed in device memory (or could not be read).
    int wait(int ticket)
    {
        Job *j = nullptr;
        {
            std::unique_lock<std::mutex> lock(mu_);
            if (ticket < 0 || static_cast<size_t>(ticket) >= jobs_.size()) { set_error("internal: no such upload section"); return ISSL_E_STATE; }
            j = jobs_[static_cast<size_t>(ticket)].get();
            cv_done_.wait(lock, [&] { return j->recorded; });
        }
        if (j->landed && !j->failed.load()) HIP_TRY(hipEventSynchronize(j->landed));
        if (j->failed.load() == 2) { set_error("Error reading index: the file shrank or could not be read while it was uploaded"); return ISSL_E_IO; }
        if (j->failed.load()) { (void)hipGetLastError(); set_error("HIP error while a section of the index was uploaded"); return ISSL_E_DEVICE; }
        return ISSL_OK;
    }
    int copy(const HostIndex &h, void *dst, const void *src, size_t bytes)
    {
        int t = -1;
        if (int rc = begin(h, dst, src, bytes, &t)) return rc;
        return wait(t);
    }
    // Stops the readers (they finish what is queued) and frees everything.
    void release()
    {
        std::vector<void *> pins = stop();
        for (void *p : pins) (void)hipHostFree(p);
    }
    double pin_ms() const { return pin_us_.load() * 1e-3; } // summed over the readers (they pin side by side)

  private:
    static constexpr uint32_t kMaxThreads = 32;
    struct Job {
        int fd = -1;
        uint64_t off = 0;
        char *dst = nullptr;
        const char *src = nullptr;
        size_t bytes = 0, n_chunks = 0;
        size_t next = 0, issued = 0; // under mu_
        std::atomic<int> failed{0};
        hipEvent_t landed = nullptr; // recorded behind the section's last copy
        bool recorded = false;       // under mu_: every chunk has been issued (or given up)
    };
    const size_t chunk_; // 16 slots of 16 MiB by default: 256 MiB of pinned memory while an upload lasts
    const size_t min_bytes_;
    const uint32_t n_threads_; // readers (upload_threads knob, default 8)

    bool ensure() // the copy stream; false: plain copies from here on
    {
        if (stream_) return true;
        if (tried_) return false;
        tried_ = true;
        if (hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); stream_ = nullptr; return false; }
        return true;
    }
    void work(uint32_t w)
    {
        (void)hipSetDevice(device_);
        for (uint32_t turn = 0;; ++turn) {
            Job *j = nullptr;
            size_t c = 0;
            {
                std::unique_lock<std::mutex> lock(mu_);
                for (;;) {
                    while (head_ < jobs_.size() && jobs_[head_]->next >= jobs_[head_]->n_chunks) ++head_;
                    if (head_ < jobs_.size()) break;
                    if (stop_) return;
                    cv_work_.wait(lock);
                }
                j = jobs_[head_].get();
                c = j->next++;
            }
            const uint32_t slot = w + (turn & 1u) * kMaxThreads; // every reader alternates between its two slots
            const size_t len = std::min(chunk_, j->bytes - c * chunk_);
            bool issued_here = false;
            if (!j->failed.load() && !abandon_.load()) {
                if (!pin_[slot] && !no_pin_[slot]) { // first use: pin it (side by side with the other readers)
                    const double t0 = wall_ms();
                    if (hipHostMalloc(&pin_[slot], chunk_, hipHostMallocDefault) != hipSuccess ||
                        hipEventCreateWithFlags(&ev_[slot], hipEventDisableTiming) != hipSuccess) {
                        (void)hipGetLastError();
                        if (pin_[slot]) (void)hipHostFree(pin_[slot]);
                        pin_[slot] = nullptr;
                        no_pin_[slot] = true;
                    }
                    pin_us_ += static_cast<long long>((wall_ms() - t0) * 1e3);
                }
                if (!pin_[slot]) { // no pinned memory to be had: this chunk straight from the mapping
                    if (hipMemcpy(j->dst + c * chunk_, j->src + c * chunk_, len, hipMemcpyHostToDevice) != hipSuccess) j->failed = 1;
                } else if (hipEventSynchronize(ev_[slot]) != hipSuccess) { // the slot's previous copy has left it
                    j->failed = 1;
                } else {
                    size_t got = 0;
                    while (got < len) {
                        const ssize_t k = ::pread(j->fd, static_cast<char *>(pin_[slot]) + got, len - got, static_cast<off_t>(j->off + c * chunk_ + got));
                        if (k < 0 && errno == EINTR) continue;
                        if (k <= 0) { j->failed = 2; break; }
                        got += static_cast<size_t>(k);
                    }
                    if (got == len) {
                        std::lock_guard<std::mutex> lock(mu_); // one thread at a time talks to the stream
                        if (hipMemcpyAsync(j->dst + c * chunk_, pin_[slot], len, hipMemcpyHostToDevice, stream_) != hipSuccess ||
                            hipEventRecord(ev_[slot], stream_) != hipSuccess) j->failed = 1;
                        finish_chunk(j);
                        issued_here = true;
                    }
                }
            }
            if (!issued_here) {
                std::lock_guard<std::mutex> lock(mu_);
                finish_chunk(j);
            }
        }
    }
    void finish_chunk(Job *j) // under mu_
    {
        if (++j->issued < j->n_chunks) return;
        if (!j->failed.load() && hipEventRecord(j->landed, stream_) != hipSuccess) j->failed = 1;
        j->recorded = true;
        cv_done_.notify_all();
    }
    std::vector<void *> stop()
    {
        {
            std::lock_guard<std::mutex> lock(mu_);
            stop_ = true;
        }
        cv_work_.notify_all();
        for (auto &th : pool_) th.join();
        pool_.clear();
        if (stream_) (void)hipStreamSynchronize(stream_);
        std::vector<void *> pins;
        for (uint32_t i = 0; i < 2 * kMaxThreads; ++i) {
            if (pin_[i]) pins.push_back(pin_[i]);
            pin_[i] = nullptr;
            if (ev_[i]) (void)hipEventDestroy(ev_[i]);
            ev_[i] = nullptr;
            no_pin_[i] = false;
        }
        for (auto &j : jobs_) if (j->landed) { (void)hipEventDestroy(j->landed); j->landed = nullptr; }
        jobs_.clear();
        head_ = 0;
        if (stream_) (void)hipStreamDestroy(stream_);
        stream_ = nullptr;
        stop_ = false;
        tried_ = false;
        return pins;
    }
    void *pin_[2 * kMaxThreads] = {};   // slot w and w + kMaxThreads belong to reader w alone
    hipEvent_t ev_[2 * kMaxThreads] = {};
    bool no_pin_[2 * kMaxThreads] = {};
    hipStream_t stream_ = nullptr;
    int device_ = 0;
    std::mutex mu_; // the queue, the sections' counts, the stream
    std::condition_variable cv_work_, cv_done_;
    std::vector<std::unique_ptr<Job>> jobs_;
    size_t head_ = 0; // first section that still has chunks to hand out
    std::vector<std::thread> pool_;
    bool stop_ = false, tried_ = false;
    std::atomic<bool> abandon_{false};
    std::atomic<long long> pin_us_{0};
};

static int finish_upload(issl_index *ix, const DeviceBuildInput *dbi = nullptr)
{
    const HostIndex &h = *ix->host;
    const Geometry &g = h.geo;
    const uint64_t nb = g.n_buckets();
    const double t_tables = wall_ms();
    const hipMemcpyKind dbi_kind = (dbi && dbi->on_device) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    uint8_t *base = static_cast<uint8_t *>(ix->d_image);
    // The big sections of a file-mapped index are queued first: the readers pin their slots and fill them while this
    // thread makes the tables below (the process's first copy also loads the runtime's copy kernels: 40 ms).
    FileUploader from_file(ix->tuning.upload_chunk_kib, ix->tuning.upload_ring_min_kib, ix->tuning.upload_threads);
    const bool lists_to_image = !dbi && !ix->hdr.lists_absent && !(ix->hdr.cold_on_host & 1u); // slice lists: file -> image, slice by slice
    const bool sites_to_image = !dbi && !(ix->hdr.cold_on_host & 2u);
    int t_sites = -1;
    std::vector<int> t_list(g.n_slices, -1);
    uint64_t *const img_entries = reinterpret_cast<uint64_t *>(base + ix->hdr.off_entries);
    auto begin_list = [&](uint64_t sl) -> int { // (two sections at most are queued ahead of the one waited for)
        if (!lists_to_image || sl >= g.n_slices || t_list[sl] >= 0) return ISSL_OK;
        return from_file.begin(h, img_entries + sl * g.n_sites, h.entries + sl * g.n_sites, 8 * g.n_sites, &t_list[sl]);
    };
    if (sites_to_image)
        if (int crc = from_file.begin(h, base + ix->hdr.off_sites, h.sites, 8 * g.n_sites, &t_sites)) return crc;
    if (int crc = begin_list(0)) return crc;
    // the head of the image -- header, bucket tables, score tables: contiguous -- as ONE copy
    std::vector<uint64_t> masks;
    std::vector<double> vals;
    h.unique_scores(masks, vals);
    std::vector<uint32_t> tfirst(nb + 1);
    {
        const uint64_t head_end = ix->hdr.off_mit_dense ? ix->hdr.off_mit_dense + 8ull * (1u << 20) : ix->hdr.off_score_val + 8 * masks.size();
        std::vector<uint8_t> head(head_end, 0);
        std::memcpy(head.data(), &ix->hdr, sizeof(ImageHeader));
        uint64_t *bstart = reinterpret_cast<uint64_t *>(head.data() + ix->hdr.off_bucket_start);
        bstart[0] = 0;
        tfirst[0] = 0;
        for (uint64_t b = 0; b < nb; ++b) {
            bstart[b + 1] = bstart[b] + h.sizes[b];
            tfirst[b + 1] = tfirst[b] + static_cast<uint32_t>((h.sizes[b] + kTileCands - 1) / kTileCands);
        }
        std::memcpy(head.data() + ix->hdr.off_tile_first, tfirst.data(), 4 * (nb + 1));
        if (!masks.empty()) {
            std::memcpy(head.data() + ix->hdr.off_score_mask, masks.data(), 8 * masks.size());
            std::memcpy(head.data() + ix->hdr.off_score_val, vals.data(), 8 * vals.size());
        }
        if (ix->hdr.off_mit_dense) {
            double *dense = reinterpret_cast<double *>(head.data() + ix->hdr.off_mit_dense);
            for (size_t i = 0; i < masks.size(); ++i) dense[dense_index(masks[i])] = vals[i];
        }
        HIP_TRY(hipMemcpy(base, head.data(), head_end, hipMemcpyHostToDevice));
    }
    ix->view = make_view(ix->hdr, ix->d_image, ix->d_cold);
    DevTemp flag_mem, occ_mem;
    HIP_TRY(hipMalloc(&flag_mem.p, 4));
    uint32_t *flag = static_cast<uint32_t *>(flag_mem.p);
    HIP_TRY(hipMemset(flag, 0, 4));
    uint32_t *scan_out = reinterpret_cast<uint32_t *>(base + ix->hdr.off_scan);
    if (dbi && !ix->hdr.off_sub_start) { // (the sorted layouts keep the counts in the image)
        HIP_TRY(hipMalloc(&occ_mem.p, 4 * g.n_sites));
        HIP_TRY(hipMemcpy(occ_mem.p, dbi->occ, 4 * g.n_sites, dbi_kind));
    }
    const uint32_t *d_occ = static_cast<const uint32_t *>(occ_mem.p);
    DevTemp seen_mem; // list-order layouts: one bit per (slice, site) -- every slice must list every site once
    uint32_t *seen = nullptr;
    if (!ix->hdr.off_sub_start) {
        const uint64_t words = (g.n_sites * g.n_slices + 31) / 32 + 1;
        if (hipMalloc(&seen_mem.p, 4 * words) != hipSuccess) { (void)hipGetLastError(); seen_mem.p = nullptr; return kSortNoRoom; } // the next layout
        HIP_TRY(hipMemset(seen_mem.p, 0, 4 * words));
        seen = static_cast<uint32_t *>(seen_mem.p);
    }
    upload_note(ix, "bucket tables, score table (the file's sections are on their way)", t_tables);
    double t0 = wall_ms();
    if (ix->hdr.off_sub_start) {
        // Sorted layouts.  Site table and counts into the image, then one slice at a time: the slice's list (in the
        // image, or -- lists in pinned host memory -- in a temporary 8 B/site device copy), the successor-byte order
        // of its buckets, its part of the stream maps.
        const uint64_t n = g.n_sites;
        uint64_t *d_sites = reinterpret_cast<uint64_t *>(base + ix->hdr.off_sites);
        uint32_t *d_site_occ = reinterpret_cast<uint32_t *>(base + ix->hdr.off_site_occ);
        if (dbi) HIP_TRY(hipMemcpy(d_sites, dbi->sigs, 8 * n, dbi_kind));
        else if (int crc = from_file.wait(t_sites)) return crc;
        if (dbi) HIP_TRY(hipMemcpy(d_site_occ, dbi->occ, 4 * n, dbi_kind)); // (k_fill_maps writes the same again)
        upload_note(ix, "sites", t0);
        t0 = wall_ms();
        // lists in pinned host memory, or nowhere (lists_absent): either way a slice's list exists on the device only while
        // the slice is worked on, in one 8 B/site temporary
        const bool lists_kept_cold = (ix->hdr.cold_on_host & 1u) != 0;
        const bool lists_cold = lists_kept_cold || ix->hdr.lists_absent != 0;
        // The scan stream is packed last (from the maps the slices leave behind): until then its section -- 20 B per site --
        // holds the sort keys and the one slice list, so that the construction needs 8 B per site beyond the image
        // (the radix passes' second buffer) and an index of the format's 2^32 - 1 sites (52 + 8 B per site) fits 288 GB.
        const uint64_t scan_bytes = ix->hdr.n_tiles * static_cast<uint64_t>(kTileCands) * 4ull, key_bytes = align256(8 * n);
        const bool lend = scan_bytes >= key_bytes + (lists_cold ? 8 * n : 0) && n > 0;
        SortTemp st;
        int src = st.alloc(n, lend ? scan_out : nullptr);
        if (src) return src;
        DevTemp list_mem;
        uint64_t *lent_list = (lend && lists_cold) ? reinterpret_cast<uint64_t *>(reinterpret_cast<uint8_t *>(scan_out) + key_bytes) : nullptr;
        uint64_t *d_entries = lists_cold ? nullptr : reinterpret_cast<uint64_t *>(base + ix->hdr.off_entries);
        uint64_t *c_entries = lists_kept_cold ? reinterpret_cast<uint64_t *>(static_cast<uint8_t *>(ix->h_cold) + ix->hdr.off_entries) : nullptr;
        if (lists_cold) {
            if (!lent_list && hipMalloc(&list_mem.p, std::max<uint64_t>(8 * n, 8)) != hipSuccess) { (void)hipGetLastError(); list_mem.p = nullptr; return kSortNoRoom; }
        } else if (dbi) { // isslCreateIndex.cpp:218-234 on the device
            int brc = launch_build_entries(d_sites, d_site_occ, n, 0, static_cast<uint32_t>(g.n_slices),
                                           static_cast<uint32_t>(g.slice_width), d_entries);
            if (brc) return brc;
            upload_note(ix, "slice lists built on the device", t0);
        }
        // Lists from the host: one slice at a time, so that the kernels that order slice s run while slice s + 1 is on
        // its way (the copy returns when the slice has landed; the kernels are asynchronous on the null stream, the
        // copies run on a stream of their own).
        const bool stream_lists = !lists_cold && !dbi;
        t0 = wall_ms();
        for (uint64_t sl = 0; sl < g.n_slices; ++sl) {
            uint64_t *const t_list_mem = lent_list ? lent_list : static_cast<uint64_t *>(list_mem.p);
            const uint64_t *d_list = lists_cold ? t_list_mem : d_entries + sl * n;
            if (stream_lists) {
                if (int crc = begin_list(sl + 1)) return crc;
                if (int crc = from_file.wait(t_list[sl])) return crc;
            }
            if (lists_cold) {
                uint64_t *t_list = t_list_mem;
                if (dbi) {
                    int brc = launch_build_entries(d_sites, d_site_occ, n, static_cast<uint32_t>(sl), static_cast<uint32_t>(sl + 1),
                                                   static_cast<uint32_t>(g.slice_width), t_list);
                    if (brc) return brc;
                    if (c_entries) HIP_TRY(hipMemcpy(c_entries + sl * n, t_list, 8 * n, hipMemcpyDeviceToHost));
                } else if (c_entries) {
                    std::memcpy(c_entries + sl * n, h.entries + sl * n, 8 * n);
                    HIP_TRY(hipMemcpy(t_list, c_entries + sl * n, 8 * n, hipMemcpyHostToDevice));
                } else {
                    if (int crc = from_file.copy(h, t_list, h.entries + sl * n, 8 * n)) return crc;
                }
            }
            src = launch_sort_slice(st, d_sites, d_list, reinterpret_cast<const uint64_t *>(base + ix->hdr.off_bucket_start),
                                    reinterpret_cast<const uint32_t *>(base + ix->hdr.off_tile_first), n,
                                    static_cast<uint32_t>(g.n_slices), static_cast<uint32_t>(nb), static_cast<uint32_t>(g.slice_width),
                                    static_cast<uint32_t>(sl), reinterpret_cast<uint32_t *>(base + ix->hdr.off_sub_start),
                                    ix->hdr.off_srec ? reinterpret_cast<StreamRec *>(base + ix->hdr.off_srec) : nullptr,
                                    ix->hdr.off_sid ? reinterpret_cast<uint32_t *>(base + ix->hdr.off_sid) : nullptr, d_site_occ, flag);
            if (src) return src;
            if (lists_cold) HIP_TRY(hipDeviceSynchronize()); // the temporary list is overwritten by the next slice
        }
        src = finish_sort(flag);
        if (src) return src;
        st.release();
        upload_note(ix, stream_lists ? "entries, slice by slice, beside the sorted layout (successor-byte order of every bucket + stream maps)"
                                     : "sorted layout (successor-byte order of every bucket + stream maps)", t0);
        t0 = wall_ms();
        launch_pack_scan_stream(ix->view, scan_out, nullptr, nullptr, flag, nullptr, nullptr);
        HIP_TRY(hipGetLastError());
        launch_tag_sites(d_sites, d_site_occ, n); // (last: from here on `sites` carries a 24-bit copy of the counts)
        HIP_TRY(hipGetLastError());
    } else if (!ix->hdr.cold_on_host) {
        // (file-mapped host arrays go through FileUploader's pinned ring, everything else through plain copies)
        if (dbi) HIP_TRY(hipMemcpy(base + ix->hdr.off_sites, dbi->sigs, 8 * g.n_sites, dbi_kind));
        else if (int crc = from_file.wait(t_sites)) return crc;
        upload_note(ix, "sites", t0);
        t0 = wall_ms();
        if (dbi) { // isslCreateIndex.cpp:218-234 on the device
            int brc = launch_build_entries(reinterpret_cast<const uint64_t *>(base + ix->hdr.off_sites), d_occ, g.n_sites,
                                           0, static_cast<uint32_t>(g.n_slices), static_cast<uint32_t>(g.slice_width),
                                           reinterpret_cast<uint64_t *>(base + ix->hdr.off_entries));
            if (brc) return brc;
            upload_note(ix, "slice lists built on the device", t0);
            t0 = wall_ms();
            launch_pack_scan_stream(ix->view, scan_out,
                                    ix->hdr.off_esig ? reinterpret_cast<uint64_t *>(base + ix->hdr.off_esig) : nullptr, nullptr, flag,
                                    seen, nullptr);
            HIP_TRY(hipGetLastError());
        } else {
            // scan stream: built on the device from sites + entries, one slice at a time: the kernel that packs slice s runs
            // while the list of slice s + 1 is on its way (a slice's buckets own a contiguous run of tiles)
            for (uint64_t sl = 0; sl < g.n_slices; ++sl) {
                if (int crc = begin_list(sl + 1)) return crc;
                if (int crc = from_file.wait(t_list[sl])) return crc;
                launch_pack_scan_range(ix->view, scan_out, ix->hdr.off_esig ? reinterpret_cast<uint64_t *>(base + ix->hdr.off_esig) : nullptr,
                                       nullptr, flag, seen, tfirst[sl << g.slice_width], tfirst[(sl + 1) << g.slice_width], nullptr);
                HIP_TRY(hipGetLastError());
            }
            upload_note(ix, "entries, slice by slice, beside the packing of the scan stream", t0);
            t0 = wall_ms();
        }
    } else {
        // List-order layout with sites and lists in pinned host memory: the scan stream is packed one slice at a time from temporary device
        // copies of the signatures (8 B/site) and of that slice's list (8 B/site); random reads of the site table
        // across PCIe would take minutes.  With a device-side build the lists are made here and copied out.
        uint8_t *cold = static_cast<uint8_t *>(ix->h_cold);
        uint64_t *c_sites = reinterpret_cast<uint64_t *>(cold + ix->hdr.off_sites);
        uint64_t *c_entries = reinterpret_cast<uint64_t *>(cold + ix->hdr.off_entries);
        const uint64_t n = g.n_sites;
        DevTemp sites_mem, list_mem;
        HIP_TRY(hipMalloc(&sites_mem.p, std::max<uint64_t>(8 * n, 8)));
        HIP_TRY(hipMalloc(&list_mem.p, std::max<uint64_t>(8 * n, 8)));
        uint64_t *t_sites = static_cast<uint64_t *>(sites_mem.p), *t_list = static_cast<uint64_t *>(list_mem.p);
        if (dbi && dbi->on_device) HIP_TRY(hipMemcpy(c_sites, dbi->sigs, 8 * n, hipMemcpyDeviceToHost));
        else std::memcpy(c_sites, dbi ? dbi->sigs : h.sites, 8 * n);
        HIP_TRY(hipMemcpy(t_sites, c_sites, 8 * n, hipMemcpyHostToDevice));
        upload_note(ix, "sites (pinned host copy + temporary device copy)", t0);
        t0 = wall_ms();
        for (uint64_t sl = 0; sl < g.n_slices; ++sl) {
            if (dbi) {
                int brc = launch_build_entries(t_sites, d_occ, n, static_cast<uint32_t>(sl), static_cast<uint32_t>(sl + 1),
                                               static_cast<uint32_t>(g.slice_width), t_list);
                if (brc) return brc;
                HIP_TRY(hipMemcpy(c_entries + sl * n, t_list, 8 * n, hipMemcpyDeviceToHost));
            } else {
                std::memcpy(c_entries + sl * n, h.entries + sl * n, 8 * n);
                HIP_TRY(hipMemcpy(t_list, c_entries + sl * n, 8 * n, hipMemcpyHostToDevice));
            }
            ImageView pv = ix->view;
            pv.sites = t_sites;
            pv.entries = t_list - sl * n; // bucket_start of the slice's first bucket is sl * n: every site sits in one bucket per slice
            launch_pack_scan_range(pv, scan_out, nullptr, reinterpret_cast<uint8_t *>(base + ix->hdr.off_occ8), flag, seen,
                                   tfirst[sl << g.slice_width], tfirst[(sl + 1) << g.slice_width], nullptr);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipDeviceSynchronize());
        }
        upload_note(ix, "slice lists into pinned host memory", t0);
        t0 = wall_ms();
    }
    uint32_t err = 0;
    HIP_TRY(hipMemcpy(&err, flag, 4, hipMemcpyDeviceToHost));
    upload_note(ix, "scan stream", t0);
    if (ix->tuning.upload_timing) std::fprintf(stderr, "[issl upload] (pinning the ring: %.1f ms, summed over the readers)\n", from_file.pin_ms());
    const double t_ring = wall_ms();
    from_file.release();
    upload_note(ix, "pinned ring given back", t_ring);
    if (err & 1u) {
        set_error("Error reading index: a slice entry refers to an off-target id beyond the site table");
        return ISSL_E_FORMAT;
    }
    if (err) {
        set_error("Error reading index: a slice list holds an off-target in a bucket its signature does not select, or twice");
        return ISSL_E_FORMAT;
    }
    return ISSL_OK;
}

// The optional in-list signatures cost 8 B per list entry (40 B per site next to the 68 B of the rest): worth it while
// the image stays a modest part of the HBM (inline_sigs knob / ISSL_INLINE_SIGS=0/1 overrides).
static bool want_inline_sigs(const Tuning &tn, const Geometry &g)
{
    if (tn.inline_sigs >= 0) return tn.inline_sigs == 1;
    return g.n_sites <= 600000000ull;
}

// The layouts an upload tries, in turn, until one fits the free HBM.  Bytes per site next to the 20 B of the scan
// stream: sorted 132 (16-byte stream records, site table, counts, slice lists), compact sorted 72 or -- slice lists in
// pinned host memory -- 32; list order 88 / 48 (with / without the in-list signatures) or, all cold sections in host
// memory, 5.  The sorted ones let the scan skip 243 of every 256 successor-byte groups; they need lists that ascend by
// site id (list_order_only: this index's do not).  Explicit options are honoured or the upload fails.
static std::vector<LayoutSpec> layout_choices(const Tuning &tn, const Geometry &g, bool list_order_only)
{
    std::vector<LayoutSpec> c;
    auto spec = [](bool esig, uint32_t cold, uint32_t sorted, bool no_lists = false) { LayoutSpec s; s.inline_sigs = esig; s.cold = cold; s.sorted = sorted; s.no_lists = no_lists; return s; };
    // Narrow slices (4 / 2 bits; round 4): the sorted layouts order a bucket by the byte of the next two / four slices (succ_byte),
    // so everything sorted applies; only the list-order layout with ALL cold sections in host memory does not -- it rebuilds a
    // candidate's signature from the stream's 16 positions + the bucket's byte, and a narrow slice leaves 18 / 19 outside.
    const bool narrow = g.slice_width != 8;
    const bool may_sort = !list_order_only && tn.sorted_layout != 0 && tn.inline_sigs != 1;
    const bool must_sort = tn.sorted_layout == 1 || tn.compact == 1 || tn.keep_lists == 0; // (only a sorted image can do without its lists)
    if (may_sort || must_sort) {
        if (tn.host_cold == 1) {
            if ((tn.compact == 1 || tn.sorted_layout == 1) && tn.keep_lists != 0) c.push_back(spec(false, 1, 2));
        } else {
            if (tn.compact != 1 && tn.keep_lists != 0) c.push_back(spec(false, 0, 1));
            if (tn.compact != 0) {
                if (tn.keep_lists != 0) c.push_back(spec(false, 0, 2));
                // the smallest image: compact and without its slice lists -- 52 B/site, self-contained (nothing in host
                // memory, so it can still be broadcast and attached elsewhere); the variant with the lists in pinned
                // host memory (40 B/site there) is made on request only (host_cold=1)
                if (tn.keep_lists != 1) c.push_back(spec(false, 0, 2, true));
            }
        }
    }
    if (!must_sort) {
        if (tn.host_cold == 1) {
            if (!narrow) c.push_back(spec(false, 3, 0));
        } else {
            if (want_inline_sigs(tn, g)) c.push_back(spec(true, 0, 0));
            if (tn.inline_sigs != 1) c.push_back(spec(false, 0, 0));
            if (tn.host_cold == -1 && tn.inline_sigs != 1 && !narrow) c.push_back(spec(false, 3, 0));
        }
    }
    return c;
}

static uint64_t count_tiles(const HostIndex &h)
{
    uint64_t t = 0;
    for (uint64_t b = 0; b < h.geo.n_buckets(); ++b) t += (h.sizes[b] + kTileCands - 1) / kTileCands;
    return t;
}

static void release_device(issl_index *ix)
{
    if (ix->device >= 0) (void)hipSetDevice(ix->device);
    for (Lane *lp : {&ix->lane, &ix->lane2}) {
        Lane &lane = *lp;
        if (lane.ready) {
            (void)hipStreamSynchronize(lane.stream);
            (void)hipStreamSynchronize(lane.tail_stream);
            for (auto &e : lane.ev) (void)hipEventDestroy(e);
            (void)hipEventDestroy(lane.done);
            (void)hipStreamDestroy(lane.stream);
            (void)hipStreamDestroy(lane.tail_stream);
            lane.ready = false;
        }
        free_workspace(lane.ws);
        lane.lean = false;
        lane.last_n = 0;
        lane.pending = 0;
    }
    ix->last_lane = nullptr;
    if (ix->have_events) {
        for (auto &e : ix->ring) (void)hipEventDestroy(e);
        ix->have_events = false;
    }
    ix->n_pending = 0;
    ix->n_ring = 0;
    ix->proven_guides = 0;
    ix->proven_chunks = 0;
    ix->proven_dist = -1;
    ix->prev_scan_end = nullptr;
    ix->prev_batch_end = nullptr;
    if (ix->d_image && ix->owns_image) (void)hipFree(ix->d_image);
    ix->d_image = nullptr;
    ix->owns_image = false;
    if (ix->h_cold && ix->owns_cold) (void)hipHostFree(ix->h_cold);
    ix->h_cold = nullptr;
    ix->d_cold = nullptr;
    ix->owns_cold = false;
}

static int new_index_from_host(std::unique_ptr<HostIndex> h, issl_index **out)
{
    issl_index *ix = new (std::nothrow) issl_index();
    if (!ix) {
        set_error("out of memory");
        return ISSL_E_NOMEM;
    }
    ix->geo = h->geo;
    ix->bucket_sizes.assign(h->sizes, h->sizes + h->geo.n_buckets());
    ix->host = std::move(h);
    *out = ix;
    return ISSL_OK;
}

// The scoring pipeline.  Guides and outputs are device pointers on ix->device.
// enqueue_batch() only launches (no host round trip); finish_batches() synchronises, checks the sticky overflow
// words the pipelines leave behind, and fills the statistics.
// `staged`: record an event at every stage boundary (bin / scan / verify / group / replay times in issl_stats).  An event
// record costs ~4 us of stream time on MI355X -- 5 % of a 10 k-guide batch for the six of them -- so the asynchronous
// back-to-back path records only the pair around the scan and the end of the batch unless the stage_timing knob is set.
// `pipelined` (asynchronous batches with the lanes option = 2): a software pipeline over two workspaces.  Binning and scan of
// a batch run on the lane's stream, its verify / group / replay on the lane's high-priority tail stream; the scans of
// consecutive batches are chained by events, so that they run one after the other at full speed while the short, latency-
// bound tail of batch i runs beside the scan of batch i + 1 -- a step then costs max(bin + scan, tail) instead of their sum.
static int enqueue_batch(issl_index *ix, Lane &lane, hipStream_t stream, const uint64_t *d_guides, size_t n, int max_dist,
                         double threshold, int method, double *d_mit, double *d_cfd, bool dump, bool staged,
                         int lanes_mode = 1)
{
    // lanes_mode 2: the software pipeline described above.  3 ("binning ahead"): two workspaces as well, but only the BINNING
    // of a batch -- seven short, latency-bound launches, 0.19 ms at 100 k guides -- runs beside the batch before it; its scan
    // waits for that batch's replay, so the heavy kernels never share the chip (which is what made mode 2 lose: they share
    // its power budget).
    const bool pipelined = lanes_mode == 2, bin_ahead = lanes_mode == 3;
    if (!ix->d_image) {
        set_error("index has no device image: call issl_index_upload first");
        return ISSL_E_STATE;
    }
    // guide slots are 27-bit fields of the raw records: one slot per guide and slice + padding
    const size_t max_batch = std::min<size_t>(kMaxBatch, ((size_t(1) << 27) - kGuideGroup * ix->hdr.n_buckets) / std::max<uint64_t>(ix->hdr.n_slices, 1));
    if (n > max_batch) {
        set_error("at most " + std::to_string(max_batch) + " guides per device batch on this index (issl_score splits larger batches itself)");
        return ISSL_E_ARG;
    }
    HIP_TRY(hipSetDevice(ix->device));
    if (n == 0) return ISSL_OK;
    const uint32_t prune_mode = prune_mode_for(ix->view, ix->tuning, static_cast<uint32_t>(n), max_dist);
    int rc = ensure_workspace(ix, n, lane, fine_ways_of(prune_mode ? prune_mode : 2u));
    if (rc) return rc;
    Workspace &ws = lane.ws;
    const Tuning &tn = ix->tuning;
    // `sorted` always has room for every raw slot, so the whole pipeline runs without a host round trip;
    // an exhausted raw buffer is detected in finish_batches() and the batch is re-run with a larger one.
    rc = ensure_hit_capacity(ws, ws.cap_chunks * (kChunkRecs - 1));
    if (rc) return rc;
    if (dump && ws.cap_hitrec < ws.cap_hits) {
        rc = dev_alloc(ws.d_hitrec, ws.cap_hits);
        if (rc) return rc;
        ws.cap_hitrec = ws.cap_hits;
    }
    // issl_dump_hits wants every hit of the batch in one array in guide order: no hit slots there
    ws.slot_hits = (!dump && tn.hit_slots && ws.cap_slot_guides >= n) ? ws.slot_width : 0u;
    ws.lean_tail = (lane.lean && ws.slot_hits >= kSlotHits && tn.lean_tail) ? 1u : 0u;
    ScoreParams p;
    p.max_dist = max_dist;
    p.method = method;
    p.maximum_sum = (10000.0 - threshold * 100) / threshold; // isslScoreOfftargets.cpp:326
    const uint32_t n32 = static_cast<uint32_t>(n);
    // The event pair around the scan (issl_stats::ms_scan_events): around every batch's (scan_events = 1; always where the stage
    // events are recorded too), around the first batch's after a finish (2, the default: the kernel's own clock stamps time
    // every launch anyway, ms_scan) or never (0) -- an event record is ~5 us of stream time, two of them a seventh of a
    // 64-guide batch.  Pipelined lanes chain their scans by these events: there, always.
    const bool scan_pair = staged || pipelined || tn.scan_events == 1 || (tn.scan_events == 2 && ix->n_ring == 0);
    const uint32_t slot = ix->n_ring % kRing;
    lane.staged = staged;
    if (pipelined && lane.pending) HIP_TRY(hipStreamWaitEvent(stream, lane.done, 0)); // the workspace's previous batch (tail stream)
    // (bin_ahead: the workspace's previous batch ran on this very stream)
    if (staged) HIP_TRY(hipEventRecord(lane.ev[0], stream));
    ws.span_slot = lane.pending % kSpanRing;
    launch_bin_guides(ix->view, ws, tn, d_guides, n32, prune_mode, stream);
    if (staged) HIP_TRY(hipEventRecord(lane.ev[1], stream));
    if (pipelined && ix->prev_scan_end) HIP_TRY(hipStreamWaitEvent(stream, ix->prev_scan_end, 0)); // one scan at a time
    if (bin_ahead && ix->prev_batch_end) HIP_TRY(hipStreamWaitEvent(stream, ix->prev_batch_end, 0)); // the batch before is through
    if (scan_pair) HIP_TRY(hipEventRecord(ix->ring[2 * slot], stream));
    launch_scan(ix->view, ws, tn, d_guides, n32, max_dist, prune_mode, stream);
    if (scan_pair) HIP_TRY(hipEventRecord(ix->ring[2 * slot + 1], stream));
    if (scan_pair) ix->n_ring += 1;
    if (staged) HIP_TRY(hipEventRecord(lane.ev[2], stream));
    hipStream_t tail = stream;
    if (pipelined) {
        tail = lane.tail_stream;
        HIP_TRY(hipStreamWaitEvent(tail, ix->ring[2 * slot + 1], 0));
        ix->prev_scan_end = ix->ring[2 * slot + 1];
    }
    launch_verify(ix->view, ws, d_guides, static_cast<uint32_t>(n), p, tail);
    if (staged) HIP_TRY(hipEventRecord(lane.ev[3], tail));
    launch_group_hits(ws, n32, tail);
    if (staged) HIP_TRY(hipEventRecord(lane.ev[4], tail));
    launch_replay(ix->view, ws, d_guides, n32, p, d_mit, d_cfd, dump ? ws.d_kept : nullptr,
                  dump ? ws.d_hitrec : nullptr, tail);
    if (staged) HIP_TRY(hipEventRecord(lane.ev[5], tail));
    if (pipelined || bin_ahead) HIP_TRY(hipEventRecord(lane.done, tail)); // (what the other lane's batches wait for)
    lane.done_recorded = pipelined || bin_ahead; // (one lane: issl_score_wait records it when somebody asks)
    lane.last_tail = tail;
    if (bin_ahead) ix->prev_batch_end = lane.done;
    ix->n_pending += 1;
    lane.pending += 1;
    lane.last_n = n32;
    lane.last_max_dist = max_dist;
    lane.last_prune = prune_mode;
    ix->last_lane = &lane;
    return ISSL_OK;
}

// Synchronises everything that was enqueued (the internal stream and, for synchronous calls, `stream`).  Returns
// ISSL_OK, or ISSL_E_RETRY when a batch since the last finish ran out of raw-record space (the buffers have been
// enlarged; the caller enqueues those batches again).
static int finish_batches(issl_index *ix, hipStream_t stream)
{
    if (!ix->d_image || ix->n_pending == 0) return ISSL_OK;
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(hipStreamSynchronize(stream));
    for (Lane *lp : {&ix->lane, &ix->lane2})
        if (lp->ready && lp->pending) {
            HIP_TRY(hipStreamSynchronize(lp->stream));
            HIP_TRY(hipStreamSynchronize(lp->tail_stream));
        }
    ix->prev_scan_end = nullptr;
    ix->prev_batch_end = nullptr;
    HIP_TRY(hipGetLastError());
    const uint32_t batches = ix->n_pending;
    const uint32_t ring_pairs = ix->n_ring; // scan event pairs recorded since the last finish (scan_events)
    ix->n_pending = 0;
    ix->n_ring = 0;
    bool retry = false;
    uint32_t max_chunks = 0; // of the lane whose counters are reported
    Lane &lane = ix->last_lane ? *ix->last_lane : ix->lane;
    double span_sum = 0.0;   // scan launches by the kernel's own clock stamps (ticks of 10 ns)
    uint32_t span_count = 0;
    for (Lane *lp : {&ix->lane, &ix->lane2}) {
        if (!lp->pending) continue;
        {
            const uint32_t have = lp->pending < kSpanRing ? lp->pending : kSpanRing;
            unsigned long long spans[2 * kSpanRing];
            HIP_TRY(hipMemcpy(spans, lp->ws.scan_span, 16 * have, hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < have; ++i)
                if (spans[2 * i + 1] > spans[2 * i]) { span_sum += static_cast<double>(spans[2 * i + 1] - spans[2 * i]); ++span_count; }
        }
        lp->pending = 0;
        uint32_t sticky[4] = {0, 0, 0, 0};
        HIP_TRY(hipMemcpy(sticky, lp->ws.sticky, sizeof sticky, hipMemcpyDeviceToHost));
        if (lp == &lane) max_chunks = sticky[1];
        if (sticky[2] & 2u) {
            HIP_TRY(hipMemset(lp->ws.sticky, 0, 16));
            set_error("internal error: scan item list overflow");
            return ISSL_E_DEVICE;
        }
        if (sticky[3] > lp->ws.cap_fitems && lp->ws.fitems) { // a pruned plan did not fit its item list: room for the next batch
            const size_t want = static_cast<size_t>(sticky[3]) + sticky[3] / 4 + 2;
            uint32_t zero = 0;
            HIP_TRY(hipMemcpy(lp->ws.sticky + 3, &zero, 4, hipMemcpyHostToDevice));
            int rc = dev_alloc(lp->ws.fitems, want + 1);
            if (rc) return rc;
            lp->ws.cap_fitems = want;
        }
        // Hit slots: when a good part of the last batch's guides had more than kSlotHits hits -- an index of billions of sites, a
        // skewed genome -- the next batches get slots for kSlotHitsWide of them (6.5 GB per 100 k guides), so that only what
        // lies beyond THAT passes through the grouping pass.  A matter of speed only: the results do not depend on the width.
        if (lp->ws.slots && lp->ws.slot_width == kSlotHits && ix->tuning.hit_slots && lp->last_n) {
            Counters c{};
            HIP_TRY(hipMemcpy(&c, lp->ws.counters, sizeof c, hipMemcpyDeviceToHost));
            const size_t want = lp->ws.cap_slot_guides * size_t(kSlotHitsWide) * sizeof(SlotRec);
            size_t free_b = 0, total_b = 0;
            if (c.overflowed > lp->last_n / 8 && want <= kSlotBytesMax && hipMemGetInfo(&free_b, &total_b) == hipSuccess &&
                free_b > want + (size_t(4) << 30)) {
                SlotRec *wide = nullptr;
                if (hipMalloc(reinterpret_cast<void **>(&wide), want) == hipSuccess) {
                    (void)hipFree(lp->ws.slots);
                    lp->ws.slots = wide;
                    lp->ws.slot_width = kSlotHitsWide;
                } else {
                    (void)hipGetLastError();
                }
            }
        }
        // the lane's next batches go without the grouping pass and the many-hit replays while no batch meets a guide beyond
        // its hit slots (bit 2: one did; bit 1: and it had been enqueued lean -- once more, with the whole tail)
        lp->lean = (sticky[0] & 6u) == 0u && lp->ws.slot_hits >= kSlotHits;
        if (sticky[0] & 2u) retry = true;
        if (sticky[0] & 1u) {
            // sticky[1] = largest number of chunks any batch asked for
            int rc = ensure_raw_capacity(lp->ws, static_cast<size_t>(sticky[1]) + sticky[1] / 8 + 1024);
            if (rc) return rc;
            retry = true;
        }
        if (sticky[0]) HIP_TRY(hipMemset(lp->ws.sticky, 0, 16));
    }
    if (retry) {
        set_error("a batch has to be scored again: its raw record buffer was too small (it has been enlarged), or it was enqueued "
                  "without the many-hit part of the pipeline and met a guide that needs it");
        return ISSL_E_RETRY;
    }
    PlanInfo pl{};
    uint32_t total_hits = 0;
    HIP_TRY(hipMemcpy(&pl, lane.ws.plan, sizeof pl, hipMemcpyDeviceToHost));
    {   // scored off-targets of the last batch before any early exit: the per-guide counts k_verify left
        std::vector<uint32_t> counts(lane.last_n);
        if (lane.last_n) HIP_TRY(hipMemcpy(counts.data(), lane.ws.gcount, 4 * counts.size(), hipMemcpyDeviceToHost));
        for (uint32_t c : counts) total_hits += c;
    }
    // comparisons the scan workgroups of the last batch counted while they made them
    std::vector<uint64_t> counted(ix->tuning.scan_blocks);
    HIP_TRY(hipMemcpy(counted.data(), lane.ws.scan_count, 8 * counted.size(), hipMemcpyDeviceToHost));
    uint64_t compared = 0;
    for (uint64_t c : counted) compared += c;
    float ms[5] = {0, 0, 0, 0, 0};
    if (lane.staged)
        for (int i = 0; i < 5; ++i) (void)hipEventElapsedTime(&ms[i], lane.ev[i], lane.ev[i + 1]);
    double scan_sum = 0.0;
    const uint32_t have = ring_pairs < kRing ? ring_pairs : kRing;
    for (uint32_t i = 0; i < have; ++i) {
        float t = 0;
        (void)hipEventElapsedTime(&t, ix->ring[2 * i], ix->ring[2 * i + 1]);
        scan_sum += t;
    }
    ix->stats = issl_stats{};
    ix->stats.n_guides = lane.last_n;
    ix->stats.ms_bin = ms[0];
    ix->stats.ms_scan_events = have ? scan_sum / have : ms[1]; // mean over the batches since the last finish
    ix->stats.ms_scan = span_count ? span_sum / span_count * 1e-5 : 0.0;
    ix->stats.ms_verify = ms[2];
    ix->stats.ms_group = ms[3];
    ix->stats.ms_replay = ms[4];
    ix->stats.ms_total = ms[0] + ms[1] + ms[2] + ms[3] + ms[4];
    ix->stats.raw_records = static_cast<uint64_t>(max_chunks) * (kChunkRecs - 1);
    ix->stats.candidates = compared;
    ix->stats.planned_comparisons = lane.last_max_dist < 0 ? 0 : pl.candidates;
    ix->stats.reference_comparisons = pl.reference_candidates;
    ix->stats.pruned = lane.last_prune ? pl.fine : 0;
    ix->stats.hits = total_hits;
    ix->stats.scan_tiles = pl.tiles;
    ix->stats.n_batches = batches;
    if (lane.ws.stamps) { // scan_stamps knob: dump the wave stamps of the last scan (4 u64 per wave)
        std::vector<unsigned long long> st(kStampsWords);
        HIP_TRY(hipMemcpy(st.data(), lane.ws.stamps, 8ull * kStampsWords, hipMemcpyDeviceToHost));
        if (FILE *f = std::fopen(ix->tuning.stamps_path.c_str(), "wb")) {
            std::fwrite(st.data(), 8, st.size(), f);
            std::fclose(f);
        }
    }
    return ISSL_OK;
}

// Synchronous batch on the caller's stream.
static int score_core(issl_index *ix, const uint64_t *d_guides, size_t n, int max_dist, double threshold, int method,
                      double *d_mit, double *d_cfd, hipStream_t stream, bool dump)
{
    int rc = finish_batches(ix, stream); // anything enqueued asynchronously before
    if (rc) return rc;
    ix->stats = issl_stats{};
    ix->stats.n_guides = n;
    if (n == 0) return ISSL_OK;
    for (int attempt = 0;; ++attempt) {
        rc = enqueue_batch(ix, ix->lane, stream, d_guides, n, max_dist, threshold, method, d_mit, d_cfd, dump, true);
        if (rc) return rc;
        rc = finish_batches(ix, stream);
        if (rc == ISSL_OK) {
            ix->stats.scan_launches = attempt + 1;
            return ISSL_OK;
        }
        if (rc != ISSL_E_RETRY) return rc;
        if (attempt >= 6) {
            set_error("internal error: raw record buffer kept overflowing");
            return ISSL_E_DEVICE;
        }
    }
}

extern "C" {

const char *issl_last_error(void) { return get_error(); }
int issl_abi_version(void) { return ISSL_ABI_VERSION; }

int issl_index_open(const char *path, issl_index **out)
{
    if (!path || !out) { set_error("null argument"); return ISSL_E_ARG; }
    std::unique_ptr<HostIndex> h(new (std::nothrow) HostIndex());
    if (!h) { set_error("out of memory"); return ISSL_E_NOMEM; }
    int rc = h->open_file(path);
    if (rc) return rc;
    return new_index_from_host(std::move(h), out);
}

int issl_index_from_memory(const void *image, size_t len, issl_index **out)
{
    if (!image || !out) { set_error("null argument"); return ISSL_E_ARG; }
    std::unique_ptr<HostIndex> h(new (std::nothrow) HostIndex());
    if (!h) { set_error("out of memory"); return ISSL_E_NOMEM; }
    int rc = h->from_memory(image, len);
    if (rc) return rc;
    return new_index_from_host(std::move(h), out);
}

int issl_index_build_from_text(const char *text, size_t n_lines, size_t seq_len, size_t slice_width,
                               issl_index **out)
{
    if (!text || !out) { set_error("null argument"); return ISSL_E_ARG; }
    std::unique_ptr<HostIndex> h(new (std::nothrow) HostIndex());
    if (!h) { set_error("out of memory"); return ISSL_E_NOMEM; }
    int rc = h->build_from_text(text, n_lines, seq_len, slice_width);
    if (rc) return rc;
    return new_index_from_host(std::move(h), out);
}

int issl_index_build_from_sites(const uint64_t *sigs, const uint32_t *occ, size_t n_sites, size_t n_lines,
                                size_t seq_len, size_t slice_width, issl_index **out)
{
    if (!sigs || !occ || !out) { set_error("null argument"); return ISSL_E_ARG; }
    std::unique_ptr<HostIndex> h(new (std::nothrow) HostIndex());
    if (!h) { set_error("out of memory"); return ISSL_E_NOMEM; }
    int rc = h->build_from_sites(sigs, occ, n_sites, n_lines, seq_len, slice_width);
    if (rc) return rc;
    return new_index_from_host(std::move(h), out);
}

int issl_index_write(const issl_index *idx, const char *path)
{
    if (!idx || !path) { set_error("null argument"); return ISSL_E_ARG; }
    if (!idx->host) { set_error("index was attached from a device image and has no host arrays"); return ISSL_E_STATE; }
    if (idx->host->has_arrays()) return idx->host->write_file(path);
    // built on the device: header, score table and bucket sizes come from the host side, sites and slice lists are
    // streamed out of the HBM image
    if (!idx->d_image) { set_error("index has neither host arrays nor a device image"); return ISSL_E_STATE; }
    HIP_TRY(hipSetDevice(idx->device));
    FILE *fp = std::fopen(path, "wb");
    if (!fp) {
        set_error(std::string("cannot write index file '") + path + "': " + std::strerror(errno));
        return ISSL_E_IO;
    }
    bool ok = idx->host->write_leading_sections(fp) == ISSL_OK;
    const uint8_t *base = static_cast<const uint8_t *>(idx->d_image);
    const uint8_t *cold = static_cast<const uint8_t *>(idx->h_cold);
    std::vector<uint8_t> stage;
    bool sig_words = false;
    auto stream_dev = [&](const uint8_t *src, uint64_t bytes) { // device memory, through a 64 MiB staging buffer
        stage.resize(size_t(64) << 20);
        for (uint64_t at = 0; ok && at < bytes; at += stage.size()) {
            const size_t len = static_cast<size_t>(std::min<uint64_t>(stage.size(), bytes - at));
            ok = hipMemcpy(stage.data(), src + at, len, hipMemcpyDeviceToHost) == hipSuccess;
            if (ok && sig_words) { // the site table of a sorted image carries a copy of the counts above the signatures
                uint64_t *w = reinterpret_cast<uint64_t *>(stage.data());
                for (size_t i = 0; i < len / 8; ++i) w[i] &= kSigMask;
            }
            ok = ok && std::fwrite(stage.data(), 1, len, fp) == len;
        }
    };
    auto stream_out = [&](uint64_t off, uint64_t bytes, bool in_host) {
        if (in_host) { // the section already sits in host memory
            ok = ok && std::fwrite(cold + off, 1, bytes, fp) == bytes;
            return;
        }
        stream_dev(base + off, bytes);
    };
    sig_words = idx->hdr.off_sub_start != 0;
    stream_out(idx->hdr.off_sites, 8 * idx->geo.n_sites, (idx->hdr.cold_on_host & 2u) != 0);
    sig_words = false;
    ok = ok && std::fwrite(idx->host->sizes, 8, idx->geo.n_buckets(), fp) == idx->geo.n_buckets();
    if (idx->hdr.lists_absent) {
        // The image holds no slice lists: on a sorted layout they are a function of the site table and the counts -- the
        // stable counting sort of isslCreateIndex.cpp:218-234 --, made again here, one slice at a time.
        DevTemp list_mem;
        const uint64_t n = idx->geo.n_sites;
        if (hipMalloc(&list_mem.p, std::max<uint64_t>(8 * n, 8)) != hipSuccess) {
            (void)hipGetLastError();
            list_mem.p = nullptr;
            std::fclose(fp);
            set_error("no device memory to rebuild the slice lists of this image (8 B per site)");
            return ISSL_E_DEVICE;
        }
        for (uint64_t sl = 0; ok && sl < idx->geo.n_slices; ++sl) {
            int brc = launch_build_entries(reinterpret_cast<const uint64_t *>(base + idx->hdr.off_sites),
                                           reinterpret_cast<const uint32_t *>(base + idx->hdr.off_site_occ), n, static_cast<uint32_t>(sl),
                                           static_cast<uint32_t>(sl + 1), static_cast<uint32_t>(idx->geo.slice_width),
                                           static_cast<uint64_t *>(list_mem.p));
            if (brc) { std::fclose(fp); return brc; }
            stream_dev(static_cast<const uint8_t *>(list_mem.p), 8 * n);
        }
    } else {
        stream_out(idx->hdr.off_entries, 8 * idx->geo.n_sites * idx->geo.n_slices, (idx->hdr.cold_on_host & 1u) != 0);
    }
    ok = (std::fclose(fp) == 0) && ok;
    if (!ok) {
        set_error(std::string("could not write '") + path + "' from the device image");
        return ISSL_E_IO;
    }
    return ISSL_OK;
}

int issl_index_header(const issl_index *idx, issl_header *out)
{
    if (!idx || !out) { set_error("null argument"); return ISSL_E_ARG; }
    out->n_sites = idx->geo.n_sites;
    out->seq_len = idx->geo.seq_len;
    out->n_lines = idx->geo.n_lines;
    out->slice_width = idx->geo.slice_width;
    out->n_slices = idx->geo.n_slices;
    out->n_scores = idx->geo.n_scores;
    return ISSL_OK;
}

int issl_index_bucket_sizes(const issl_index *idx, uint64_t *out, size_t n)
{
    if (!idx || !out) { set_error("null argument"); return ISSL_E_ARG; }
    if (n < idx->bucket_sizes.size()) { set_error("bucket size buffer too small"); return ISSL_E_ARG; }
    std::copy(idx->bucket_sizes.begin(), idx->bucket_sizes.end(), out);
    return ISSL_OK;
}

int issl_index_close(issl_index *idx)
{
    if (!idx) return ISSL_OK;
    release_device(idx);
    delete idx;
    return ISSL_OK;
}

int issl_index_device_bytes(const issl_index *idx, size_t *out)
{
    if (!idx || !out) { set_error("null argument"); return ISSL_E_ARG; }
    if (idx->d_image) { *out = idx->hdr.total_bytes; return ISSL_OK; }
    if (!idx->host) { set_error("index has neither host arrays nor a device image"); return ISSL_E_STATE; }
    int rc = supported_geometry(idx->geo);
    if (rc) return rc;
    std::vector<uint64_t> m;
    std::vector<double> v;
    idx->host->unique_scores(m, v);
    ImageHeader h;
    // the layout an upload tries first (issl_index_upload falls back to smaller ones when the HBM is short)
    const std::vector<LayoutSpec> choices = layout_choices(idx->tuning, idx->geo, idx->list_order_only);
    if (choices.empty()) { set_error("the layout options of this index contradict each other"); return ISSL_E_ARG; }
    layout_image(h, idx->geo, m.size(), count_tiles(*idx->host), masks_are_dense(m), choices.front());
    *out = h.total_bytes;
    return ISSL_OK;
}

int issl_index_set_option(issl_index *idx, const char *key, const char *value)
{
    if (!idx || !key || !value) { set_error("null argument"); return ISSL_E_ARG; }
    if (idx->n_pending) { set_error("issl_index_set_option: batches are in flight, call issl_score_finish first"); return ISSL_E_STATE; }
    Tuning t = idx->tuning;
    if (!t.set(key, value)) {
        set_error(std::string("unknown option or value out of range: ") + key + "=" + value);
        return ISSL_E_ARG;
    }
    if (idx->d_image && t.scan_blocks != idx->tuning.scan_blocks) {
        // every scan wave owns the raw chunk with its own number: keep at least that many
        HIP_TRY(hipSetDevice(idx->device));
        for (Lane *lp : {&idx->lane, &idx->lane2})
            if (lp->ws.cap_chunks && lp->ws.cap_chunks < size_t(scan_waves(t)) * 2) {
                int rc = ensure_raw_capacity(lp->ws, size_t(scan_waves(t)) * 4);
                if (rc) return rc;
            }
    }
    idx->tuning = t;
    return ISSL_OK;
}

int issl_index_get_option(const issl_index *idx, const char *key, long long *value)
{
    if (!idx || !key || !value) { set_error("null argument"); return ISSL_E_ARG; }
    const Tuning &t = idx->tuning;
    const std::string k(key);
    if (k == "scan_blocks") *value = t.scan_blocks;
    else if (k == "scan_threads") *value = t.scan_threads;
    else if (k == "upload_chunk_kib") *value = static_cast<long long>(t.upload_chunk_kib);
    else if (k == "upload_ring_min_kib") *value = static_cast<long long>(t.upload_ring_min_kib);
    else if (k == "upload_threads") *value = t.upload_threads;
    else if (k == "item_guides") *value = t.item_guides;
    else if (k == "scan_generic") *value = t.scan_generic;
    else if (k == "stage_timing") *value = t.stage_timing;
    else if (k == "scan_events") *value = t.scan_events;
    else if (k == "raw_chunks") *value = static_cast<long long>(t.raw_chunks);
    else if (k == "inline_sigs") *value = t.inline_sigs;
    else if (k == "host_cold") *value = t.host_cold;
    else if (k == "sorted_layout") *value = t.sorted_layout;
    else if (k == "compact") *value = t.compact;
    else if (k == "keep_lists") *value = t.keep_lists;
    else if (k == "lists_absent") *value = idx->d_image ? static_cast<long long>(idx->hdr.lists_absent) : -1; // read-only
    else if (k == "prune") *value = t.prune;
    else if (k == "lanes") *value = t.lanes;
    else if (k == "tail_shapes") *value = t.tail_shapes;
    else if (k == "hit_slots") *value = t.hit_slots;
    else if (k == "lean_tail") *value = t.lean_tail;
    else if (k == "small_bin") *value = t.small_bin;
    else if (k == "expect_guides") *value = static_cast<long long>(t.expect_guides);
    else if (k == "fine_items") *value = static_cast<long long>(t.fine_items);
    else if (k == "is_sorted") *value = idx->d_image ? ((idx->hdr.off_srec || idx->hdr.off_sid) ? 1 : 0) : -1; // read-only
    else if (k == "is_compact") *value = idx->d_image ? (idx->hdr.off_sid ? 1 : 0) : -1;                   // read-only
    else if (k == "cold_on_host") *value = idx->d_image ? (idx->hdr.cold_on_host ? 1 : 0) : -1;            // read-only: layout in use
    else if (k == "cold_sections") *value = idx->d_image ? static_cast<long long>(idx->hdr.cold_on_host) : -1; // read-only: 0, 1 (lists), 3 (lists + sites)
    else if (k == "dense_mit") *value = idx->d_image ? (idx->hdr.off_mit_dense ? 1 : 0) : -1;              // read-only
    else if (k == "has_inline_sigs") *value = idx->d_image ? (idx->hdr.off_esig ? 1 : 0) : -1;            // read-only
    else { set_error(std::string("unknown option: ") + key); return ISSL_E_ARG; }
    return ISSL_OK;
}

static int parse_build_options(issl_index *ix, const char *options);

static int upload_common(issl_index *idx, int device, void *buf, size_t bytes, const DeviceBuildInput *dbi = nullptr)
{
    if (!idx) { set_error("null argument"); return ISSL_E_ARG; }
    if (!idx->host) { set_error("index has no host arrays to upload"); return ISSL_E_STATE; }
    if (!dbi && !idx->host->has_arrays()) {
        // built on the device: its arrays exist only in that image
        if (idx->d_image && idx->device == device && !buf) return ISSL_OK;
        set_error("index was built on the device and has no host arrays: replicate its image with issl_index_image + "
                  "issl_index_attach_image");
        return ISSL_E_STATE;
    }
    int rc = supported_geometry(idx->geo);
    if (rc) return rc;
    double t0 = wall_ms();
    rc = select_device(device);
    if (rc) return rc;
    release_device(idx);
    (void)hipFree(nullptr); // creates the context
    upload_note(idx, "device runtime start", t0);
    t0 = wall_ms();
    std::vector<uint64_t> m;
    std::vector<double> v;
    idx->host->unique_scores(m, v);
    const Tuning &tn = idx->tuning;
    const std::vector<LayoutSpec> choices = layout_choices(tn, idx->geo, idx->list_order_only);
    idx->device = device;
    const uint64_t n_tiles = count_tiles(*idx->host);
    const bool dense = masks_are_dense(m);
    std::string why = "the layout options of this index contradict each other";
    for (const LayoutSpec &c : choices) {
        layout_image(idx->hdr, idx->geo, m.size(), n_tiles, dense, c);
        // temporary device memory: while a list-order host-cold image is packed, signatures + one slice list; for the
        // sorted layouts two key arrays of 8 B per site (one slice at a time) and, lists in host memory, one slice list
        const uint64_t ns = idx->geo.n_sites;
        const uint64_t temp = c.sorted ? 8 * ns + (64ull << 20) // (keys and slice list live in the image's scan section while it is built)
                              : (c.cold ? 16 * ns : 0) + ns * idx->geo.n_slices / 8 + 8; // (list order: + the `seen` bitmap)
        if (buf) {
            if (bytes < idx->hdr.total_bytes || (reinterpret_cast<uintptr_t>(buf) & 255u)) {
                why = "device buffer too small or not 256-byte aligned";
                continue;
            }
            idx->d_image = buf;
            idx->owns_image = false;
        } else {
            // leave room for the scoring workspace: the larger of 2 GiB and 3 % of the device
            size_t free_b = 0, total_b = 0;
            HIP_TRY(hipMemGetInfo(&free_b, &total_b)); // (nothing allocated yet on this turn of the loop)
            const uint64_t reserve = std::max<uint64_t>(uint64_t(2) << 30, total_b / 32);
            if (idx->hdr.total_bytes + temp + reserve > free_b) {
                why = "the image (" + std::to_string(idx->hdr.total_bytes >> 20) + " MiB) does not fit the free device memory (" +
                      std::to_string(free_b >> 20) + " MiB)";
                continue;
            }
            if (hipMalloc(&idx->d_image, idx->hdr.total_bytes) != hipSuccess) {
                (void)hipGetLastError();
                idx->d_image = nullptr;
                why = "hipMalloc of the image failed";
                continue;
            }
            idx->owns_image = true;
        }
        if (c.cold) {
            // portable + mapped: every device of the node can read the one host copy (issl_node)
            if (hipHostMalloc(&idx->h_cold, std::max<uint64_t>(idx->hdr.cold_bytes, 256), hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
                (void)hipGetLastError();
                idx->h_cold = nullptr;
                if (idx->owns_image) (void)hipFree(idx->d_image);
                idx->d_image = nullptr;
                idx->owns_image = false;
                why = "cannot pin " + std::to_string(idx->hdr.cold_bytes >> 20) + " MiB of host memory for the cold sections";
                continue;
            }
            idx->owns_cold = true;
            if (hipHostGetDevicePointer(&idx->d_cold, idx->h_cold, 0) != hipSuccess) {
                (void)hipGetLastError();
                release_device(idx);
                set_error("HIP error: the pinned host buffer of the cold sections has no device address");
                return ISSL_E_DEVICE;
            }
        }
        upload_note(idx, c.cold ? "layout + allocation (cold sections in pinned host memory)" : "layout + allocation", t0);
        // expect_guides: the caller will score a batch of about this many guides right after the upload (the one-shot scorer
        // knows its page).  The scoring workspace -- streams, events, the buffers of a small batch -- is set up on a thread of
        // its own while the file's sections are on their way (the upload is bound by the PCIe link): 20 ms less in front of a
        // one-shot process's first kernel.  Only where image and temporaries leave the device half empty; a failure here is
        // the first scoring call's to report.
        std::thread prep;
        if (tn.expect_guides && !buf) {
            size_t free_now = 0, total_now = 0;
            if (hipMemGetInfo(&free_now, &total_now) == hipSuccess && free_now > temp + (size_t(24) << 30) + total_now / 2)
                prep = std::thread([idx] { // (reads the header the layout has just made; the view is finish_upload's)
                    (void)hipSetDevice(idx->device);
                    // (for at most 32 k guides: the streams, the events, the small buffers -- the first call's fixed 20 ms.  The
                    // gigabytes a page of a million guides needs are allocated by the scoring call itself: beside the upload they
                    // held its copies up for as long as they took, 27 ms moved from one stage to the other)
                    const size_t n = std::min<size_t>(idx->tuning.expect_guides, size_t(1) << 15);
                    if (ensure_workspace(idx, n, idx->lane) != ISSL_OK) (void)hipGetLastError();
                });
            else
                (void)hipGetLastError();
        }
        rc = finish_upload(idx, dbi);
        if (prep.joinable()) prep.join();
        if (rc == ISSL_OK) return ISSL_OK;
        release_device(idx);
        if (rc == kSortNoRoom) { // the temporaries of the sort did not fit after all: the next, smaller layout
            why = "no device memory for the temporaries of the sorted layout";
            t0 = wall_ms();
            continue;
        }
        if (rc == kSortNeedsListOrder) {
            // (keep_lists=0 forces a sorted layout like the other two -- only a sorted image can do without its lists --, and an
            // index that has been here before is not sent round again: the second turn would end where the first did)
            if (tn.sorted_layout == 1 || tn.compact == 1 || tn.keep_lists == 0 || idx->list_order_only) {
                set_error("this index cannot take the sorted layout that was asked for: a list is not ascending by site id, "
                          "holds a site in a bucket its signature does not select, or carries different counts for one site");
                return ISSL_E_UNSUPPORTED;
            }
            idx->list_order_only = true; // once more, with the stream in list order
            return upload_common(idx, device, buf, bytes, dbi);
        }
        return rc;
    }
    set_error("cannot place the index image: " + why);
    return buf ? ISSL_E_ARG : ISSL_E_DEVICE;
}

int issl_index_build_on_device_opt(const uint64_t *sigs, const uint32_t *occ, size_t n_sites, size_t n_lines,
                                   size_t seq_len, size_t slice_width, int device, const char *options, issl_index **out)
{
    if (!sigs || !occ || !out) { set_error("null argument"); return ISSL_E_ARG; }
    std::unique_ptr<HostIndex> h(new (std::nothrow) HostIndex());
    if (!h) { set_error("out of memory"); return ISSL_E_NOMEM; }
    int rc = h->init_without_arrays(sigs, n_sites, n_lines, seq_len, slice_width);
    if (rc) return rc;
    issl_index *ix = nullptr;
    rc = new_index_from_host(std::move(h), &ix);
    if (rc) return rc;
    rc = parse_build_options(ix, options);
    if (rc) { issl_index_close(ix); return rc; }
    const DeviceBuildInput dbi{sigs, occ, false};
    rc = upload_common(ix, device, nullptr, 0, &dbi);
    if (rc) {
        issl_index_close(ix);
        return rc;
    }
    *out = ix;
    return ISSL_OK;
}

static int parse_build_options(issl_index *ix, const char *options)
{
    for (std::string rest = options ? options : ""; !rest.empty();) { // "key=value,key=value"
        const size_t comma = rest.find(',');
        const std::string item = rest.substr(0, comma);
        rest = comma == std::string::npos ? std::string() : rest.substr(comma + 1);
        const size_t eq = item.find('=');
        if (eq == std::string::npos || !ix->tuning.set(item.substr(0, eq).c_str(), item.substr(eq + 1).c_str())) {
            set_error("unknown option or value out of range: " + item);
            return ISSL_E_ARG;
        }
    }
    return ISSL_OK;
}

int issl_index_build_from_device_sites(const uint64_t *d_sigs, const uint32_t *d_occ, size_t n_sites, size_t n_lines,
                                       size_t seq_len, size_t slice_width, int device, const char *options, issl_index **out)
{
    if (!d_sigs || !d_occ || !out) { set_error("null argument"); return ISSL_E_ARG; }
    if (seq_len == 0 || seq_len > 32 || slice_width < 2 || slice_width > 8 || (seq_len * 2) / slice_width == 0 ||
        (seq_len * 2) / slice_width > kMaxSlices) {
        set_error("bad sequence length or slice width");
        return ISSL_E_ARG;
    }
    int rc = select_device(device);
    if (rc) return rc;
    const uint32_t n_slices = static_cast<uint32_t>((seq_len * 2) / slice_width);
    std::vector<uint64_t> sizes(size_t(n_slices) << slice_width);
    rc = launch_bucket_sizes(d_sigs, n_sites, static_cast<uint32_t>(slice_width), n_slices, sizes.data());
    if (rc) return rc;
    std::unique_ptr<HostIndex> h(new (std::nothrow) HostIndex());
    if (!h) { set_error("out of memory"); return ISSL_E_NOMEM; }
    rc = h->init_from_bucket_sizes(sizes.data(), n_sites, n_lines, seq_len, slice_width);
    if (rc) return rc;
    issl_index *ix = nullptr;
    rc = new_index_from_host(std::move(h), &ix);
    if (rc) return rc;
    rc = parse_build_options(ix, options);
    if (rc) { issl_index_close(ix); return rc; }
    const DeviceBuildInput dbi{d_sigs, d_occ, true};
    rc = upload_common(ix, device, nullptr, 0, &dbi);
    if (rc) {
        issl_index_close(ix);
        return rc;
    }
    *out = ix;
    return ISSL_OK;
}

int issl_index_build_on_device(const uint64_t *sigs, const uint32_t *occ, size_t n_sites, size_t n_lines,
                               size_t seq_len, size_t slice_width, int device, issl_index **out)
{
    return issl_index_build_on_device_opt(sigs, occ, n_sites, n_lines, seq_len, slice_width, device, nullptr, out);
}

int issl_device_memory(int device, size_t *free_bytes, size_t *total_bytes)
{
    if (!free_bytes || !total_bytes) { set_error("null argument"); return ISSL_E_ARG; }
    int rc = select_device(device);
    if (rc) return rc;
    HIP_TRY(hipMemGetInfo(free_bytes, total_bytes));
    return ISSL_OK;
}

int issl_index_upload(issl_index *idx, int device) { return upload_common(idx, device, nullptr, 0); }

int issl_index_upload_into(issl_index *idx, int device, void *dev_buf, size_t bytes)
{
    if (!dev_buf) { set_error("null device buffer"); return ISSL_E_ARG; }
    return upload_common(idx, device, dev_buf, bytes);
}

static int attach_common(int device, void *dev_buf, size_t bytes, void *cold_host, size_t cold_bytes, issl_index **out)
{
    if (!dev_buf || !out) { set_error("null argument"); return ISSL_E_ARG; }
    int rc = select_device(device);
    if (rc) return rc;
    if (bytes < kHeaderBytes || (reinterpret_cast<uintptr_t>(dev_buf) & 255u)) {
        set_error("device image too small or not 256-byte aligned");
        return ISSL_E_ARG;
    }
    ImageHeader h;
    HIP_TRY(hipMemcpy(&h, dev_buf, sizeof h, hipMemcpyDeviceToHost));
    if (h.magic != kImageMagic || h.version != kImageVersion || h.tile_cands != kTileCands ||
        h.total_bytes > bytes) {
        set_error("device buffer does not hold an ISSL image of this library version");
        return ISSL_E_FORMAT;
    }
    void *d_cold = nullptr;
    if (h.cold_on_host) {
        if (!cold_host || cold_bytes < h.cold_bytes) {
            set_error("this image keeps its cold sections (sites, slice lists) in pinned host memory: attach it with "
                      "issl_index_attach_image_cold and the buffer of issl_index_cold");
            return ISSL_E_STATE;
        }
        HIP_TRY(hipHostGetDevicePointer(&d_cold, cold_host, 0));
    }
    issl_index *ix = new (std::nothrow) issl_index();
    if (!ix) { set_error("out of memory"); return ISSL_E_NOMEM; }
    ix->geo.n_sites = h.n_sites;
    ix->geo.seq_len = h.seq_len;
    ix->geo.n_lines = h.n_lines;
    ix->geo.slice_width = h.slice_width;
    ix->geo.n_slices = h.n_slices;
    ix->geo.n_scores = h.n_scores_file;
    ix->hdr = h;
    ix->device = device;
    ix->d_image = dev_buf;
    ix->owns_image = false;
    ix->h_cold = h.cold_on_host ? cold_host : nullptr;
    ix->d_cold = d_cold;
    ix->owns_cold = false;
    ix->view = make_view(h, dev_buf, d_cold);
    std::vector<uint64_t> bstart(h.n_buckets + 1);
    hipError_t e = hipMemcpy(bstart.data(), static_cast<uint8_t *>(dev_buf) + h.off_bucket_start,
                             8 * (h.n_buckets + 1), hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        delete ix;
        set_error(std::string("HIP error: ") + hipGetErrorString(e));
        return ISSL_E_DEVICE;
    }
    ix->bucket_sizes.resize(h.n_buckets);
    for (uint64_t b = 0; b < h.n_buckets; ++b) ix->bucket_sizes[b] = bstart[b + 1] - bstart[b];
    *out = ix;
    return ISSL_OK;
}

int issl_index_attach_image(int device, void *dev_buf, size_t bytes, issl_index **out)
{
    return attach_common(device, dev_buf, bytes, nullptr, 0, out);
}

int issl_index_attach_image_cold(int device, void *dev_buf, size_t bytes, void *cold_host, size_t cold_bytes,
                                 issl_index **out)
{
    return attach_common(device, dev_buf, bytes, cold_host, cold_bytes, out);
}

int issl_index_cold(const issl_index *idx, void **host_ptr, size_t *bytes)
{
    if (!idx || !host_ptr || !bytes) { set_error("null argument"); return ISSL_E_ARG; }
    if (!idx->d_image) { set_error("index has no device image"); return ISSL_E_STATE; }
    *host_ptr = idx->hdr.cold_on_host ? idx->h_cold : nullptr;
    *bytes = idx->hdr.cold_on_host ? idx->hdr.cold_bytes : 0;
    return ISSL_OK;
}

int issl_index_image(const issl_index *idx, void **dev_ptr, size_t *bytes)
{
    if (!idx || !dev_ptr || !bytes) { set_error("null argument"); return ISSL_E_ARG; }
    if (!idx->d_image) { set_error("index has no device image"); return ISSL_E_STATE; }
    *dev_ptr = idx->d_image;
    *bytes = idx->hdr.total_bytes;
    return ISSL_OK;
}

int issl_index_copy_image_to(const issl_index *idx, void *dev_dst, size_t bytes)
{
    if (!idx || !dev_dst) { set_error("null argument"); return ISSL_E_ARG; }
    if (!idx->d_image) { set_error("index has no device image"); return ISSL_E_STATE; }
    if (bytes < idx->hdr.total_bytes || (reinterpret_cast<uintptr_t>(dev_dst) & 255u)) {
        set_error("destination too small or not 256-byte aligned");
        return ISSL_E_ARG;
    }
    HIP_TRY(hipSetDevice(idx->device));
    HIP_TRY(hipMemcpy(dev_dst, idx->d_image, idx->hdr.total_bytes, hipMemcpyDeviceToDevice));
    return ISSL_OK;
}

int issl_encode_guides(const char *text, size_t n, size_t seq_len, size_t stride, uint64_t *out)
{
    if ((!text && n) || !out || seq_len == 0 || seq_len > 32 || stride < seq_len) {
        set_error("bad argument to issl_encode_guides");
        return ISSL_E_ARG;
    }
    for (size_t i = 0; i < n; ++i) out[i] = encode_guide(text + i * stride, seq_len);
    return ISSL_OK;
}

int issl_decode_guide(uint64_t sig, size_t seq_len, char *out)
{
    if (!out || seq_len == 0 || seq_len > 32) { set_error("bad argument to issl_decode_guide"); return ISSL_E_ARG; }
    decode_guide(sig, seq_len, out);
    return ISSL_OK;
}

// (issl_read_query_file, issl_format_scores: issl_text.cpp)

void issl_free(void *p) { std::free(p); }

int issl_method_from_string(const char *s) { return method_from_string(s); }

int issl_score_device(issl_index *idx, const uint64_t *d_guides, size_t n, int max_dist, double threshold,
                      int method, double *d_mit, double *d_cfd, void *stream)
{
    if (!idx || (n && (!d_guides || !d_mit || !d_cfd))) { set_error("null argument"); return ISSL_E_ARG; }
    return score_core(idx, d_guides, n, max_dist, threshold, method, d_mit, d_cfd, static_cast<hipStream_t>(stream),
                      false);
}

int issl_score_device_async(issl_index *idx, const uint64_t *d_guides, size_t n, int max_dist, double threshold,
                            int method, double *d_mit, double *d_cfd, void *stream)
{
    if (!idx || (n && (!d_guides || !d_mit || !d_cfd))) { set_error("null argument"); return ISSL_E_ARG; }
    if (!idx->d_image) { set_error("index has no device image: call issl_index_upload first"); return ISSL_E_STATE; }
    if (n == 0) return ISSL_OK;
    if (n > kMaxBatch) { // before any workspace is sized for it
        set_error("at most 2^24 guides per device batch (issl_score splits larger batches itself)");
        return ISSL_E_ARG;
    }
    HIP_TRY(hipSetDevice(idx->device));
    // lanes option = 2: consecutive batches use different workspaces and streams, so that the short, latency-bound
    // kernels behind one batch's scan run in the wave slots the next batch's scan leaves free: +9-11 % guides/s at 100 k
    // guides x 300 M sites; every kernel then shares the chip and takes longer, which is why it is not the default
    Lane &lane = (idx->tuning.lanes >= 2 && (idx->n_async++ & 1u)) ? idx->lane2 : idx->lane;
    int rc = ensure_workspace(idx, n, lane); // creates the internal stream on first use
    if (rc) return rc;
    if (stream) { // inputs are produced on the caller's stream: the batch starts after what is enqueued there now
        HIP_TRY(hipEventRecord(lane.ev[0], static_cast<hipStream_t>(stream)));
        HIP_TRY(hipStreamWaitEvent(lane.stream, lane.ev[0], 0));
    }
    return enqueue_batch(idx, lane, lane.stream, d_guides, n, max_dist, threshold, method, d_mit, d_cfd, false,
                         idx->tuning.stage_timing, idx->tuning.lanes);
}

int issl_score_wait(issl_index *idx, void *stream)
{
    if (!idx) { set_error("null argument"); return ISSL_E_ARG; }
    if (idx->device >= 0) HIP_TRY(hipSetDevice(idx->device));
    for (Lane *lp : {&idx->lane, &idx->lane2})
        if (lp->ready && lp->pending) {
            if (!lp->done_recorded) { // the end of the lane's last batch, recorded now: everything enqueued on its stream so far
                HIP_TRY(hipEventRecord(lp->done, lp->last_tail));
                lp->done_recorded = true;
            }
            HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream), lp->done, 0));
        }
    return ISSL_OK;
}

int issl_score_finish(issl_index *idx, void *stream)
{
    if (!idx) { set_error("null argument"); return ISSL_E_ARG; }
    return finish_batches(idx, static_cast<hipStream_t>(stream));
}

int issl_score(issl_index *idx, const uint64_t *guides, size_t n, int max_dist, double threshold, int method,
               double *mit, double *cfd)
{
    if (!idx || (n && (!guides || !mit || !cfd))) { set_error("null argument"); return ISSL_E_ARG; }
    if (!idx->d_image) { set_error("index has no device image: call issl_index_upload first"); return ISSL_E_STATE; }
    if (n == 0) return ISSL_OK;
    HIP_TRY(hipSetDevice(idx->device));
    // Crackling hands over pages of up to 5 M guides (config.ini:112); larger batches go through in pieces of at most
    // 2^22 guides.  The guides are in host memory here, so the comparison count is five table look-ups per guide
    // away (SURVEY 8d cross-check).  Uniform data leaves one raw record per ~26 k comparisons (16 positions, <= 4
    // mismatches); the record buffers (32 B per slot with the sorted keys and score terms) are sized for twice
    // that up front, which saves the first large batch on an index its grow-and-rerun round, and a piece ends
    // early when its estimate would not fit a quarter of the free HBM.  Denser data still grows the buffers.
    // (the pruned scan places every guide in up to 65 groups: pieces of at most 2^20 guides while it may be chosen)
    const uint32_t piece_mode = prune_mode_for(idx->view, idx->tuning, 1, max_dist);
    size_t piece = piece_mode ? size_t(prune_max_guides(piece_mode, idx->view.n_slices)) : size_t(1) << 22;
    // ... and of no more guides than get hit slots (kSlotBytesMax: 512 k): a batch beyond that sends every hit through the
    // grouping pass -- 4 ms per million guides on an even index, where two batches of half a million pay nothing for it
    // (kernels 23.3 -> 21.6 ms) --, in pieces of equal size (a page of 1 M guides: 2 x 500 k, not 512 k + 488 k).
    if (idx->tuning.hit_slots) piece = std::min(piece, kSlotBytesMax / (size_t(kSlotHits) * sizeof(SlotRec)));
    {
        const size_t n_pieces = (n + piece - 1) / piece;
        piece = std::min(piece, (((n + n_pieces - 1) / n_pieces) + 7) & ~size_t(7));
    }
    // Every piece is cut where its estimated records would outgrow the record buffers this handle may have: five table
    // look-ups per guide.  What is skipped on a handle whose buffers already cover a piece is only the question how much
    // memory is free (hipMemGetInfo: asked lazily, once per call, when a piece's estimate first exceeds the buffers in
    // hand) and the call that grows them.  (Small pages: the default buffers do.)
    const bool estimate = !idx->tuning.raw_chunks && n >= (size_t(1) << 15);
    const double records_per_comparison = 8e-5;
    double budget_slots = -1.0; // records the buffers may grow to: the larger of what they hold and a quarter of the free HBM (<= 32 GiB)
    const uint64_t per = idx->geo.buckets_per_slice();
    if (estimate && !idx->worst_per_guide) // the most a guide can be compared with: the longest bucket of every slice
        for (uint64_t sl = 0; sl < idx->geo.n_slices; ++sl)
            idx->worst_per_guide += *std::max_element(idx->bucket_sizes.begin() + sl * per, idx->bucket_sizes.begin() + (sl + 1) * per);
    const size_t wave_chunks = size_t(scan_waves(idx->tuning)) * 10; // every scan wave's own first chunk and the unused tail of its last reservation of up to 16
    issl_stats total{};
    for (size_t at = 0; at < n;) {
        uint64_t cand = 0;
        size_t cnt = 0;
        const size_t most = std::min(piece, n - at);
        const size_t cap_chunks = idx->lane.ws.cap_chunks;
        const double have_slots = static_cast<double>(cap_chunks > wave_chunks ? cap_chunks - wave_chunks : 0) * (kChunkRecs - 1);
        // (an index of even buckets on a handle that has grown its buffers: the bound alone says the piece fits)
        const bool covered = static_cast<double>(most) * static_cast<double>(idx->worst_per_guide) * records_per_comparison <= have_slots ||
                             (most <= idx->proven_guides && cap_chunks >= idx->proven_chunks && idx->proven_chunks > 0 && max_dist <= idx->proven_dist);
        if (!estimate || covered) cnt = most;
        while (estimate && !covered && at + cnt < n && cnt < piece) {
            uint64_t c = 0;
            for (uint64_t sl = 0; sl < idx->geo.n_slices; ++sl)
                c += idx->bucket_sizes[sl * per + ((guides[at + cnt] >> (idx->geo.slice_width * sl)) & (per - 1))];
            const double want = static_cast<double>(cand + c) * records_per_comparison;
            if (want > have_slots) { // beyond the buffers in hand: may they grow that far?
                if (budget_slots < 0.0) {
                    size_t free_b = 0, total_b = 0;
                    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
                    budget_slots = std::max(have_slots, static_cast<double>(std::min<size_t>(free_b / 4, size_t(32) << 30)) / 32.0);
                }
                if (cnt > 0 && want > budget_slots) break;
            }
            cand += c;
            ++cnt;
        }
        Workspace &ws = idx->lane.ws;
        int rc = finish_batches(idx, nullptr); // asynchronous batches may still use the staging buffers
        if (rc) return rc;
        rc = ensure_workspace(idx, cnt, idx->lane);
        if (rc) return rc;
        if (estimate && !covered) {
            double slots = static_cast<double>(cand) * records_per_comparison;
            if (budget_slots >= 0.0) slots = std::min(slots, budget_slots);
            // (a buffer that has to grow grows by a quarter at least: the pieces of a page have estimates a few per cent apart, and
            // every step up is a free and an allocation of gigabytes -- the record buffer and the four arrays sized by it)
            size_t want_chunks = static_cast<size_t>(slots / (kChunkRecs - 1)) + wave_chunks;
            if (want_chunks > ws.cap_chunks && ws.cap_chunks > 0) {
                const size_t roomy = ws.cap_chunks + ws.cap_chunks / 4;
                const size_t most_chunks = budget_slots >= 0.0 ? static_cast<size_t>(budget_slots / (kChunkRecs - 1)) + wave_chunks : roomy;
                want_chunks = std::max(want_chunks, std::min(roomy, std::max(most_chunks, want_chunks)));
            }
            rc = ensure_raw_capacity(ws, want_chunks);
            if (rc) return rc;
        }
        if (ensure_stage(ws, 24 * cnt)) { // guides in, scores out through pinned memory: one DMA each, one synchronisation
            uint64_t *sg = static_cast<uint64_t *>(ws.h_stage);
            double *sm = reinterpret_cast<double *>(sg + cnt), *sc = sm + cnt;
            std::memcpy(sg, guides + at, 8 * cnt);
            HIP_TRY(hipMemcpyAsync(ws.d_guides, sg, 8 * cnt, hipMemcpyHostToDevice, nullptr));
            rc = score_core(idx, ws.d_guides, cnt, max_dist, threshold, method, ws.d_mit, ws.d_cfd, nullptr, false);
            if (rc) return rc;
            HIP_TRY(hipMemcpyAsync(sm, ws.d_mit, 8 * cnt, hipMemcpyDeviceToHost, nullptr));
            HIP_TRY(hipMemcpyAsync(sc, ws.d_cfd, 8 * cnt, hipMemcpyDeviceToHost, nullptr));
            HIP_TRY(hipStreamSynchronize(nullptr));
            std::memcpy(mit + at, sm, 8 * cnt);
            std::memcpy(cfd + at, sc, 8 * cnt);
        } else {
            HIP_TRY(hipMemcpy(ws.d_guides, guides + at, 8 * cnt, hipMemcpyHostToDevice));
            rc = score_core(idx, ws.d_guides, cnt, max_dist, threshold, method, ws.d_mit, ws.d_cfd, nullptr, false);
            if (rc) return rc;
            HIP_TRY(hipMemcpy(mit + at, ws.d_mit, 8 * cnt, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(cfd + at, ws.d_cfd, 8 * cnt, hipMemcpyDeviceToHost));
        }
        const issl_stats &s = idx->stats;
        // (a piece that went through at once: the next ones within its size and distance need no estimate; denser guides than
        // these still take the grow-and-rerun round)
        if (s.scan_launches == 1 && (max_dist > idx->proven_dist || (max_dist == idx->proven_dist && cnt > idx->proven_guides))) {
            idx->proven_dist = max_dist;
            idx->proven_guides = cnt;
            idx->proven_chunks = ws.cap_chunks;
        }
        total.n_guides += s.n_guides; total.candidates += s.candidates; total.hits += s.hits;
        total.planned_comparisons += s.planned_comparisons; total.reference_comparisons += s.reference_comparisons;
        total.pruned = std::max(total.pruned, s.pruned);
        total.scan_tiles += s.scan_tiles; total.ms_bin += s.ms_bin; total.ms_scan += s.ms_scan;
        total.ms_scan_events += s.ms_scan_events;
        total.ms_verify += s.ms_verify; total.ms_group += s.ms_group; total.ms_replay += s.ms_replay;
        total.ms_total += s.ms_total; total.scan_launches += s.scan_launches;
        total.raw_records = std::max(total.raw_records, s.raw_records); total.n_batches += s.n_batches;
        at += cnt;
    }
    idx->stats = total;
    return ISSL_OK;
}

int issl_dump_hits(issl_index *idx, const uint64_t *guides, size_t n, int max_dist, double threshold, int method,
                   issl_hit *hits, size_t cap, size_t *n_hits)
{
    if (!idx || !n_hits || (n && !guides) || (cap && !hits)) { set_error("null argument"); return ISSL_E_ARG; }
    if (!idx->d_image) { set_error("index has no device image: call issl_index_upload first"); return ISSL_E_STATE; }
    *n_hits = 0;
    if (n == 0) return ISSL_OK;
    if (n > (size_t(1) << 22)) { set_error("issl_dump_hits takes at most 2^22 guides per call"); return ISSL_E_ARG; }
    HIP_TRY(hipSetDevice(idx->device));
    int rc = finish_batches(idx, nullptr);
    if (rc) return rc;
    Workspace &ws = idx->lane.ws;
    rc = ensure_workspace(idx, n, idx->lane);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(ws.d_guides, guides, 8 * n, hipMemcpyHostToDevice));
    rc = score_core(idx, ws.d_guides, n, max_dist, threshold, method, ws.d_mit, ws.d_cfd, nullptr, true);
    if (rc) return rc;
    std::vector<uint32_t> goff(n + 1), kept(n);
    HIP_TRY(hipMemcpy(goff.data(), ws.goff, 4 * (n + 1), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(kept.data(), ws.d_kept, 4 * n, hipMemcpyDeviceToHost));
    std::vector<issl_hit> all(goff[n]);
    if (goff[n]) HIP_TRY(hipMemcpy(all.data(), ws.d_hitrec, sizeof(issl_hit) * goff[n], hipMemcpyDeviceToHost));
    size_t total = 0;
    for (size_t g = 0; g < n; ++g) {
        for (uint32_t k = 0; k < kept[g]; ++k) {
            if (total < cap) hits[total] = all[goff[g] + k];
            ++total;
        }
    }
    *n_hits = total;
    return ISSL_OK;
}

int issl_last_stats(const issl_index *idx, issl_stats *out)
{
    if (!idx || !out) { set_error("null argument"); return ISSL_E_ARG; }
    *out = idx->stats;
    return ISSL_OK;
}

int issl_count_candidates(const issl_index *idx, const uint64_t *guides, size_t n, uint64_t *out)
{
    if (!idx || !out || (n && !guides)) { set_error("null argument"); return ISSL_E_ARG; }
    const uint64_t per = idx->geo.buckets_per_slice();
    uint64_t total = 0;
    for (size_t i = 0; i < n; ++i)
        for (uint64_t s = 0; s < idx->geo.n_slices; ++s)
            total += idx->bucket_sizes[s * per + ((guides[i] >> (idx->geo.slice_width * s)) & (per - 1))];
    *out = total;
    return ISSL_OK;
}

// Crackling.py:780-835.  The caller compares float(<"%f" text>) with the threshold, so a score within 1e-6 of the
// threshold is sent through the same text round trip; everything else compares the same way without it.
int issl_verdicts(const double *mit, const double *cfd, size_t n, double threshold, const char *method,
                  uint8_t *accepted)
{
    if (!method || (n && (!mit || !cfd || !accepted))) { set_error("null argument"); return ISSL_E_ARG; }
    const int printed = method_from_string(method); // exact match, :121-143
    const bool has_mit = printed == ISSL_METHOD_MIT || printed == ISSL_METHOD_AND || printed == ISSL_METHOD_OR ||
                         printed == ISSL_METHOD_AVG;
    const bool has_cfd = printed == ISSL_METHOD_CFD || printed == ISSL_METHOD_AND || printed == ISSL_METHOD_OR ||
                         printed == ISSL_METHOD_AVG;
    std::string m(method); // str(...).strip().lower()
    const char *ws = " \t\n\r\f\v";
    const size_t b = m.find_first_not_of(ws);
    m = b == std::string::npos ? std::string() : m.substr(b, m.find_last_not_of(ws) - b + 1);
    for (char &c : m) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
    const int rule = method_from_string(m.c_str());
    auto as_read = [&](double x, bool present) {
        if (!present) return -1.0;
        if (std::fabs(x - threshold) > 1e-5 && rule != ISSL_METHOD_AVG) return x; // text rounding cannot flip it
        char buf[400];
        std::snprintf(buf, sizeof buf, "%f", x);
        return std::strtod(buf, nullptr);
    };
    for (size_t i = 0; i < n; ++i) {
        const double a = as_read(mit[i], has_mit), c = as_read(cfd[i], has_cfd);
        bool reject;
        switch (rule) {
        case ISSL_METHOD_MIT: reject = a < threshold; break;
        case ISSL_METHOD_CFD: reject = c < threshold; break;
        case ISSL_METHOD_AND: reject = a < threshold && c < threshold; break;
        case ISSL_METHOD_OR: reject = a < threshold || c < threshold; break;
        case ISSL_METHOD_AVG: reject = (a + c) / 2 < threshold; break;
        default: accepted[i] = ISSL_VERDICT_NONE; continue;
        }
        accepted[i] = reject ? ISSL_VERDICT_REJECTED : ISSL_VERDICT_ACCEPTED;
    }
    return ISSL_OK;
}

} // extern "C"
