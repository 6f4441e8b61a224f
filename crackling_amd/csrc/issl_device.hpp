// Device-side data model of the ISSL scorer: HBM image layout, workspace, kernel launchers.
// Reference paths are relative to /root/reference.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

#include "issl_host.hpp"

namespace issl {

// ---- HBM image -------------------------------------------------------------------------------
// One contiguous, self-describing device buffer so that it can be produced on one rank and
// broadcast to the others as a single RCCL message.  All section offsets are 256-byte aligned.
//
//   [0,4096)       ImageHeader
//   bucket_start   u64[nb+1]   prefix of the bucket lengths -> index into `entries`
//                              (isslScoreOfftargets.cpp:261-270)
//   tile_first     u32[nb+1]   first scan tile of every bucket (bucket b owns tiles
//                              [tile_first[b], tile_first[b+1]); a tile is kTileCands candidates)
//   score_mask     u64[ns]     sorted unique mismatch masks     (isslScoreOfftargets.cpp:188-197)
//   score_val      f64[ns]     their local MIT scores
//   sites          u64[N]      packed signatures                (:200-204); sorted layouts: | min(count, 2^24-1) << 40 (k_tag_sites)
//   entries        u64[N*S]    bucket contents occ<<32|id       (:235-240)
//   mit_dense      f64[2^20]   local MIT scores indexed by the 20 mismatch flags (when the table allows it)
//   scan           u32[tiles*kTileCands]  the scan stream: for every bucket, in bucket order, the candidate signature
//                              with its own slice removed, 16 positions, bit-sliced (32 candidates per plane word,
//                              plane_word()); each bucket zero-padded to a whole number of tiles
//   sorted layouts (every bucket ordered by the byte of the successor slice; what the pruned scan needs):
//     sub_start    u32[nb*257] inside every bucket, the first stream position of every successor-byte value
//     srec         StreamRec[tiles*kTileCands]  per stream position: signature | 24-bit count, site id, list position; or
//     sid          u32[tiles*kTileCands]        (compact) per stream position: the site id alone
//     site_occ     u32[N]      occurrence count by site id
//   list-order layouts:
//     esig         u64[N*S]    (optional) the signature of the site behind every list entry, in list order
//     occ8         u8[N*S]     (host-cold only) min(occurrences, 255) per list entry
//
// Hot and cold.  The scan reads ONLY `scan` (20 B per site); everything else is read for the ~2e-5 of the comparisons
// that come within max_dist (k_verify, the replay kernels).  Sections may live in mapped, pinned HOST memory
// (`cold_on_host`, layout_choices() in issl_capi.cpp): the slice lists of a compact sorted image -- which scoring never
// reads: 52 B per site stay in HBM, so that every index the format can express (4.29 G sites) is scored from one GPU's
// HBM --, or site table AND lists of a list-order image (25 B per site in HBM: there the candidate's signature is rebuilt
// from its 32 bit planes in the scan stream, the count comes from `occ8`, and host memory is read only for counts >= 255
// and for issl_dump_hits).  Their offsets are then relative to that second buffer and `total_bytes` covers the HBM part.
constexpr uint64_t kImageMagic = 0x314C535349444D41ull; // "AMDISSL1"
constexpr uint32_t kImageVersion = 9;
constexpr uint32_t kTileCands = 2048; // candidates per scan tile: one wave, 64 lanes x 32 registers
constexpr uint32_t kHeaderBytes = 4096;

struct ImageHeader {
    uint64_t magic;
    uint32_t version;
    uint32_t kind; // 0: 20 bp / 8-bit slices / 5 slices, 32-bit scan words
    uint64_t n_sites, seq_len, n_lines, slice_width, n_slices, n_scores_file;
    uint64_t n_buckets;
    uint64_t n_scores_unique;
    uint64_t n_tiles;
    uint64_t tile_cands;
    uint64_t total_bytes;
    uint64_t off_bucket_start, off_tile_first, off_score_mask, off_score_val, off_sites, off_entries,
        off_scan;
    uint64_t off_mit_dense; // 0: absent (table holds masks outside the 20 even bits)
    uint64_t off_esig;      // 0: absent (large indexes: +8 B per list entry do not pay for themselves in HBM)
    uint64_t cold_on_host;  // bit 0: off_entries, bit 1: off_sites are offsets into the pinned host buffer of cold_bytes
                            // bytes (3: the list-order host-cold layout; 1: a compact sorted image whose slice lists
                            // did not fit the HBM; 0: everything in HBM)
    uint64_t cold_bytes;    // bytes of the sections that live in that buffer
    uint64_t off_occ8;      // u8[N*S]: min(occurrences, 255) per list entry; present (in HBM) only when cold_on_host == 3
    // "Sorted" layouts (pruned scan, see below): inside every bucket the scan stream holds the candidates ordered by the
    // byte of the SUCCESSOR slice ((slice + 1) mod n_slices), ties in list order.  0 = absent (stream in list order).
    // They need every list ascending by site id (what isslCreateIndex.cpp:218-234 writes: ids are appended in
    // ascending order) and one occurrence count per site: a hit's place in the reference's scoring order is then
    // (first matching slice, site id) and nothing has to remember list positions.
    uint64_t off_sub_start; // u32[nb * 257]: first stream position (inside the bucket) of every successor-byte value
    uint64_t off_srec;      // StreamRec[n_tiles * 2048]: position in the (tile-padded) scan stream -> the candidate there
                            // (16 B): a noted record names its entry by itself (tile, offset), no table in between; or
    uint64_t off_sid;       // u32[n_tiles * 2048]: ... -> its site id alone (the COMPACT sorted layout: 4 B instead of 16 B per
                            // list entry; signature and count come from `sites` / `site_occ` by id)
    uint64_t off_site_occ;  // u32[N]: occurrence count of every site (sorted layouts)
    uint64_t lists_absent;  // 1: the image holds no slice lists (`entries`), anywhere.  A sorted layout does not need them to
                            // score, and its premises (every list ascending by site id, one count per site, every site in
                            // the bucket its signature selects) make them a function of `sites` and `site_occ`: the stable
                            // counting sort of isslCreateIndex.cpp:218-234, which issl_index_write and issl_dump_hits redo
                            // on the device when asked.  52 B per site in HBM and nothing in host memory: BASELINE
                            // configs[4]'s 3 G lines are 152 GB on one GPU, and the image moves as ONE broadcast.
};
static_assert(sizeof(ImageHeader) <= kHeaderBytes, "header must fit its block");

// Kernel-side view (raw pointers into the image).
// Sorted layout: what k_verify needs to know about the candidate at a stream position, in one 16-byte load.
constexpr uint32_t kOccSaturated = 0xFFFFFFu; // StreamRec (and the site table of the sorted layouts): occurrence counts from here on are looked up
constexpr uint64_t kSigMask = (1ull << 40) - 1ull; // the 20-mer's bits of a signature word

// The successor unit of slice `s`: the four positions (one byte of 2-bit codes) that follow the slice cyclically -- slice
// s + 1 of five 8-bit slices, slices s + 1 and s + 2 of ten 4-bit ones, s + 1 .. s + 4 of twenty 2-bit ones.  The sorted layouts order every bucket by it and
// the pruned scan visits the 13 (1, 67) values within one (no, two) mismatches of a guide's own.  Applied to
// a word of mismatch flags (one flag per position at bit 2p) it yields the flags of those four positions.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t succ_byte(uint64_t sig, uint32_t s, uint32_t slice_width)
{
    const uint32_t sh = (slice_width * (s + 1u)) % 40u;
    const uint64_t x = sig & kSigMask;
    return static_cast<uint32_t>(sh ? (x >> sh) | (x << (40u - sh)) : x) & 0xFFu;
}
struct alignas(16) StreamRec {
    uint64_t sig;  // packed signature of the site (bits 0..39) | min(occurrences, kOccSaturated) << 40
    uint32_t id;   // site id (low half of the list entry)
    uint32_t pos;  // position of the entry in the bucket's list = the reference's iteration order (:344)
};

struct ImageView {
    const uint64_t *bucket_start;
    const uint32_t *tile_first;
    const uint64_t *score_mask;
    const double *score_val;
    const double *mit_dense; // 2^20 doubles indexed by the 20 mismatch flags, or null
    const uint64_t *sites;
    const uint64_t *entries;
    const uint64_t *esig; // signature of the site behind every list entry, in list order, or null
    const uint8_t *occ8;  // min(occurrences, 255) of every list entry, or null (only with host-resident cold sections)
    const uint32_t *sub_start; // sorted layout: [nb][257] stream offsets of the successor-byte groups, or null
    const StreamRec *srec;     // sorted layout: stream position -> candidate, or null
    const uint32_t *sid;       // compact sorted layout: stream position -> site id, or null
    const uint32_t *site_occ;  // sorted layouts: occurrence count by site id, or null
    const uint32_t *scan;
    uint64_t n_sites;
    uint32_t n_buckets;
    uint32_t n_scores;
    uint32_t slice_width;
    uint32_t n_slices;
    uint32_t n_tiles;
};

// Layout computation shared by upload and attach.  Fills every field of `h` from the geometry,
// the number of unique scores and the bucket sizes (sizes may be null when n_tiles is given).
// What an image keeps beside the scan stream.
struct LayoutSpec {
    bool inline_sigs = false; // `esig` (list-order layouts only)
    uint32_t cold = 0;        // ImageHeader::cold_on_host: 0, 1 (slice lists in host memory) or 3 (site table too)
    uint32_t sorted = 0;      // 0: stream in list order; 1: sorted, 16-byte stream records; 2: sorted, compact
    bool no_lists = false;    // ImageHeader::lists_absent (sorted layouts only; cold == 0)
};
void layout_image(ImageHeader &h, const Geometry &g, uint64_t n_scores_unique, uint64_t n_tiles, bool dense_mit,
                  const LayoutSpec &spec);
// `cold`: device-visible address of the pinned host buffer when h.cold_on_host != 0, else ignored.
ImageView make_view(const ImageHeader &h, void *base, void *cold);

// ---- tuning knobs ------------------------------------------------------------------------------
// Read from the environment ONCE, when an index handle is created (never inside a scoring call: several scoring
// threads share the process environment), and changed afterwards only through issl_index_set_option().
struct Tuning {
    uint32_t scan_blocks;   // ISSL_SCAN_BLOCKS   workgroups of the scan launch
    uint32_t scan_threads;  // ISSL_SCAN_THREADS  threads per scan workgroup (64 .. 1024, a multiple of 64; default 1024 = 16 waves: two workgroups
                            //                    per CU = 8 waves per SIMD; 768: 6 per SIMD) -- an occupancy experiment, not a tuning knob
    int upload_threads; // ISSL_UPLOAD_THREADS: readers of the ring below (1..32, default 8)
    size_t upload_chunk_kib, upload_ring_min_kib; // ISSL_UPLOAD_CHUNK_KIB / ISSL_UPLOAD_RING_MIN_KIB: the pinned ring a file-mapped index is uploaded through (FileUploader,
                            //                    issl_capi.cpp): KiB per slot (default 16384) and the section size from which it is used (default 65536); tests
    uint32_t item_guides;   // ISSL_ITEM_GUIDES   guides per scan item (multiple of 8, <= kItemGuides)
    bool scan_generic;      // ISSL_SCAN_GENERIC  force the runtime-threshold build of the scan kernel
    int scan_events;        // ISSL_SCAN_EVENTS   HIP event pair around the scan: 1 every batch, 2 (default) the first batch after a finish, 0 never
    bool stage_timing;      // ISSL_STAGE_TIMING  asynchronous batches record an event at every stage boundary
    bool upload_timing;     // ISSL_UPLOAD_TIMING one stderr line per upload stage
    size_t raw_chunks;      // ISSL_RAW_CHUNKS    initial raw-record buffer in chunks (0: sized from the launch)
    int inline_sigs;        // ISSL_INLINE_SIGS   -1 automatic, 0 never, 1 always
    int sorted_layout;      // ISSL_SORTED_LAYOUT -1 automatic (whenever the lists allow it and an image fits), 0 never, 1 always
    int compact;            // ISSL_COMPACT       the sorted layout with 4 instead of 16 bytes per stream position: -1 when
                            //                    the 16-byte one does not fit the free HBM, 0 never, 1 always (with host_cold=1:
                            //                    the slice lists -- read by issl_dump_hits and issl_index_write only -- in
                            //                    pinned host memory)
    int prune;              // ISSL_PRUNE         scan only the successor-byte groups that can hold a hit (needs the sorted
                            //                    layout and max_dist <= 4): -1 when the plan estimates it to be faster,
                            //                    0 never, 1 whenever possible
    int tail_shapes;        // ISSL_TAIL_SHAPES   1 (default): the short last unit of a successor-byte group runs 2 / 4 guides
                            //                    per pass on 16 / 8 candidates per lane; 0: every unit is a full one (A/B)
    int hit_slots;          // ISSL_HIT_SLOTS     1 (default): Workspace::slot_hits = kSlotHits when the arrays fit (kSlotHitsWide once a batch has shown
                            //                    many guides beyond that); 0: never; 2: kSlotHitsWide from the first batch on (A/B, tests)
    size_t fine_items;      // ISSL_FINE_ITEMS    0 (default): the pruned plan's item list is sized from the index; n: it starts with room for n items
    size_t expect_guides;   // ISSL_EXPECT_GUIDES 0 (default); n: a batch of about n guides follows the upload at once: its workspace is set up beside the upload
    int small_bin;          // ISSL_SMALL_BIN     1 (default): batches of up to 8192 placements (102 guides of five slices) are binned by ONE workgroup
                            // in one launch instead of seven (k_bin_small); 0: always the general kernels
    int lean_tail;          // ISSL_LEAN_TAIL     1 (default): a lane whose batches meet no guide beyond its hit slots enqueues the next ones
                            //                    without the grouping pass and the many-hit replays (Workspace::lean_tail); 0: never (A/B)
    int lanes;              // ISSL_LANES         1|2|3 (default 1; 3: only the binning of a batch beside the batch before it): workspaces + streams that asynchronous batches alternate
                            //                    between (2: the short kernels of one batch fill the wave slots the scan of
                            //                    the next leaves)
    int host_cold;          // ISSL_FORCE_HOST_COLD  -1 automatic (image larger than the free HBM), 0 never, 1 always
    int keep_lists;         // ISSL_KEEP_LISTS    the slice lists of a compact sorted image: -1 dropped when the image with them
                            //                    does not fit the free HBM (ImageHeader::lists_absent), 0 always dropped, 1 always kept
    std::string stamps_path; // ISSL_SCAN_STAMPS  dump per-wave clocks of the scan here (diagnostics)
    static Tuning from_env();
    // Returns false when the key is unknown or the value out of range.
    bool set(const char *key, const char *value);
};

// Slice lists built on the device (issl_build.hip): d_entries[(s - slice_begin) * n_sites + k] for the slices
// [slice_begin, slice_end), from the site signatures and their occurrence counts, both already in device memory.
// Synchronous.
int launch_build_entries(const uint64_t *d_sites, const uint32_t *d_occ, uint64_t n_sites, uint32_t slice_begin,
                         uint32_t slice_end, uint32_t slice_width, uint64_t *d_entries);
// Bucket lengths (isslScoreOfftargets.cpp:221-226: slice-major, n_slices << slice_width of them) of a site table that is
// already in device memory.  Synchronous.
int launch_bucket_sizes(const uint64_t *d_sites, uint64_t n_sites, uint32_t slice_width, uint32_t n_slices, uint64_t *h_sizes);

// Sorted layouts (issl_build.hip): order every bucket's list by the successor slice's byte (two stable radix passes per
// slice over keys built from the signatures) and write the maps of the image.  One slice at a time, 16 B per site of
// temporary device memory (SortTemp, allocated once per upload) -- of which the upload lends the first 8 out of the image
// itself: the scan stream (20 B per site) is packed last, and until then its section holds the keys (and, where the slice
// lists are not kept in HBM, the one list being worked on), so that a sorted image needs 8 B per site beyond its own size
// while it is built.
constexpr int kSortNeedsListOrder = -1000; // the index cannot take a sorted layout: an entry sits in a bucket its signature
                                           // does not select, a list is not ascending by site id, or a site carries
                                           // different counts in different lists (no builder writes any of these; the
                                           // reference does not care, isslScoreOfftargets.cpp:344-348,376) -- the caller
                                           // uploads a list-order layout instead
constexpr int kSortNoRoom = -1001;         // no device memory for the temporaries: the caller tries the next layout
struct SortTemp {
    uint64_t *keys = nullptr, *tmp = nullptr;
    uint32_t *hist = nullptr;
    bool keys_borrowed = false;
    int alloc(uint64_t n_sites, void *borrowed_keys = nullptr); // ISSL_OK or kSortNoRoom; borrowed_keys: 8 B per site somebody else owns
    void release();
    ~SortTemp() { release(); }
};
// Slice `slice`: d_list = its n_sites list entries (occ << 32 | id) in list order, anywhere on the device; writes the
// slice's part of sub_start, of srec OR sid (the other one null) and -- slice 0 -- site_occ (later slices check their
// counts against it).  Asynchronous on the null stream; d_flag: a device word zeroed before the first slice.
int launch_sort_slice(SortTemp &t, const uint64_t *d_sites, const uint64_t *d_list, const uint64_t *d_bucket_start,
                      const uint32_t *d_tile_first, uint64_t n_sites, uint32_t n_slices, uint32_t n_buckets, uint32_t slice_width, uint32_t slice, uint32_t *d_sub_start,
                      StreamRec *d_srec, uint32_t *d_sid, uint32_t *d_site_occ, uint32_t *d_flag);
// Synchronises and reads the flag word: ISSL_OK, ISSL_E_FORMAT (an id beyond the site table), kSortNeedsListOrder.
int finish_sort(uint32_t *d_flag);
void launch_tag_sites(uint64_t *d_sites, const uint32_t *d_site_occ, uint64_t n_sites);

// ---- scoring workspace -------------------------------------------------------------------------
constexpr uint32_t kGuideGroup = 8;    // guide words fetched per scalar load
constexpr uint32_t kItemGuides = 512;  // guides per scan item (bounds one tile's work)
constexpr uint32_t kGuideCost = 4;      // cost of one guide against one full unit of 2048 candidates (32 per lane); against a
                                        // short unit of 16 / 8 candidates per lane it is 2 / 1: two / four guides then share
                                        // every instruction (ScanItem::shape)
constexpr uint32_t kTileFixedCost = 8 * kGuideCost; // cost of a unit beside its passes (ticket, item, fetch, set-up) = eight guide comparisons of a full one (4 until round 4: the
                                                    // 12-position pass is a third cheaper and the unit is not; same-box A/B of 4 / 8 / 14: k_scan 1.40 / 1.37 / 1.38 ms)
constexpr uint32_t kNoGuide = 0xFFFFFFFFu;
constexpr uint32_t kPadGuideWord = 0xFFFFFFFFu; // scan word of padding guide slots: 16 x T, distance 16 from tile padding
constexpr uint32_t kScanGridBlocks = 256u * 4u; // scan launch: 256 CUs x 2 resident workgroups of 16 waves, two rounds
constexpr uint32_t kScanMaxBlocks = 8192u;       // upper bound of the ISSL_SCAN_BLOCKS knob
constexpr uint32_t kScanWaves = kScanMaxBlocks * 16u;
constexpr uint32_t kMaxRanges = kScanMaxBlocks;  // one equal-cost range per workgroup
// ISSL_SCAN_STAMPS diagnostics buffer (u64 words): 4 per scan wave for the largest scan grid, then 16 per guide for the first
// 4096 guides of k_replay_mid, k_replay_big<256> and k_replay_big<1024> -- regions of their own behind the scan's, whatever the grid
constexpr uint32_t kStampsScanWords = 4u * kScanWaves, kStampsPerReplay = 16u * 4096u;
constexpr uint32_t kStampsMid = kStampsScanWords, kStampsBig256 = kStampsMid + kStampsPerReplay, kStampsBig1024 = kStampsBig256 + kStampsPerReplay;
constexpr uint32_t kStampsWords = kStampsBig1024 + kStampsPerReplay;
constexpr uint32_t kSpanRing = 64;              // batches per lane whose scan spans are kept until the next finish
constexpr uint32_t kChunkRecs = 128;            // raw-record chunk: 1 KiB, slot 0 is the fill count
constexpr uint64_t kDeadKey = ~0ull;            // raw slot that did not survive the exact check
// Key of a scored off-target: guide << 37 | first matching slice << 32 | site id (sorted layouts) or position in the
// bucket's list -- the reference's scoring order inside a guide (isslScoreOfftargets.cpp:330,344) is the numeric order
// of the low 37 bits.  Five bits of slice: up to 20 slices (slice width 2).
constexpr uint32_t kKeySliceShift = 32, kKeyGuideShift = 37, kKeySliceMask = 31, kMaxSlices = 20;

// One unit of scan work: a run of tiles of one bucket against one group of guides.  Full scan: all tiles of the bucket
// against <= 512 of the guides whose slice value selects it.  Pruned scan: the tiles that hold one successor-byte group
// of the bucket against the guides that can have a hit there.
struct ScanItem {
    uint32_t bucket; // full scan: bucket; pruned scan: bucket << 8 | successor byte
    uint32_t g0, g1; // range in the grouped guide arrays, g0 % kGuideGroup == 0
    uint32_t n_tiles;
    uint64_t cost0;  // sum of costs of all earlier items; tile cost = (g1-g0)+kTileFixedCost
    uint32_t tile0;  // sum of n_tiles of all earlier items (tiles are numbered in item order)
    uint32_t last_cands; // real candidates in the item's last unit (1..kTileCands): its other units are full
    uint32_t group_abs;  // where the item's first unit starts in the scan stream, in groups of 32 candidates (tile * 64 +
                         // group): a unit is 2048 consecutive candidates of the bucket from there on -- whole tiles for
                         // bucket-level items, a window that may straddle two tiles for the pruned scan's
    uint32_t window;     // candidates of the item: from offset (window & 0xFFFF) of its first unit up to (not including)
                         // offset (window >> 16) of its last unit; the rest are padding or a neighbouring group's
    uint32_t shape;      // candidates per lane of the item's units: 32 = a full unit of 2048 candidates, one guide per
                         // pass; 16 / 8 = the SHORT last unit of a successor-byte group (<= 1024 / <= 512 candidates):
                         // every register then holds the lane's 16 / 8 candidates two / four times over and a pass
                         // compares two / four guides at once, with masks the wave makes in LDS (short_unit_masks)
    uint32_t gmid;       // pruned scan: the group's class-0 guides (successor byte = the group's own) sit in the slots from
                         // here on, its class-1 guides (one mismatch there) in front of them (fine_word, issl_kernels.hip)
};
static_assert(sizeof(ScanItem) == 48, "scan items are fetched with scalar loads");

// Start of a cost range of the scan: (item, tile inside the item, guide offset); item == n_items marks the end.
struct RangeStart {
    uint32_t item;
    uint32_t tile;
    uint32_t goff; // guide offset inside the item (multiple of 8): a tile may be shared by two ranges
    uint32_t pad;
};

// Written by the planning kernel, read-only for the scan.
struct PlanInfo {
    uint32_t n_items;
    uint32_t error;       // bit 1: item list overflow
    uint32_t n_ranges;    // equal-cost ranges of the scan, one per workgroup
    uint32_t fine;        // 0: the scan works through the bucket-level items; 1 / 2: through the successor-byte groups
                          // (k_fine_plan decides, per batch, from the two plans' estimated times)
    uint64_t total_cost;
    uint64_t candidates;  // what the plan EXPECTS the scan to compare: sum over guides of the lengths of the buckets
                          // (full scan) or successor-byte groups (pruned scan) it visits
    uint64_t reference_candidates; // sum over guides of their five bucket lengths = what the reference compares
    uint64_t tiles;       // (tile, item) pairs the scan works through
    uint32_t fine_slots;  // pruned plan: guide slots in use (a multiple of 8)
    uint32_t pad;
};

// Updated by the scan with atomics.
struct Counters {
    uint32_t next_range;  // ticket counter of the scan's dynamic tail
    uint32_t n_big;       // guides with more than kReplayLds hits (the list k_replay_mid and k_replay_big share)
    uint32_t n_big2;      // guides k_replay_mid passes on to the 256-thread k_replay_big: more than kMidHits hits, or a slice too long for its buffers
    uint32_t n_big3;      // ... to the 1024-thread build (more than kBigSmall hits): listed from the far end of the same array
    uint32_t raw_chunks;  // chunks of the raw record buffer handed out
    uint32_t raw_overflow; // set when the raw buffer was too small
    uint32_t overflowed;  // hit slots: guides with more than kReplayLds hits (k_verify); 0 = nothing to group, no many-hit replay
    uint32_t replay_next[3]; // tickets of the many-hit replays (k_replay_mid, k_replay_big<256>, k_replay_big<1024>): the next entry
                             // of the shared guide list a workgroup of that kernel takes
};

// Per bucket, pruned scan: what its successor-byte groups add to the plan (k_fine_count -> k_fine_plan -> k_fine_scatter).
struct FineSum {
    uint64_t cost, cand;  // cost units and planned comparisons of the bucket's items
    uint32_t slots, items, units;
    uint32_t places;      // guides placed in the bucket's groups (every guide counts once per group it visits)
};

constexpr uint32_t kSlotHits = 512;     // = kReplayLds: what k_replay takes
constexpr uint32_t kSlotHitsWide = 2048; // = kMidHits: what a workspace grows its slots to when many guides of its batches have more
                                        // than kSlotHits hits (indexes of billions of sites, skewed genomes): the grouping pass then moves
                                        // only what lies beyond 2048 hits of a guide, and k_replay_mid's guides take no part in it
// One hit in a guide's slots: terms and key in ONE aligned 32-byte record -- k_verify scatters them, and a write that
// fills a whole 32-byte sector goes out as it is, where an 8- and a 16-byte piece of two arrays cost two partial ones.
struct alignas(32) SlotRec {
    double mit, cfd;
    uint64_t key, pad;
};
constexpr uint32_t kFineWays = 13;      // successor bytes within one mismatch of a guide's: itself + 4 positions x 3 bases
constexpr uint32_t kFineWays2 = 67;     // ... within two (max_dist 5): + 6 pairs of positions x 9 base pairs
constexpr uint32_t kFetchPairs = 8;     // time of fetching one 8 KiB tile, in (guide, tile) comparisons of the chip: measured 7
                                        // (50 M sites, neighbouring groups share tiles in L2) to 12 (300 M sites)
constexpr uint32_t kPruneMaxGuides = 1u << 20; // guides per pruned launch: 65 slots per guide + padding must fit the 27-bit slot field
constexpr uint32_t kPruneMaxGuides2 = 1u << 18; // ... 335 slots per guide (max_dist 5)
// (ten 4-bit / twenty 2-bit slices: two / four times the buckets per guide, a half / a quarter of the guides)
inline uint32_t prune_max_guides(uint32_t prune_mode, uint32_t n_slices)
{
    const uint32_t m = prune_mode == 3 ? kPruneMaxGuides2 : kPruneMaxGuides;
    return n_slices > 10 ? m >> 2 : n_slices > 5 ? m >> 1 : m;
}

// What k_verify needs to know about a guide slot of the pruned plan, in one 16-byte load.
struct alignas(16) FineMeta {
    uint32_t guide;  // index into the batch, kNoGuide in padding slots
    uint32_t where;  // bucket << 8 | successor byte of the group the slot belongs to
    uint64_t gsig;   // the guide's packed signature
};

struct Workspace {
    uint32_t *ng = nullptr;      // [nb]   guides per bucket
    uint32_t *gfill = nullptr;   // [nb]
    uint32_t *gstart = nullptr;  // [nb+1] padded start of every bucket in gword/gidx
    uint32_t *gword = nullptr;   // scan word of the guide for that slice
    uint32_t *gidx = nullptr;    // guide index, kNoGuide in padding
    uint32_t *gbucket = nullptr; // bucket of the slot (valid where gidx is a guide)
    ScanItem *items = nullptr;   // [max_items+1]
    // pruned scan: the guide arrays and the item list once more, grouped by (bucket, successor byte)
    uint32_t *fword = nullptr;   // scan words of the guides, grouped
    FineMeta *fmeta = nullptr;   // per slot: its guide (kNoGuide in padding), its group, the guide's signature
    ScanItem *fitems = nullptr;
    uint32_t *fcount = nullptr;  // [nb * 256] guides per (bucket, successor byte)
    uint32_t *fcount0 = nullptr; // [nb * 256] ... of them the ones whose own successor byte it is (class 0)
    FineSum *fsum = nullptr;     // [nb] per-bucket totals, then their exclusive prefix
    size_t cap_fslots = 0, cap_fitems = 0;
    uint32_t fine_ways = 0;      // placements per guide and bucket the pruned plan's arrays are sized for (kFineWays / kFineWays2)
    PlanInfo *plan = nullptr;
    RangeStart *range_start = nullptr; // [kMaxRanges + 1]
    uint64_t *raw = nullptr;     // [(cap_chunks+1) * kChunkRecs] raw records of the scan, chunked
    uint32_t *raw_used = nullptr; // [cap_chunks+1] slots in use per chunk (slot 0, which holds nothing, included)
    size_t cap_chunks = 0;
    Counters *counters = nullptr;
    unsigned long long *scan_span = nullptr; // [2 * kSpanRing] first start / last end of the scan workgroups of the lane's
                                    // recent batches, in ticks of the 100 MHz constant clock (s_memrealtime): the
                                    // kernel's own duration also when another stream shares the chip
    uint32_t span_slot = 0;         // slot of the batch being enqueued
    uint64_t *scan_count = nullptr; // [kScanMaxBlocks] comparisons every scan workgroup actually made (real candidates x real guides)
    unsigned long long *stamps = nullptr; // [kStampsWords] scan wave clocks, then replay phase clocks (ISSL_SCAN_STAMPS diagnostics)
    uint32_t *sticky = nullptr;  // [4] survives the per-batch resets: [0] bit 0 raw overflow seen, bit 1 a lean tail met a guide beyond its hit slots (run
                                 // the batch again), bit 2 a guide beyond its hit slots seen at all; [1] max chunks asked, [2] plan errors, [3] items a pruned plan wanted beyond cap_fitems
    uint64_t *sorted = nullptr;  // [hit_cap] keys guide<<37 | slice<<32 | site id or list position, grouped by guide (with hit slots: of the guides that outgrew them)
    uint32_t *gcount = nullptr;  // [G+1] hits per guide
    uint32_t *goff = nullptr;    // [G+1] exclusive prefix
    uint32_t *gcur_big = nullptr; // [G] guides with more than kReplayLds hits
    uint32_t *gcur_big2 = nullptr; // [G + 1] ... that k_replay_mid passes on to k_replay_big (Counters::n_big2 from the front, n_big3 from the back)
    double *terms = nullptr;     // [2 * hit_cap] MIT/CFD terms of the hits, grouped by guide like `sorted`
    double *pay = nullptr;       // [2 * hit_cap] the same terms as k_verify computed them, by raw-record slot
    uint32_t *rank = nullptr;    // [hit_cap] place of a surviving raw record inside its guide's segment, by raw-record slot
    // Hit slots: the first slot_hits (0 or kSlotHits) hits of every guide go straight from k_verify to a place of their own,
    // slots[guide * slot_hits + rank]; only guides with more hits than that take part in the grouping pass
    // (on an even index: none -- the pass and its buffers drop out of the step).  0: every hit is grouped (issl_dump_hits,
    // batches too large for the slot arrays, the hit_slots knob).
    SlotRec *slots = nullptr;       // [cap_slot_guides * slot_width]
    uint32_t slot_hits = 0;         // of the batch being enqueued: 0 or slot_width
    uint32_t slot_width = kSlotHits; // slots per guide the array was allocated with (kSlotHits / kSlotHitsWide)
    uint32_t lean_tail = 0;         // of the batch being enqueued: 1 = the grouping pass and the three many-hit replays are NOT
                                    // launched for it (five dependent launches that find nothing to do on an index where no
                                    // guide outgrows its hit slots -- 25 us of every batch, a fifth of a 64-guide one).  A
                                    // prediction from the lane's previous batches; k_replay sets sticky[0] bit 1 when a guide
                                    // of this batch did outgrow its slots, and the batch is run again with the whole tail
                                    // (the same ISSL_E_RETRY round an exhausted record buffer takes)
    size_t cap_slot_guides = 0;
    uint32_t *blocksum = nullptr;
    uint64_t *d_guides = nullptr; // staging for the host API
    void *h_stage = nullptr;      // ... and its pinned host side: guides in, MIT and CFD scores out (issl_score; 24 B per guide)
    size_t h_stage_bytes = 0;
    double *d_mit = nullptr, *d_cfd = nullptr;
    uint32_t *d_kept = nullptr;   // [G] hits scored before early exit (dump_hits)
    issl_hit *d_hitrec = nullptr; // [hit_cap] expanded records (dump_hits)
    size_t cap_hitrec = 0;
    size_t cap_guides = 0;       // capacity in guides
    size_t cap_hits = 0;
    size_t cap_items = 0;
    size_t cap_gslots = 0;
    uint32_t n_buckets = 0;
};

struct ScoreParams {
    int max_dist;
    int method;
    double maximum_sum; // (10000 - 100*thr)/thr, isslScoreOfftargets.cpp:326
};

// Launchers (issl_kernels.hip).  All asynchronous on `stream`.
// error_flag bits: 1 = an entry's id lies beyond the site table, 4 = a list holds a site in a bucket its signature does
// not select, or twice (`seen`: a zeroed bitmap of n_slices * n_sites bits, list-order layouts; null for the sorted
// ones, whose construction has checked both).
void launch_pack_scan_stream(const ImageView &v, uint32_t *scan_out, uint64_t *esig_out, uint8_t *occ8_out,
                             uint32_t *error_flag, uint32_t *seen, void *stream);
void launch_pack_scan_range(const ImageView &v, uint32_t *scan_out, uint64_t *esig_out, uint8_t *occ8_out,
                            uint32_t *error_flag, uint32_t *seen, uint32_t tile_begin, uint32_t tile_end, void *stream);
// prune_mode: 0 = full scan of the five buckets of every guide; 1 / 2 / 3 = pruned scan over the successor-byte groups equal
// to / within one / within two mismatches of the guide's own successor byte (enough for max_dist <= 2 / <= 4 / = 5, see
// k_fine_count).
inline uint32_t fine_ways_of(uint32_t prune_mode) { return prune_mode == 1 ? 1u : prune_mode == 3 ? kFineWays2 : kFineWays; }
uint32_t prune_mode_for(const ImageView &v, const Tuning &tn, uint32_t n_guides, int max_dist);
void launch_bin_guides(const ImageView &v, const Workspace &ws, const Tuning &tn, const uint64_t *d_guides, uint32_t n,
                       uint32_t prune_mode, void *stream);
void launch_scan(const ImageView &v, const Workspace &ws, const Tuning &tn, const uint64_t *d_guides, uint32_t n,
                 int max_dist, uint32_t prune_mode, void *stream);
void launch_verify(const ImageView &v, const Workspace &ws, const uint64_t *d_guides, uint32_t n, const ScoreParams &p, void *stream);
void launch_group_hits(const Workspace &ws, uint32_t n, void *stream);
void launch_replay(const ImageView &v, const Workspace &ws, const uint64_t *d_guides, uint32_t n,
                   const ScoreParams &p, double *d_mit, double *d_cfd, uint32_t *d_kept, issl_hit *d_hitrec,
                   void *stream);

} // namespace issl
