// HIP kernels of the ISSL off-target scorer for gfx950 (MI355X).  Wave = 64 lanes.
//
// Pipeline for one batch of guides (reference: src/ISSL/isslScoreOfftargets.cpp:307-511):
//   bin_*      group the guides of the batch by (slice, slice value) = by index bucket; on a sorted image
//              once more by (bucket, byte of the successor slice) for the pruned scan                 (A3)
//   scan       bucket tiles (pruned: the units of the successor-byte groups that can hold a hit) against
//              the guides placed there: XOR, fold, popcount, bit-sliced; candidates within max_dist are
//              NOTED as 8-byte records (guide slot, tile, offset)                                     (A4-A6)
//   verify     every record: exact test on the whole signatures, first-matching-slice rule (replaces
//              the seen-bitmap, A7), key (guide, slice, site id or list position), MIT / CFD terms;
//              the first 512 hits of a guide go straight to its hit slots (Workspace)                 (A7-A9)
//   group_*    counting sort by guide of the keys and terms that lie beyond their guide's slots
//              (nothing on an index where no guide has more than 512 hits)
//   replay     per guide: its hits in key order = the reference's scan order, terms added sequentially
//              with the reference's early exit                                                        (A10-A11)
//
// The scan is the ALU-critical kernel.  It streams 4 B per candidate (the signature with the bucket's own
// slice removed, stored bit-sliced: 32 candidates per 32-bit plane) and keeps a tile of 2048 candidates in
// registers while the guide words of the item arrive through scalar loads, so one HBM read of a tile serves
// every guide of the batch that is placed there.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>
#include <cstdio>

#include "issl_device.hpp"

namespace issl {

// CFD penalty tables (cfdPenalties.h:1-346) live in the code object's constant segment.
#define ISSL_CFD_QUAL __constant__
#include "cfd_tables.inc"

// ------------------------------------------------------------------------------------------------
// bit helpers
// ------------------------------------------------------------------------------------------------

// The short kernels around the scan raise their wave priority: when they share the GPU with a scan (two lanes),
// they are latency-bound and few, and should not queue behind the scan's older waves.
__device__ __forceinline__ void short_kernel_priority() { __builtin_amdgcn_s_setprio(3); }

// Even bits of a 32-bit word gathered into the low 16 bits.
__host__ __device__ inline uint32_t gather_even16(uint32_t x)
{
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu;
    x = (x | (x >> 8)) & 0x0000FFFFu;
    return x;
}

// Scan word of a 20 bp signature for slice `s` (8-bit slices): drop the slice's own byte (it is
// equal for every candidate and guide of the bucket), then split the remaining 16 positions into
// low-bit plane (bits 0..15) and high-bit plane (bits 16..31).  Two words differ at position p
// iff bit p of (x | x>>16) is set, x = a ^ b.
// Narrower slices (width 4 / 2: the reference scorer takes any width, :261-270,330-341) leave 18 / 19 other positions:
// the scan word keeps the first 16 of them, so the scan's count is a lower bound of the distance there -- it notes a
// superset of the hits and k_verify's exact test on the whole signatures decides, as it does anyway.
__host__ __device__ inline uint32_t scan_word(uint64_t sig, uint32_t s, uint32_t width = 8u)
{
    const uint32_t sh = width * s;
    const uint64_t low = sig & ((1ull << sh) - 1ull);
    const uint64_t high = (sig >> (sh + width)) << sh;
    const uint32_t rem = static_cast<uint32_t>(low | high);
    return gather_even16(rem) | (gather_even16(rem >> 1) << 16);
}

// Narrow slices on a SORTED layout (round 4; ten 4-bit or twenty 2-bit slices): the word's first quad holds the four
// positions of the successor unit (the next two / four slices: one byte, the same for every candidate of a successor-byte
// group, so the pruned scan leaves that quad in memory as it does with 8-bit slices), then the previous slice's two / one
// (fine_dup), then the ten / eleven positions that follow the successor unit; the last two / three before the previous slice
// are left out -- the count is a lower bound, k_verify decides.
__host__ __device__ inline uint32_t scan_word_sorted_narrow(uint64_t sig, uint32_t s, uint32_t width)
{
    const uint32_t sh = (width * (s + 1u)) % 40u;
    const uint64_t x = sig & kSigMask;
    const uint64_t a = (sh ? (x >> sh) | (x << (40u - sh)) : x) & kSigMask; // slices s + 1, s + 2, ... , s - 1, s from bit 0 on
    const uint64_t prev = (a >> (40u - 2u * width)) & ((1ull << width) - 1ull);
    const uint64_t mid = (a >> 8) & ((1ull << (24u - width)) - 1ull);
    const uint32_t rem = static_cast<uint32_t>((a & 0xFFull) | (prev << 8) | (mid << (8u + width)));
    return gather_even16(rem) | (gather_even16(rem >> 1) << 16);
}
// The word of a signature in slice s as the image's stream holds it (guides are packed the same way).
__host__ __device__ inline uint32_t image_word(uint64_t sig, uint32_t s, uint32_t width, bool sorted_layout)
{
    return (sorted_layout && width < 8u) ? scan_word_sorted_narrow(sig, s, width) : scan_word(sig, s, width);
}

// Mismatch flags of two packed signatures, one flag at bit 2p (isslScoreOfftargets.cpp:376-379).
__host__ __device__ inline uint64_t mismatch_mask(uint64_t a, uint64_t b)
{
    const uint64_t x = a ^ b;
    return ((x & 0xAAAAAAAAAAAAAAAAull) >> 1) | (x & 0x5555555555555555ull);
}

// Bit-sliced tile layout.  A tile holds 2048 candidates = 64 groups of 32.  Group G is owned by lane G of
// the scanning wave and consists of 32 PLANES: plane r < 16 holds, for its 32 candidates (bit j =
// candidate at tile offset 32 G + j), the low bit of the 2-bit code at position r of the scan word;
// plane 16 + r the high bit.  The word of (plane r, group G) sits at index ((r / 4) * 64 + G) * 4 + r % 4,
// so the scanning wave fetches its 32 planes with 8 coalesced 16-byte loads per lane.
constexpr int kPlanes = 32; // planes per lane = VGPRs holding the lane's 32 candidates

__host__ __device__ inline uint32_t plane_word(uint32_t r, uint32_t group)
{
    return ((r >> 2) * 64u + group) * 4u + (r & 3u);
}

// ------------------------------------------------------------------------------------------------
// upload: build the scan stream from sites + bucket entries
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_pack_scan_stream(ImageView v, uint32_t *__restrict__ scan_out,
                                                          uint64_t *__restrict__ esig_out,
                                                          uint8_t *__restrict__ occ8_out,
                                                          uint32_t *__restrict__ error_flag, uint32_t *__restrict__ seen,
                                                          uint32_t tile_begin, uint32_t tile_end)
{
    for (uint32_t t = tile_begin + blockIdx.x; t < tile_end; t += gridDim.x) {
        // bucket of tile t: last b with tile_first[b] <= t (uniform binary search)
        uint32_t lo = 0, hi = v.n_buckets;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (v.tile_first[mid] <= t) lo = mid; else hi = mid;
        }
        const uint32_t b = lo;
        const uint32_t slice = b >> v.slice_width;
        const uint64_t start = v.bucket_start[b];
        const uint64_t len = v.bucket_start[b + 1] - start;
        const uint64_t tile_pos = static_cast<uint64_t>(t - v.tile_first[b]) * kTileCands;
        // 64 consecutive candidates per wave and step: lane j computes the scan word of candidate j, then the
        // wave transposes the 64 x 32 bit matrix with ballots: plane r of the two 32-candidate groups is
        // the low / high half of ballot(bit r).  Lane r (< 32) keeps plane r and stores it.
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t wave = threadIdx.x >> 6;
        uint32_t *tile_out = scan_out + static_cast<uint64_t>(t) * kTileCands;
        for (uint32_t k0 = wave * 64u; k0 < kTileCands; k0 += 256u) {
            const uint64_t pos = tile_pos + k0 + lane;
            uint32_t w = 0;
            if (pos < len && (v.srec || v.sid)) {
                // sorted layouts: the stream holds the candidates v.srec / v.sid list (built and checked by launch_sort_slice)
                const uint64_t at = static_cast<uint64_t>(t) * kTileCands + k0 + lane; // the maps are indexed like the stream
                w = image_word(v.srec ? v.srec[at].sig & kSigMask : v.sites[v.sid[at]] & kSigMask, slice, v.slice_width, true);
            } else if (pos < len) {
                const uint64_t e = v.entries[start + pos];
                const uint64_t id = e & 0xFFFFFFFFull;
                if (id < v.n_sites) {
                    const uint64_t sig = v.sites[id] & kSigMask;
                    // Every slice must list every site once, in the bucket its signature selects: the scan compares the
                    // 16 positions outside the slice and the first-matching-slice rule stands in for the reference's
                    // seen-bitmap (:385-390) on exactly that premise.  `seen`: one bit per (slice, site).
                    if (((sig >> (v.slice_width * slice)) & ((1ull << v.slice_width) - 1ull)) != (b & ((1u << v.slice_width) - 1u)))
                        atomicOr(error_flag, 4u);
                    if (seen) {
                        const uint64_t bit = static_cast<uint64_t>(slice) * v.n_sites + id;
                        if (atomicOr(&seen[bit >> 5], 1u << (bit & 31u)) & (1u << (bit & 31u))) atomicOr(error_flag, 4u);
                    }
                    w = scan_word(sig, slice, v.slice_width);
                    if (esig_out) esig_out[start + pos] = sig;
                    if (occ8_out) occ8_out[start + pos] = static_cast<uint8_t>((e >> 32) < 255ull ? (e >> 32) : 255ull);
                } else {
                    atomicOr(error_flag, 1u);
                }
            }
            uint32_t mine_lo = 0, mine_hi = 0;
            for (uint32_t r = 0; r < 32; ++r) {
                const uint64_t m = __ballot((w >> r) & 1u);
                if (lane == r) { mine_lo = static_cast<uint32_t>(m); mine_hi = static_cast<uint32_t>(m >> 32); }
            }
            if (lane < 32) {
                const uint32_t group = k0 >> 5; // lane index (in the scan kernel) that owns candidates k0..k0+31
                tile_out[plane_word(lane, group)] = mine_lo;
                tile_out[plane_word(lane, group + 1u)] = mine_hi;
            }
        }
    }
}

// Tiles [tile_begin, tile_end) only: the upload of an image whose cold sections stay in host memory packs one slice
// at a time from temporary device copies (v.entries then points at the slice's list minus the slice's offset).
void launch_pack_scan_range(const ImageView &v, uint32_t *scan_out, uint64_t *esig_out, uint8_t *occ8_out,
                            uint32_t *error_flag, uint32_t *seen, uint32_t tile_begin, uint32_t tile_end, void *stream)
{
    if (tile_end <= tile_begin) return;
    const uint32_t n = tile_end - tile_begin;
    const uint32_t grid = n < 65536u ? n : 65536u;
    hipLaunchKernelGGL(k_pack_scan_stream, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), v,
                       scan_out, esig_out, occ8_out, error_flag, seen, tile_begin, tile_end);
}

void launch_pack_scan_stream(const ImageView &v, uint32_t *scan_out, uint64_t *esig_out, uint8_t *occ8_out,
                             uint32_t *error_flag, uint32_t *seen, void *stream)
{
    launch_pack_scan_range(v, scan_out, esig_out, occ8_out, error_flag, seen, 0u, v.n_tiles, stream);
}

// The packed signature of the candidate at offset `offset` of scan tile `tile` of bucket `bucket`, rebuilt from the
// scan stream: bit (offset % 32) of the 32 plane words of its group gives the 16 remaining positions (inverse of
// scan_word), the bucket number gives the slice's own byte.  8 loads of 16 B -- HBM, where the site table may be in
// host memory.
__device__ inline uint64_t candidate_signature(const ImageView &v, uint32_t bucket, uint32_t tile, uint32_t offset)
{
    const uint4 *src = reinterpret_cast<const uint4 *>(v.scan + static_cast<uint64_t>(tile) * kTileCands);
    const uint32_t group = offset >> 5, bit = offset & 31u;
    uint32_t w = 0; // plane r of the candidate at bit r: bits 0..15 = low code bits of the 16 positions, 16..31 = high bits
#pragma unroll
    for (uint32_t q = 0; q < kPlanes / 4; ++q) {
        const uint4 t4 = src[q * 64u + group];
        w |= ((t4.x >> bit) & 1u) << (4 * q) | ((t4.y >> bit) & 1u) << (4 * q + 1) | ((t4.z >> bit) & 1u) << (4 * q + 2) |
             ((t4.w >> bit) & 1u) << (4 * q + 3);
    }
    uint64_t rem = 0; // 16 positions x 2 bits: low bit of position p at bit 2p, high bit at 2p + 1
#pragma unroll
    for (uint32_t p = 0; p < 16; ++p)
        rem |= static_cast<uint64_t>(((w >> p) & 1u) | (((w >> (16 + p)) & 1u) << 1)) << (2 * p);
    const uint32_t slice = bucket >> v.slice_width;
    const uint32_t sh = v.slice_width * slice;
    const uint64_t key = bucket & ((1u << v.slice_width) - 1u);
    return (rem & ((1ull << sh) - 1ull)) | (key << sh) | ((rem >> sh) << (sh + v.slice_width));
}

// ------------------------------------------------------------------------------------------------
// guide binning
// ------------------------------------------------------------------------------------------------

constexpr uint32_t kMaxBuckets = 2048;

// Histogram of the guides' slice keys, and -- in the same launch -- the reset of everything a scoring call
// accumulates into (nothing here depends on it; the kernels that do come later on the stream).  ng and gfill are
// not reset here: k_plan leaves them zeroed for the next batch.
__global__ __launch_bounds__(256) void k_guide_hist(Workspace ws, const uint64_t *__restrict__ guides, uint32_t n,
                                                    uint32_t slice_width, uint32_t n_slices, uint32_t n_buckets,
                                                    uint32_t n_slots, uint32_t n_scan_waves)
{
    short_kernel_priority();
    __shared__ uint32_t hist[kMaxBuckets];
    for (uint32_t b = threadIdx.x; b < n_buckets; b += 256) hist[b] = 0;
    __syncthreads();
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g < n) {
        const uint64_t sig = guides[g];
        const uint32_t low = (1u << slice_width) - 1u;
        for (uint32_t s = 0; s < n_slices; ++s) {
            const uint32_t key = static_cast<uint32_t>(sig >> (slice_width * s)) & low;
            atomicAdd(&hist[(s << slice_width) + key], 1u);
        }
    }
    const uint32_t stride = gridDim.x * 256;
    for (uint32_t k = g; k < n_slots; k += stride) { ws.gidx[k] = kNoGuide; ws.gword[k] = kPadGuideWord; }
    for (uint32_t k = g; k <= n; k += stride) ws.gcount[k] = 0;
    // chunk fill counts: a chunk nobody writes must read as empty; chunks [0, n_scan_waves) belong to the scan waves.  (The
    // counts have an array of their own: cleared in one sweep, where a header word inside every 1 KiB chunk cost a cache
    // line per chunk -- 0.2 ms on a skewed index, whose raw buffer has grown -- and out of the way of k_replay_big, which
    // uses the raw buffer as scratch.)
    for (uint32_t k = g; k <= ws.cap_chunks; k += stride) ws.raw_used[k] = 0;
    if (g == 0) {
        Counters c{};
        c.raw_chunks = n_scan_waves;
        *ws.counters = c;
        ws.scan_span[2u * ws.span_slot] = ~0ull;
        ws.scan_span[2u * ws.span_slot + 1u] = 0ull;
    }
    __syncthreads();
    if (blockIdx.x * 256 < n)
        for (uint32_t b = threadIdx.x; b < n_buckets; b += 256)
            if (hist[b]) atomicAdd(&ws.ng[b], hist[b]);
}

// Exclusive scan of one uint64 per thread over a 256-thread block: shuffles inside the four waves, one LDS
// exchange across them (two barriers instead of the seventeen of a Hillis-Steele scan through LDS).
__device__ inline uint64_t wave_inclusive_scan_u64(uint64_t x)
{
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t lo = __shfl_up(static_cast<uint32_t>(x), d, 64);
        const uint32_t hi = __shfl_up(static_cast<uint32_t>(x >> 32), d, 64);
        if (lane >= d) x += (static_cast<uint64_t>(hi) << 32) | lo;
    }
    return x;
}

__device__ inline uint64_t block_exclusive_scan(uint64_t v, uint64_t *lds /*[256], 4 used*/, uint64_t *total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t incl = wave_inclusive_scan_u64(v);
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    uint64_t before = 0, all = 0;
    for (uint32_t w = 0; w < 4; ++w) {
        const uint64_t s = lds[w];
        if (w < wave) before += s;
        all += s;
    }
    if (total) *total = all;
    __syncthreads();
    return before + incl - v;
}

// Start of cost range r of n_ranges (r == n_ranges: the end marker).
template <bool COOP = false>
__device__ __forceinline__ RangeStart range_start_of(const ScanItem *__restrict__ items, uint32_t n_items, uint64_t total,
                                                     uint32_t n_ranges, uint32_t r)
{
    RangeStart out;
    out.item = n_items;
    out.tile = 0;
    out.goff = 0;
    out.pad = 0;
    if (r < n_ranges) {
        // no 128-bit intermediate: costs < 2^50 and range counts <= 2^15
        const uint64_t lo = total / n_ranges * r + (total % n_ranges) * r / n_ranges;
        uint32_t a = 0, z = n_items; // last item with cost0 <= lo
        if (COOP) {
            // the whole wave looks for ONE range's start: 64 probes per round trip instead of one (a list of a million items: four
            // dependent loads instead of twenty -- the chain was most of k_fine_ranges' 7 - 12 us)
            const uint32_t lane = threadIdx.x & 63u;
            while (z - a > 1) {
                const uint32_t step = (z - a + 63u) / 64u;
                const uint64_t pos = static_cast<uint64_t>(a) + static_cast<uint64_t>(lane + 1u) * step;
                const bool le = pos < z && items[pos].cost0 <= lo;
                const uint32_t k = static_cast<uint32_t>(__popcll(__ballot(le))); // (costs ascend: the lanes that say yes are the first k)
                const uint64_t na = static_cast<uint64_t>(a) + static_cast<uint64_t>(k) * step, nz = na + step;
                a = static_cast<uint32_t>(na);
                if (nz < z) z = static_cast<uint32_t>(nz);
            }
        } else
        while (z - a > 1) {
            const uint32_t mid = (a + z) >> 1;
            if (items[mid].cost0 <= lo) a = mid; else z = mid;
        }
        // A unit of an item costs kTileFixedCost (fetching it) + shape / 8 per guide (kGuideCost for a full unit).  A
        // boundary may fall between two groups of 8 guides INSIDE a unit: then two waves share that unit (both fetch
        // it), which makes the ranges equal to within 8 guides instead of within one unit.
        const ScanItem it = items[a];
        const uint32_t len = it.g1 - it.g0;
        const uint32_t per_guide = it.shape >> 3;
        const uint64_t tile_cost = static_cast<uint64_t>(len) * per_guide + kTileFixedCost;
        const uint64_t rel = lo - it.cost0;
        uint64_t k = rel / tile_cost;
        const uint64_t rem = rel % tile_cost;
        uint32_t goff = 0;
        if (rem > kTileFixedCost) { // (inside the fetch part the unit starts the range: positions stay monotone in r)
            goff = ((static_cast<uint32_t>(rem - kTileFixedCost) + per_guide - 1u) / per_guide + kGuideGroup - 1u) & ~(kGuideGroup - 1u);
            if (goff >= len) { goff = 0; ++k; }
        }
        if (k >= it.n_tiles) { out.item = a + 1; out.tile = 0; out.goff = 0; }
        else { out.item = a; out.tile = static_cast<uint32_t>(k); out.goff = goff; }
    }
    return out;
}

// Scan workgroups that get a range of the plan: all of the launch's for a plan of any size, fewer for a small one -- a workgroup
// with less than a few units per wave is no faster, and every scan wave owns a record chunk that k_verify then has to visit
// (a 64-guide batch of 12 k units: 1024 workgroups = 16 384 nearly empty chunks were 25 us of its 130).  The record chunks
// [0, 16 x ranges) are the scan waves' own; what is handed out later comes behind them (Counters::raw_chunks).
__device__ __forceinline__ uint32_t ranges_for(uint64_t units, uint32_t scan_blocks)
{
    const uint64_t want = units / 64u + 1u; // ~4 units per wave
    const uint32_t floor_ = scan_blocks < 64u ? scan_blocks : 64u;
    return want >= scan_blocks ? scan_blocks : want < floor_ ? floor_ : static_cast<uint32_t>(want);
}

// One block: lay out the bucket-sorted guide arrays and the list of scan items.
__global__ __launch_bounds__(256) void k_plan(ImageView v, uint32_t *__restrict__ ng, uint32_t *__restrict__ gfill,
                                              uint32_t *__restrict__ gstart, ScanItem *__restrict__ items,
                                              uint32_t cap_items, PlanInfo *__restrict__ plan, uint32_t item_guides,
                                              uint32_t scan_blocks, Counters *__restrict__ counters)
{
    short_kernel_priority();
    __shared__ uint64_t lds[256];
    const uint32_t nb = v.n_buckets;
    const uint32_t per = (nb + 255u) / 256u;
    const uint32_t b0 = threadIdx.x * per;
    const uint32_t b1 = (b0 + per < nb) ? b0 + per : nb;

    uint64_t slots = 0, n_it = 0, cost = 0, cand = 0, wtiles = 0;
    for (uint32_t b = b0; b < b1; ++b) {
        const uint32_t g = ng[b];
        const uint32_t nt = v.tile_first[b + 1] - v.tile_first[b];
        slots += (g + kGuideGroup - 1u) / kGuideGroup * kGuideGroup;
        if (g && nt) {
            const uint32_t k = (g + item_guides - 1u) / item_guides;
            n_it += k;
            cost += static_cast<uint64_t>(nt) * (static_cast<uint64_t>(g) * kGuideCost + static_cast<uint64_t>(k) * kTileFixedCost);
            cand += (v.bucket_start[b + 1] - v.bucket_start[b]) * g;
            wtiles += static_cast<uint64_t>(nt) * k;
        }
    }
    uint64_t tot_slots, tot_items, tot_cost, tot_cand, tot_tiles;
    uint64_t slot_at = block_exclusive_scan(slots, lds, &tot_slots);
    uint64_t item_at = block_exclusive_scan(n_it, lds, &tot_items);
    uint64_t cost_at = block_exclusive_scan(cost, lds, &tot_cost);
    (void)block_exclusive_scan(cand, lds, &tot_cand);
    uint64_t tile_at = block_exclusive_scan(wtiles, lds, &tot_tiles);

    const bool overflow = tot_items > cap_items;
    for (uint32_t b = b0; b < b1; ++b) {
        const uint32_t g = ng[b];
        const uint32_t nt = v.tile_first[b + 1] - v.tile_first[b];
        gstart[b] = static_cast<uint32_t>(slot_at);
        if (g && nt && !overflow) {
            const uint64_t blen = v.bucket_start[b + 1] - v.bucket_start[b];
            for (uint32_t done = 0; done < g; done += item_guides) {
                const uint32_t len = (g - done < item_guides) ? g - done : item_guides;
                ScanItem it;
                it.bucket = b;
                it.g0 = static_cast<uint32_t>(slot_at) + done;
                it.g1 = it.g0 + len;
                it.n_tiles = nt;
                it.cost0 = cost_at;
                it.tile0 = static_cast<uint32_t>(tile_at);
                it.last_cands = static_cast<uint32_t>(blen - static_cast<uint64_t>(nt - 1u) * kTileCands);
                it.group_abs = v.tile_first[b] * 64u;
                it.window = it.last_cands << 16;
                it.shape = 32; it.gmid = 0;
                items[item_at++] = it;
                cost_at += static_cast<uint64_t>(nt) * (len * kGuideCost + kTileFixedCost);
                tile_at += nt;
            }
        }
        slot_at += (g + kGuideGroup - 1u) / kGuideGroup * kGuideGroup;
        ng[b] = 0;    // for the next batch's histogram
        gfill[b] = 0; // for this batch's scatter
    }
    if (threadIdx.x == 255) {
        gstart[nb] = static_cast<uint32_t>(tot_slots);
        if (!overflow) {
            ScanItem end;
            end.bucket = 0; end.g0 = 0; end.g1 = 0; end.n_tiles = 0; end.cost0 = tot_cost;
            end.tile0 = static_cast<uint32_t>(tot_tiles); end.last_cands = 0; end.group_abs = 0; end.window = 0;
            end.shape = 32; end.gmid = 0;
            items[tot_items] = end;
        }
        plan->n_items = overflow ? 0u : static_cast<uint32_t>(tot_items);
        plan->total_cost = overflow ? 0ull : tot_cost;
        plan->candidates = tot_cand;
        plan->reference_candidates = tot_cand;
        plan->fine = 0;
        plan->tiles = tot_tiles;
        // one equal-cost range per scan workgroup; inside a workgroup the waves share the tiles dynamically
        plan->n_ranges = (overflow || tot_tiles == 0) ? 0u : ranges_for(tot_tiles, scan_blocks);
        counters->raw_chunks = (plan->n_ranges ? plan->n_ranges : 1u) * 16u; // (k_guide_hist has reset the counters; k_fine_plan may choose again)
        plan->error = overflow ? 2u : 0u;
    }
}

// Scatter every guide into its bucket's range of (gword, gidx), once per slice.
__global__ __launch_bounds__(256) void k_guide_scatter(const uint64_t *__restrict__ guides, uint32_t n,
                                                       uint32_t slice_width, uint32_t n_slices, uint32_t sorted_layout,
                                                       uint32_t n_buckets, const uint32_t *__restrict__ gstart,
                                                       uint32_t *__restrict__ gfill, uint32_t *__restrict__ gword,
                                                       uint32_t *__restrict__ gidx, uint32_t *__restrict__ gbucket,
                                                       uint32_t guide_blocks, const PlanInfo *__restrict__ plan,
                                                       const ScanItem *__restrict__ items,
                                                       RangeStart *__restrict__ starts)
{
    short_kernel_priority();
    if (blockIdx.x >= guide_blocks) {
        // the workgroups behind the guides resolve the cost ranges of the scan (independent of the scatter; one launch
        // less).  First tile of every range: range r owns the tiles whose start cost lies in [lo(r), lo(r+1)); done
        // once here so that the scan waves neither divide nor search.
        const uint32_t n_ranges = plan->n_ranges;
        const uint32_t r = (blockIdx.x - guide_blocks) * 256 + threadIdx.x;
        if (r <= n_ranges && n_ranges != 0) starts[r] = range_start_of(items, plan->n_items, plan->total_cost, n_ranges, r);
        return;
    }
    __shared__ uint32_t hist[kMaxBuckets];
    __shared__ uint32_t base[kMaxBuckets];
    for (uint32_t b = threadIdx.x; b < n_buckets; b += 256) hist[b] = 0;
    __syncthreads();
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    const uint32_t low = (1u << slice_width) - 1u;
    uint64_t sig = 0;
    uint32_t rank[kMaxSlices];
    if (g < n) {
        sig = guides[g];
#pragma unroll
        for (uint32_t s = 0; s < kMaxSlices; ++s) {
            if (s < n_slices) {
                const uint32_t key = static_cast<uint32_t>(sig >> (slice_width * s)) & low;
                rank[s] = atomicAdd(&hist[(s << slice_width) + key], 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < n_buckets; b += 256)
        base[b] = hist[b] ? atomicAdd(&gfill[b], hist[b]) : 0u;
    __syncthreads();
    if (g < n) {
#pragma unroll
        for (uint32_t s = 0; s < kMaxSlices; ++s) {
            if (s < n_slices) {
                const uint32_t key = static_cast<uint32_t>(sig >> (slice_width * s)) & low;
                const uint32_t b = (s << slice_width) + key;
                const uint32_t slot = gstart[b] + base[b] + rank[s];
                gword[slot] = image_word(sig, s, slice_width, sorted_layout != 0u);
                gidx[slot] = g;
                gbucket[slot] = b;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// pruned scan: guides grouped by (bucket, successor byte)
// ------------------------------------------------------------------------------------------------
// A site within max_dist <= 4 mismatches of a guide matches it exactly in some of the five slices (set E), and for at
// least one slice i of E the NEXT slice (i + 1 mod 5) has at most one mismatch: otherwise every slice of E is followed by
// a slice with >= 2 mismatches, these followers are distinct and lie outside E, the other slices outside E have >= 1
// each, and 2|E| + (5 - 2|E|) = 5 > 4 mismatches.  (For max_dist <= 2 the next slice is even exact for some i of E;
// both facts are checked by enumeration in tests/test_oracle_golden.py.)  With every bucket's candidates ordered by
// the byte of the successor slice, a guide therefore needs, in each of its five buckets, only the 13 groups whose
// successor byte is within one mismatch of its own (1 group for max_dist <= 2) instead of all 256 -- the reference
// scans the whole bucket, isslScoreOfftargets.cpp:344, and finds the same sites.  k_verify re-attributes a hit to the
// first exactly matching slice and its position there (pos_of), which is what the reference's order is made of.

// Units of one successor-byte group: its candidates [s0, s1) of the bucket's stream are covered from the group's first
// lane group (32 candidates) on by n_full full units of 2048 candidates and, for the `rest` behind them, one last unit of
// the smallest shape that holds it: 8, 16 or 32 candidates per lane (512 / 1024 / 2048 per unit).  In a short unit every
// plane register holds the lane's candidates 4 / 2 times over and one pass of the distance test serves 4 / 2 guides:
// the same comparisons per instruction as a full unit, a quarter / half of the idle lanes.
struct GroupUnits {
    uint32_t s0a, n_full, rest, shape, units;
};
__device__ __forceinline__ GroupUnits group_units(uint32_t s0, uint32_t s1, uint32_t tail_shapes)
{
    GroupUnits u;
    u.s0a = s0 & ~31u;
    const uint32_t span = s1 - u.s0a;
    u.n_full = span / kTileCands;
    u.rest = span - u.n_full * kTileCands;
    u.shape = (!tail_shapes || u.rest > 1024u) ? 32u : u.rest > 512u ? 16u : 8u;
    u.units = u.n_full + (u.rest ? 1u : 0u);
    return u;
}
// cost of the group's units against `len` guides (one chunk of at most item_guides of them)
__device__ __forceinline__ uint64_t group_cost(const GroupUnits &u, uint32_t len)
{
    return static_cast<uint64_t>(u.n_full) * (static_cast<uint64_t>(len) * kGuideCost + kTileFixedCost) +
           (u.rest ? static_cast<uint64_t>(len) * (u.shape >> 3) + kTileFixedCost : 0ull);
}

// The successor bytes a guide with successor byte `gj` visits: way 0 = gj itself, ways 1..12 = one position changed.
__device__ __forceinline__ uint32_t fine_way(uint32_t gj, uint32_t way)
{
    if (way == 0) return gj;
    if (way < kFineWays) {
        const uint32_t q = (way - 1u) / 3u, d = (way - 1u) % 3u + 1u;
        return gj ^ (d << (2u * q));
    }
    // ways 13..66 (max_dist 5): two of the four positions changed -- pair (a, b) of 6, bases (d1, d2) of 9
    const uint32_t w2 = way - kFineWays, pair = w2 / 9u, d1 = (w2 % 9u) / 3u + 1u, d2 = w2 % 3u + 1u;
    const uint32_t a = pair < 3u ? 0u : pair < 5u ? 1u : 2u;
    const uint32_t b = pair < 3u ? pair + 1u : pair < 5u ? pair - 1u : 3u;
    return gj ^ (d1 << (2u * a)) ^ (d2 << (2u * b));
}
// mismatches in the successor slice of a guide placed by `way`: 0, 1 or 2 -- the guide's class (fine_word)
__device__ __forceinline__ uint32_t fine_class(uint32_t way) { return way == 0u ? 0u : way < kFineWays ? 1u : 2u; }

// The pruned scan compares 12 positions, not 16.  An item of its plan is ONE (bucket, successor byte) group: inside the
// item's window every candidate carries the same four bases in the successor slice, and how far a guide is from them is
// known when the guide is placed -- 0 mismatches in its own group (way 0, "class 0"), 1 in the twelve others (ways 1..12,
// "class 1").  The scan therefore leaves the successor slice's planes in memory (two of the eight 16-byte loads per lane,
// scan_word: positions 4 s' .. 4 s' + 3 of the 16, s' = fine_quad(slice)) and counts the other 12 positions against
// max_dist - class; the reference's test :376-382 on the full signatures is k_verify's.  Guide word of the pruned plan:
// bits 0..11 the low code bits of the 12 positions (fine_order), bits 12..23 the high ones, bits 24..25 the class.
// A group's slots hold its class-1 guides first (from a multiple of 8 on), its class-0 guides behind them from
// ScanItem::gmid on: full units run the two classes as two loops with their own compiled tests, short units take the
// class bit as a thirteenth plane.
__host__ __device__ __forceinline__ uint32_t fine_quad(uint32_t slice) { return slice < 4u ? slice : 0u; }
// The order of the 12 positions (three quads of the scan word's four): the quad of the PREVIOUS slice (slice - 1) first --
// its four mismatch planes tell, for nothing, whether the slice before the bucket's own matches the guide exactly too, and
// a candidate for which it does is reported from that slice's bucket already (fine_dup) -- then the other two, ascending.
// Slice 0 has no previous slice: its quads in ascending order.  (Quad q of slice s's scan word holds slice q < s ? q : q + 1.)
// Narrow slices (scan_word_sorted_narrow): the successor unit is quad 0 for every slice, the previous slice opens quad 1: quads 1, 2, 3.
__host__ __device__ __forceinline__ uint32_t fine_order(uint32_t slice, uint32_t j, uint32_t slice_width = 8u)
{
    if (slice_width != 8u) return j + 1u;
    const uint32_t sq = fine_quad(slice);
    if (slice == 0u) return j + 1u;                                // quads 1, 2, 3
    const uint32_t prev = slice - 1u;                              // slice - 1 sits in quad slice - 1 (it is below the own slice)
    if (j == 0u) return prev;
    uint32_t q = 0, seen = 0;                                      // the j-th of the quads that are neither sq nor prev
    for (; q < 4u; ++q) {
        if (q == sq || q == prev) continue;
        if (++seen == j) break;
    }
    return q;
}
__host__ __device__ __forceinline__ uint32_t fine_word(uint32_t word, uint32_t slice, uint32_t slice_width = 8u)
{
    const uint32_t lo = word & 0xFFFFu, hi = word >> 16;
    uint32_t lo12 = 0, hi12 = 0;
    for (uint32_t j = 0; j < 3u; ++j) {
        const uint32_t q = fine_order(slice, j, slice_width);
        lo12 |= ((lo >> (4u * q)) & 0xFu) << (4u * j);
        hi12 |= ((hi >> (4u * q)) & 0xFu) << (4u * j);
    }
    return lo12 | (hi12 << 12);
}

// Per bucket: guides per successor byte, and what the bucket's groups add to the plan.
template <uint32_t WAYS>
__global__ __launch_bounds__(256) void k_fine_count(ImageView v, const uint64_t *__restrict__ guides,
                                                    const uint32_t *__restrict__ gstart, const uint32_t *__restrict__ gfill,
                                                    const uint32_t *__restrict__ gidx, uint32_t *__restrict__ fcount,
                                                    uint32_t *__restrict__ fcount0,
                                                    FineSum *__restrict__ fsum, uint32_t item_guides, uint32_t tail_shapes)
{
    constexpr uint32_t ways = WAYS; // compiled in: the 13 (or 67) LDS atomics of a guide are in flight together
    short_kernel_priority();
    __shared__ uint32_t cnt[256], cnt0[256];
    __shared__ uint64_t lds[256];
    const uint32_t b = blockIdx.x, slice = b >> v.slice_width;
    cnt[threadIdx.x] = 0;
    cnt0[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t g0 = gstart[b], n = gfill[b];
    for (uint32_t i = threadIdx.x; i < n; i += 256) { // one guide per thread and step: its loads once, its ways from registers
        const uint32_t gj = succ_byte(guides[gidx[g0 + i]], slice, v.slice_width);
        atomicAdd(&cnt0[gj], 1u); // class 0: the guide's successor byte is the group's own (fine_class)
#pragma unroll
        for (uint32_t way = 0; way < ways; ++way) atomicAdd(&cnt[fine_way(gj, way)], 1u);
    }
    __syncthreads();
    const uint32_t w = threadIdx.x, c = cnt[w];
    const uint32_t *ss = v.sub_start + static_cast<uint64_t>(b) * 257u;
    const uint32_t s0 = ss[w], s1 = ss[w + 1];
    fcount[static_cast<uint64_t>(b) * 256u + w] = (s1 > s0) ? c : 0u; // a group without candidates takes no guides
    fcount0[static_cast<uint64_t>(b) * 256u + w] = (s1 > s0) ? cnt0[w] : 0u;
    uint64_t cost = 0, cand = 0, slots = 0, items = 0, units = 0;
    if (c && s1 > s0) {
        const GroupUnits gu = group_units(s0, s1, tail_shapes);
        const uint32_t kk = (c + item_guides - 1u) / item_guides;
        slots = (c + kGuideGroup - 1u) / kGuideGroup * kGuideGroup;
        units = static_cast<uint64_t>(gu.units) * kk;
        items = units; // one item per unit and chunk of guides: the scan finds the item of a unit without a search
        for (uint32_t done = 0; done < c; done += item_guides) cost += group_cost(gu, c - done < item_guides ? c - done : item_guides);
        cand = static_cast<uint64_t>(s1 - s0) * c;
    }
    uint64_t t_cost, t_cand, t_slots, t_items, t_units, t_places;
    (void)block_exclusive_scan(cost, lds, &t_cost);
    (void)block_exclusive_scan(cand, lds, &t_cand);
    (void)block_exclusive_scan(slots, lds, &t_slots);
    (void)block_exclusive_scan(items, lds, &t_items);
    (void)block_exclusive_scan(units, lds, &t_units);
    (void)block_exclusive_scan((c && s1 > s0) ? c : 0u, lds, &t_places);
    if (threadIdx.x == 0) {
        FineSum f;
        f.cost = t_cost; f.cand = t_cand; f.slots = static_cast<uint32_t>(t_slots); f.items = static_cast<uint32_t>(t_items);
        f.units = static_cast<uint32_t>(t_units); f.places = static_cast<uint32_t>(t_places);
        fsum[b] = f;
    }
}

// One block: exclusive prefix of the per-bucket totals (in place) and the plan of the pruned scan.
__global__ __launch_bounds__(256) void k_fine_plan(FineSum *__restrict__ fsum, uint32_t nb, ScanItem *__restrict__ fitems,
                                                   uint32_t cap_items, uint32_t cap_slots, PlanInfo *__restrict__ plan,
                                                   uint32_t scan_blocks, uint32_t prune_mode, uint32_t always,
                                                   uint32_t *__restrict__ sticky, Counters *__restrict__ counters)
{
    short_kernel_priority();
    __shared__ uint64_t lds[256];
    const uint32_t per = (nb + 255u) / 256u;
    const uint32_t b0 = threadIdx.x * per, b1 = (b0 + per < nb) ? b0 + per : nb;
    uint64_t cost = 0, cand = 0, slots = 0, items = 0, units = 0, places = 0;
    for (uint32_t b = b0; b < b1; ++b) {
        cost += fsum[b].cost; cand += fsum[b].cand; slots += fsum[b].slots; items += fsum[b].items; units += fsum[b].units;
        places += fsum[b].places;
    }
    uint64_t t_cost, t_cand, t_slots, t_items, t_units, t_places;
    (void)block_exclusive_scan(places, lds, &t_places);
    uint64_t cost_at = block_exclusive_scan(cost, lds, &t_cost);
    (void)block_exclusive_scan(cand, lds, &t_cand);
    uint64_t slot_at = block_exclusive_scan(slots, lds, &t_slots);
    uint64_t item_at = block_exclusive_scan(items, lds, &t_items);
    uint64_t unit_at = block_exclusive_scan(units, lds, &t_units);
    for (uint32_t b = b0; b < b1; ++b) {
        const FineSum f = fsum[b];
        FineSum at;
        at.cost = cost_at; at.cand = 0; at.slots = static_cast<uint32_t>(slot_at); at.items = static_cast<uint32_t>(item_at);
        at.units = static_cast<uint32_t>(unit_at); at.places = 0;
        fsum[b] = at;
        cost_at += f.cost; slot_at += f.slots; item_at += f.items; unit_at += f.units;
    }
    if (threadIdx.x == 255) {
        // Which plan is faster?  Comparing and fetching overlap: time ~ max((guide, tile) pairs, kFetchPairs x tile
        // fetches).  Few guides per successor-byte group make the pruned scan fetch-bound (every group reads its own
        // tiles, a bucket-level item reads a tile once for up to 512 guides); costs are pairs + kTileFixedCost x fetches.
        // (all in cost units: kGuideCost per pair of a guide with a full unit)
        const uint64_t fetch_cost = static_cast<uint64_t>(kFetchPairs) * kGuideCost;
        const uint64_t full_fetch = plan->tiles, full_pairs = plan->total_cost - kTileFixedCost * full_fetch;
        const uint64_t fine_pairs = t_cost - kTileFixedCost * t_units;
        const uint64_t est_full = full_pairs > fetch_cost * full_fetch ? full_pairs : fetch_cost * full_fetch;
        // (+ one comparison per place of a guide in a group: binning every guide 65 times is not free either)
        const uint64_t est_fine = (fine_pairs > fetch_cost * t_units ? fine_pairs : fetch_cost * t_units) + t_places * kGuideCost;
        const bool fits = t_items <= cap_items && t_slots <= cap_slots; // (the slots cover every guide in 13 groups)
        if (t_items > cap_items) // the host enlarges the item list for the next batches; this one scans whole buckets
            atomicMax(&sticky[3], static_cast<uint32_t>(t_items < 0xFFFFFFFFull ? t_items : 0xFFFFFFFFull));
        if (fits && plan->error == 0 && (always || est_fine < est_full)) {
            ScanItem end;
            end.bucket = 0; end.g0 = 0; end.g1 = 0; end.n_tiles = 0; end.cost0 = t_cost;
            end.tile0 = static_cast<uint32_t>(t_units); end.last_cands = 0; end.group_abs = 0; end.window = 0;
            end.shape = 32; end.gmid = 0;
            fitems[t_items] = end;
            plan->n_items = static_cast<uint32_t>(t_items);
            plan->total_cost = t_cost;
            plan->candidates = t_cand; // reference_candidates stays what the bucket-level plan counted
            plan->tiles = t_units;
            plan->n_ranges = t_units == 0 ? 0u : ranges_for(t_units, scan_blocks);
            counters->raw_chunks = (plan->n_ranges ? plan->n_ranges : 1u) * 16u;
            plan->fine = prune_mode;
            plan->fine_slots = static_cast<uint32_t>(t_slots);
        }
    }
}

// Per bucket: the items of its successor-byte groups and the guides of every group in its slots.
template <uint32_t WAYS>
__global__ __launch_bounds__(256) void k_fine_scatter(ImageView v, const uint64_t *__restrict__ guides,
                                                      const uint32_t *__restrict__ gstart, const uint32_t *__restrict__ gfill,
                                                      const uint32_t *__restrict__ gword, const uint32_t *__restrict__ gidx,
                                                      const uint32_t *__restrict__ fcount, const uint32_t *__restrict__ fcount0,
                                                      const FineSum *__restrict__ fbase,
                                                      const PlanInfo *__restrict__ plan, uint32_t *__restrict__ fword,
                                                      FineMeta *__restrict__ fmeta,
                                                      ScanItem *__restrict__ fitems, uint32_t item_guides, uint32_t tail_shapes)
{
    constexpr uint32_t ways = WAYS;
    short_kernel_priority();
    if (!plan->fine) return; // the bucket-level plan stays
    __shared__ uint64_t lds[256];
    __shared__ uint32_t slot_of[256], slot0_of[256], cursor[256], cursor0[256], has_cands[256];
    const uint32_t b = blockIdx.x, slice = b >> v.slice_width;
    const uint32_t w = threadIdx.x;
    const uint32_t c = fcount[static_cast<uint64_t>(b) * 256u + w];
    const uint32_t c1 = c - fcount0[static_cast<uint64_t>(b) * 256u + w]; // class 1 first, class 0 behind it (fine_class)
    const uint32_t *ss = v.sub_start + static_cast<uint64_t>(b) * 257u;
    const uint32_t s0 = ss[w], s1 = ss[w + 1];
    uint64_t cost = 0, slots = 0, items = 0;
    uint32_t kk = 0;
    GroupUnits gu{};
    if (c) { // (fcount is zero where the group has no candidates)
        gu = group_units(s0, s1, tail_shapes);
        kk = (c + item_guides - 1u) / item_guides;
        slots = (c + kGuideGroup - 1u) / kGuideGroup * kGuideGroup;
        items = static_cast<uint64_t>(gu.units) * kk;
        for (uint32_t done = 0; done < c; done += item_guides) cost += group_cost(gu, c - done < item_guides ? c - done : item_guides);
    }
    const FineSum base = fbase[b];
    uint64_t cost_at = base.cost + block_exclusive_scan(cost, lds, nullptr);
    const uint32_t slot_at = base.slots + static_cast<uint32_t>(block_exclusive_scan(slots, lds, nullptr));
    uint32_t item_at = base.items + static_cast<uint32_t>(block_exclusive_scan(items, lds, nullptr));
    slot_of[w] = slot_at;
    slot0_of[w] = slot_at + c1;
    cursor[w] = 0;
    cursor0[w] = 0;
    // The bucket's items -- one per unit and chunk of guides, ~770 of 48 bytes -- are written by the whole workgroup, item i by
    // thread i % 256 into a staging row in LDS and from there in 16-byte pieces that consecutive lanes put side by side: every
    // group's thread writing its own three items one after the other touched each 64-byte line of the list three times from
    // different lanes (four times the requests of the bytes moved; the kernel is the largest part of the binning).
    __shared__ uint32_t item0_of[257], c_of[256];
    __shared__ uint64_t cost0_of[256];
    __shared__ __attribute__((aligned(16))) ScanItem stage[256];
    item0_of[w] = item_at - base.items;
    if (w == 255u) item0_of[256] = item_at - base.items + static_cast<uint32_t>(items);
    c_of[w] = c;
    cost0_of[w] = cost_at;
    if (c) // padding slots behind the group's guides
        for (uint32_t k2 = c; k2 < static_cast<uint32_t>(slots); ++k2) { fmeta[slot_at + k2] = FineMeta{kNoGuide, 0u, 0ull}; fword[slot_at + k2] = kPadGuideWord; }
    has_cands[w] = s1 > s0 ? 1u : 0u;
    __syncthreads();
    {
        const uint32_t total = item0_of[256];
        const uint64_t blen = v.bucket_start[b + 1] - v.bucket_start[b];
        const uint32_t tile_first_b = v.tile_first[b];
        for (uint32_t i0 = 0; i0 < total; i0 += 256u) {
            const uint32_t i = i0 + w;
            if (i < total) {
                uint32_t lo = 0, hi = 256; // the group of item i: the last one whose first item is <= i (groups without items share a start)
                while (hi - lo > 1u) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (item0_of[mid] <= i) lo = mid; else hi = mid;
                }
                const uint32_t gw = lo, gc = c_of[gw];
                const uint32_t gs0 = ss[gw], gs1 = ss[gw + 1];
                const GroupUnits g2 = group_units(gs0, gs1, tail_shapes);
                const uint32_t j = i - item0_of[gw];
                const uint32_t chunk = j / g2.units, t = j - chunk * g2.units; // single-unit items: chunk after chunk, unit after unit
                const uint32_t done = chunk * item_guides;
                const uint32_t len = (gc - done < item_guides) ? gc - done : item_guides;
                const bool full = t < g2.n_full;
                const uint32_t shape = full ? 32u : g2.shape, cap = 64u * shape;  // candidates the unit covers
                const uint32_t wstart = g2.s0a + t * kTileCands;               // position in the bucket (a lane group)
                const uint64_t after = blen - wstart;                           // candidates of the bucket from there on
                const uint32_t gslot = slot_of[gw];
                ScanItem it;
                it.bucket = (b << 8) | gw;
                it.g0 = gslot + done; // item_guides is a multiple of 8
                it.g1 = it.g0 + len;
                it.n_tiles = 1;
                // the chunks in front of this one are full ones; the units in front of this one inside its chunk are full units
                it.cost0 = cost0_of[gw] + static_cast<uint64_t>(chunk) * group_cost(g2, item_guides) +
                           static_cast<uint64_t>(t) * (static_cast<uint64_t>(len) * kGuideCost + kTileFixedCost);
                it.tile0 = base.items + i;
                it.last_cands = after < cap ? static_cast<uint32_t>(after) : cap;
                it.group_abs = tile_first_b * 64u + (wstart >> 5);
                it.window = (t == 0 ? gs0 - g2.s0a : 0u) | ((gs1 - wstart < cap ? gs1 - wstart : cap) << 16);
                it.shape = shape; it.gmid = slot0_of[gw];
                stage[w] = it;
            }
            __syncthreads();
            const uint32_t n_here = total - i0 < 256u ? total - i0 : 256u;
            const uint4 *src4 = reinterpret_cast<const uint4 *>(stage);
            uint4 *dst4 = reinterpret_cast<uint4 *>(fitems + base.items + i0);
            for (uint32_t q = w; q < n_here * 3u; q += 256u) dst4[q] = src4[q];
            __syncthreads();
        }
    }
    // One guide per thread and step: its index, scan word and signature are loaded once (a chain of two round trips),
    // its 13 (or 1) places come from registers and LDS.  (One (guide, way) pair per thread and step repeated that chain
    // 13 times over: 0.18 ms at 100 k guides, two thirds of the binning.)
    const uint32_t g0 = gstart[b], n = gfill[b];
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const uint32_t guide = gidx[g0 + i], word = gword[g0 + i];
        const uint64_t gsig = guides[guide];
        const uint32_t gj = succ_byte(gsig, slice, v.slice_width);
        const uint32_t word12 = fine_word(word, slice, v.slice_width);
#pragma unroll
        for (uint32_t way = 0; way < ways; ++way) {
            const uint32_t ww = fine_way(gj, way);
            if (!has_cands[ww]) continue; // no candidates there: the group has no slots
            const uint32_t slot = way ? slot_of[ww] + atomicAdd(&cursor[ww], 1u) : slot0_of[ww] + atomicAdd(&cursor0[ww], 1u);
            fword[slot] = word12 | (fine_class(way) << 24);
            fmeta[slot] = FineMeta{guide, (b << 8) | ww, gsig};
        }
    }
}

// Cost ranges of the pruned scan (the bucket-level ones are resolved by k_guide_scatter's last workgroups).
__global__ __launch_bounds__(256) void k_fine_ranges(const PlanInfo *__restrict__ plan, const ScanItem *__restrict__ fitems,
                                                     RangeStart *__restrict__ starts)
{
    short_kernel_priority();
    if (!plan->fine) return;
    const uint32_t n_ranges = plan->n_ranges;
    const uint32_t r = blockIdx.x * 4u + (threadIdx.x >> 6); // one wave per range start
    if (r <= n_ranges && n_ranges != 0) {
        const RangeStart st = range_start_of<true>(fitems, plan->n_items, plan->total_cost, n_ranges, r);
        if ((threadIdx.x & 63u) == 0u) starts[r] = st;
    }
}

// ---- a small batch: the whole binning in ONE launch -------------------------------------------------------------------
// Seven dependent launches bin a batch (histogram, plan, scatter, group counts, group plan, group scatter, ranges); for a
// page of a few dozen guides they are 48 us of a 110 us step, and nearly all of that is launch boundaries and the chains of
// dependent loads behind each.  A batch of up to kSmallPairs (guide, slice) pairs (102 guides of five slices) is planned here
// without any grouping: EVERY (guide, slice, way) placement becomes a group of its own -- eight slots, the guide in the
// first --, so there is nothing to count, sort or scatter: one thread per (guide, slice) pair looks its 13 (or 1) groups up, a
// prefix sum over the pairs lays out slots, items and costs.  One workgroup PER WAY: each of them makes the whole prefix
// (loads that hit the L2) and writes the slots and items of its own way -- the stores of 4160 placements from one CU alone
// took 40 us.  Two guides that would have shared a group fetch its units twice; at this size that is nothing.  Same slots /
// items / plan as k_fine_* leave behind (k_fine_ranges follows); the bucket-level plan is not made (the host takes this path
// only where the pruned plan wins anyway: small_bin_ok).
constexpr uint32_t kSmallPairs = 512;

__device__ inline uint64_t block512_exclusive_scan(uint64_t v, uint64_t *lds /*[8]*/, uint64_t *total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t incl = wave_inclusive_scan_u64(v);
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    uint64_t before = 0, all = 0;
    for (uint32_t w = 0; w < kSmallPairs / 64u; ++w) {
        const uint64_t x = lds[w];
        if (w < wave) before += x;
        all += x;
    }
    if (total) *total = all;
    __syncthreads();
    return before + incl - v;
}

template <uint32_t WAYS>
__global__ __launch_bounds__(kSmallPairs) void k_bin_small(ImageView v, Workspace ws, const uint64_t *__restrict__ guides, uint32_t n,
                                                           uint32_t prune_mode, uint32_t tail_shapes, uint32_t scan_blocks,
                                                           uint32_t sorted_layout)
{
    short_kernel_priority();
    __shared__ uint64_t lds[kSmallPairs / 64u];
    const uint32_t t = threadIdx.x, my_way = blockIdx.x; // gridDim.x == WAYS
    // what k_guide_hist resets (shared out among the workgroups)
    for (uint32_t k = my_way * kSmallPairs + t; k <= n; k += WAYS * kSmallPairs) ws.gcount[k] = 0;
    for (uint32_t k = my_way * kSmallPairs + t; k <= ws.cap_chunks; k += WAYS * kSmallPairs) ws.raw_used[k] = 0;
    if (t == 0 && my_way == 0) {
        ws.scan_span[2u * ws.span_slot] = ~0ull;
        ws.scan_span[2u * ws.span_slot + 1u] = 0ull;
    }
    const uint32_t pairs = n * v.n_slices;
    const uint32_t low = (1u << v.slice_width) - 1u;
    const bool mine = t < pairs;
    const uint32_t g = mine ? t / v.n_slices : 0u, sl = mine ? t - g * v.n_slices : 0u;
    // the pair's bucket, its 13 (1) groups and what they add; pairs in thread order, ways in order = the order of slots and items
    uint64_t sig = 0, blen = 0, units = 0, cost = 0, cand = 0, ref = 0, valid = 0;
    uint64_t cost_before = 0;               // ... of the pair's ways in front of this workgroup's
    uint32_t units_before = 0, valid_before = 0;
    uint32_t b = 0, gj = 0, tf = 0, my_s0 = 0, my_s1 = 0;
    if (mine) {
        sig = guides[g];
        b = (sl << v.slice_width) + (static_cast<uint32_t>(sig >> (v.slice_width * sl)) & low);
        gj = succ_byte(sig, sl, v.slice_width);
        const uint32_t *ss = v.sub_start + static_cast<uint64_t>(b) * 257u;
        uint32_t s0[WAYS], s1[WAYS];
#pragma unroll
        for (uint32_t way = 0; way < WAYS; ++way) { // (all in flight together)
            const uint32_t ww = fine_way(gj, way);
            s0[way] = ss[ww];
            s1[way] = ss[ww + 1];
        }
        blen = v.bucket_start[b + 1] - v.bucket_start[b];
        tf = v.tile_first[b];
        ref = blen;
#pragma unroll
        for (uint32_t way = 0; way < WAYS; ++way) {
            if (way == my_way) { my_s0 = s0[way]; my_s1 = s1[way]; units_before = static_cast<uint32_t>(units); cost_before = cost; valid_before = static_cast<uint32_t>(valid); }
            if (s1[way] > s0[way]) {
                const GroupUnits gu = group_units(s0[way], s1[way], tail_shapes);
                units += gu.units;
                cost += group_cost(gu, 1u);
                cand += s1[way] - s0[way];
                ++valid;
            }
        }
    }
    uint64_t t_units, t_cost, t_cand, t_ref, t_valid;
    uint32_t item_at = static_cast<uint32_t>(block512_exclusive_scan(units, lds, &t_units)) + units_before;
    uint64_t cost_at = block512_exclusive_scan(cost, lds, &t_cost) + cost_before;
    (void)block512_exclusive_scan(cand, lds, &t_cand);
    (void)block512_exclusive_scan(ref, lds, &t_ref);
    const uint32_t slot_at = (static_cast<uint32_t>(block512_exclusive_scan(valid, lds, &t_valid)) + valid_before) * kGuideGroup;
    const uint64_t t_slots = t_valid * kGuideGroup;
    const bool overflow = t_units > ws.cap_fitems || t_slots > ws.cap_fslots;
    // slots and items of this workgroup's way of every pair
    if (mine && !overflow && my_s1 > my_s0) {
        const uint32_t way = my_way, s0 = my_s0, s1 = my_s1;
        const uint32_t word12 = fine_word(image_word(sig, sl, v.slice_width, sorted_layout != 0u), sl, v.slice_width);
        const uint32_t ww = fine_way(gj, way);
        const uint32_t c1 = way ? 1u : 0u; // a group's class-1 guides come first, its class-0 guides from gmid on: here ONE guide
        uint4 *fw = reinterpret_cast<uint4 *>(ws.fword + slot_at); // (slot_at is a multiple of 8: 32-byte aligned)
        fw[0] = make_uint4(word12 | (fine_class(way) << 24), kPadGuideWord, kPadGuideWord, kPadGuideWord);
        fw[1] = make_uint4(kPadGuideWord, kPadGuideWord, kPadGuideWord, kPadGuideWord);
        ws.fmeta[slot_at] = FineMeta{g, (b << 8) | ww, sig};
#pragma unroll
        for (uint32_t k2 = 1; k2 < kGuideGroup; ++k2) ws.fmeta[slot_at + k2] = FineMeta{kNoGuide, 0u, 0ull};
        const GroupUnits gu = group_units(s0, s1, tail_shapes);
        for (uint32_t u = 0; u < gu.units; ++u) { // as k_fine_scatter lays a group's units out, for one guide
            const bool full = u < gu.n_full;
            const uint32_t shape = full ? 32u : gu.shape, cap = 64u * shape;
            const uint32_t wstart = gu.s0a + u * kTileCands;
            const uint64_t after = blen - wstart;
            ScanItem it;
            it.bucket = (b << 8) | ww;
            it.g0 = slot_at;
            it.g1 = slot_at + 1u;
            it.n_tiles = 1;
            it.cost0 = cost_at;
            it.tile0 = item_at;
            it.last_cands = after < cap ? static_cast<uint32_t>(after) : cap;
            it.group_abs = tf * 64u + (wstart >> 5);
            it.window = (u == 0 ? s0 - gu.s0a : 0u) | ((s1 - wstart < cap ? s1 - wstart : cap) << 16);
            it.shape = shape; it.gmid = slot_at + c1;
            ws.fitems[item_at++] = it;
            cost_at += static_cast<uint64_t>(shape >> 3) + kTileFixedCost;
        }
    }
    if (t == 0 && my_way == 0) {
        const uint32_t n_items = overflow ? 0u : static_cast<uint32_t>(t_units);
        ScanItem end;
        end.bucket = 0; end.g0 = 0; end.g1 = 0; end.n_tiles = 0; end.cost0 = overflow ? 0ull : t_cost;
        end.tile0 = n_items; end.last_cands = 0; end.group_abs = 0; end.window = 0; end.shape = 32; end.gmid = 0;
        ws.fitems[n_items] = end;
        PlanInfo pl{};
        pl.n_items = n_items;
        pl.error = 0;
        pl.n_ranges = n_items == 0 ? 0u : ranges_for(t_units, scan_blocks);
        pl.fine = prune_mode;
        pl.total_cost = overflow ? 0ull : t_cost;
        pl.candidates = overflow ? 0ull : t_cand;
        pl.reference_candidates = t_ref;
        pl.tiles = n_items;
        pl.fine_slots = overflow ? 0u : static_cast<uint32_t>(t_slots);
        *ws.plan = pl;
        Counters c{};
        c.raw_chunks = (pl.n_ranges ? pl.n_ranges : 1u) * 16u;
        *ws.counters = c;
        if (overflow) { // room for the next try (finish_batches enlarges the item list), and this batch once more
            atomicMax(&ws.sticky[3], static_cast<uint32_t>(t_units < 0xFFFFFFFFull ? t_units : 0xFFFFFFFFull));
            atomicOr(&ws.sticky[0], 2u);
        }
    }
}

// The one-launch binning where it is safe and pays: a sorted image (the pruned plan exists), at most kSmallPairs
// (guide, slice) pairs, 13 ways or 1 (max_dist <= 4), and an index on which the pruned plan beats the bucket-level one for a lone
// guide anyway -- its buckets hold more units than the 13 groups a guide visits (k_fine_plan's estimate, taken for the mean
// bucket) -- or a caller who asked for the pruned plan always (prune = 1).
static bool small_bin_ok(const ImageView &v, const Workspace &ws, const Tuning &tn, uint32_t n, uint32_t prune_mode)
{
    if (!tn.small_bin || (prune_mode != 1u && prune_mode != 2u) || !ws.fitems || !v.sub_start) return false;
    if (n == 0 || static_cast<uint64_t>(n) * v.n_slices > kSmallPairs) return false;
    const uint64_t buckets_per_slice = 1ull << v.slice_width;
    return tn.prune == 1 || v.n_sites / buckets_per_slice >= 16ull * kTileCands;
}

uint32_t prune_mode_for(const ImageView &v, const Tuning &tn, uint32_t n_guides, int max_dist)
{
    const bool geometry = v.n_slices * v.slice_width == 40u && (v.slice_width == 8 || v.slice_width == 4 || v.slice_width == 2); // succ_byte
    if ((!v.srec && !v.sid) || tn.prune == 0 || max_dist < 0 || max_dist > 5 || !geometry) return 0;
    if (n_guides > prune_max_guides(max_dist == 5 ? 3u : 2u, v.n_slices)) return 0;
    // max_dist 5: a hit the reference can find matches some slice exactly (:330-344 walks the buckets of the guide's own
    // slice values), and then some exact slice is followed by one with at most TWO mismatches (the cycle lemma of the
    // comment above with 3 |E| + (5 - 2 |E|) > 5): 67 of a bucket's 256 groups instead of all of them.
    return max_dist <= 2 ? 1u : max_dist <= 4 ? 2u : 3u;
}

void launch_bin_guides(const ImageView &v, const Workspace &ws, const Tuning &tn, const uint64_t *d_guides, uint32_t n,
                       uint32_t prune_mode, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const uint32_t nb = v.n_buckets;
    if (small_bin_ok(v, ws, tn, n, prune_mode)) { // two launches instead of seven
        const uint32_t sorted = (v.srec || v.sid) ? 1u : 0u;
        if (fine_ways_of(prune_mode) == 1u)
            hipLaunchKernelGGL(k_bin_small<1u>, dim3(1), dim3(kSmallPairs), 0, stream, v, ws, d_guides, n, prune_mode,
                               static_cast<uint32_t>(tn.tail_shapes), tn.scan_blocks, sorted);
        else
            hipLaunchKernelGGL(k_bin_small<kFineWays>, dim3(kFineWays), dim3(kSmallPairs), 0, stream, v, ws, d_guides, n, prune_mode,
                               static_cast<uint32_t>(tn.tail_shapes), tn.scan_blocks, sorted);
        hipLaunchKernelGGL(k_fine_ranges, dim3((tn.scan_blocks + 1u + 3u) / 4u), dim3(256), 0, stream, ws.plan, ws.fitems, ws.range_start);
        return;
    }
    // slots in use: 8-padded guides per bucket, at most n * slices + 8 * buckets
    const uint32_t n_slots = static_cast<uint32_t>(
        std::min<size_t>(ws.cap_gslots, static_cast<size_t>(n) * v.n_slices + static_cast<size_t>(kGuideGroup) * nb));
    // three launches: histogram (+ resets), plan, scatter (+ ranges)
    const uint32_t blocks = (n + 255u) / 256u;
    const uint32_t reset_blocks = std::min<uint32_t>(1024u, (std::max(n_slots, nb) + 255u) / 256u);
    hipLaunchKernelGGL(k_guide_hist, dim3(std::max(blocks, reset_blocks)), dim3(256), 0, stream, ws, d_guides, n,
                       v.slice_width, v.n_slices, nb, n_slots, tn.scan_blocks * 16u);
    hipLaunchKernelGGL(k_plan, dim3(1), dim3(256), 0, stream, v, ws.ng, ws.gfill, ws.gstart, ws.items,
                       static_cast<uint32_t>(ws.cap_items), ws.plan, tn.item_guides, tn.scan_blocks, ws.counters);
    const uint32_t range_blocks = (tn.scan_blocks + 1u + 255u) / 256u;
    hipLaunchKernelGGL(k_guide_scatter, dim3(blocks + range_blocks), dim3(256), 0, stream, d_guides, n, v.slice_width,
                       v.n_slices, (v.srec || v.sid) ? 1u : 0u, nb, ws.gstart, ws.gfill, ws.gword, ws.gidx, ws.gbucket, blocks, ws.plan, ws.items,
                       ws.range_start);
    if (prune_mode) { // regroup by (bucket, successor byte); k_fine_plan decides which of the two plans the scan follows
        const uint32_t ways = fine_ways_of(prune_mode);
        // (max_dist 5: three classes of guides in a pass -- the class plane of the short units has weight one only)
        const uint32_t tail_shapes = prune_mode == 3 ? 0u : static_cast<uint32_t>(tn.tail_shapes);
        auto launch_fine = [&](auto ways_tag) {
            constexpr uint32_t W = decltype(ways_tag)::value;
            hipLaunchKernelGGL(k_fine_count<W>, dim3(nb), dim3(256), 0, stream, v, d_guides, ws.gstart, ws.gfill, ws.gidx, ws.fcount,
                               ws.fcount0, ws.fsum, tn.item_guides, tail_shapes);
            hipLaunchKernelGGL(k_fine_plan, dim3(1), dim3(256), 0, stream, ws.fsum, nb, ws.fitems,
                               static_cast<uint32_t>(ws.cap_fitems), static_cast<uint32_t>(ws.cap_fslots), ws.plan, tn.scan_blocks,
                               prune_mode, tn.prune == 1 ? 1u : 0u, ws.sticky, ws.counters);
            hipLaunchKernelGGL(k_fine_scatter<W>, dim3(nb), dim3(256), 0, stream, v, d_guides, ws.gstart, ws.gfill, ws.gword, ws.gidx,
                               ws.fcount, ws.fcount0, ws.fsum, ws.plan, ws.fword, ws.fmeta, ws.fitems, tn.item_guides, tail_shapes);
        };
        if (ways == 1u) launch_fine(std::integral_constant<uint32_t, 1u>{});
        else if (ways == kFineWays) launch_fine(std::integral_constant<uint32_t, kFineWays>{});
        else launch_fine(std::integral_constant<uint32_t, kFineWays2>{});
        hipLaunchKernelGGL(k_fine_ranges, dim3((tn.scan_blocks + 1u + 3u) / 4u), dim3(256), 0, stream, ws.plan, ws.fitems, ws.range_start);
    }
}

// ------------------------------------------------------------------------------------------------
// scan
// ------------------------------------------------------------------------------------------------

struct alignas(4 * kGuideGroup) GuideGroup {
    uint32_t w[kGuideGroup];
};


// ---- raw records ------------------------------------------------------------------------------
// A candidate that the scan finds within max_dist of a guide is only NOTED by the scan kernel, as an
// 8-byte record (guide slot, tile, offset in tile), with plain stores into a chunk of the raw buffer
// that the wave owns -- no dependent load, no returning atomic on the hot path (one hit per ~50k
// comparisons is frequent enough that a latency chain per hit would dominate the kernel).
// k_verify then checks every record exactly, applies the first-matching-slice rule and turns the
// survivors into keys.  Chunk = kChunkRecs slots of 8 bytes, slot 0 unused; raw_used[chunk] = slots in use (slot 0 included).
__device__ __forceinline__ uint64_t raw_record(uint32_t gslot, uint32_t tile, uint32_t offset)
{
    return (static_cast<uint64_t>(gslot) << 37) | (static_cast<uint64_t>(tile) << 11) | offset;
}

struct RawWriter {
    uint64_t *chunk;   // current chunk of this wave (wave-uniform)
    uint32_t fill;     // used slots of the current chunk, header included
    uint32_t left;     // further chunks of the wave's current reservation (they follow the current one)
    uint32_t reserve;  // chunks the next reservation takes: 1, 2, 4 ... 16 -- a wave in a hit-dense bucket fills a chunk
                       // every few guides, and every reservation is a returning atomic that the wave waits for with all
                       // its record stores; a wave with few hits never reserves a chunk it does not use
};

__device__ __forceinline__ void raw_retire(const RawWriter &w, uint32_t lane, const uint64_t *raw, uint32_t *raw_used)
{
    if (lane == 0) raw_used[(w.chunk - raw) / kChunkRecs] = w.fill;
}

__device__ __forceinline__ void raw_acquire(RawWriter &w, uint64_t *raw, uint32_t max_chunks, Counters *counters,
                                            uint32_t lane)
{
    w.fill = 1;
    if (w.left != 0u) { // the next chunk of the reservation: its header was cleared with all the others (k_guide_hist)
        w.chunk += kChunkRecs;
        w.left -= 1u;
        return;
    }
    const uint32_t take = w.reserve;
    uint32_t idx = 0;
    if (lane == 0) idx = atomicAdd(&counters->raw_chunks, take);
    idx = __builtin_amdgcn_readfirstlane(idx);
    if (idx + take > max_chunks) { // buffer exhausted: write into the spare chunk, the host grows the buffer and re-runs
        idx = max_chunks;
        if (lane == 0) counters->raw_overflow = 1u;
        w.chunk = raw + static_cast<uint64_t>(idx) * kChunkRecs;
        return; // (left stays 0: every further chunk comes here again)
    }
    w.chunk = raw + static_cast<uint64_t>(idx) * kChunkRecs;
    w.left = take - 1u;
    if (take < 16u) w.reserve = take * 2u;
}

// ---- bit-sliced distance test -------------------------------------------------------------------
// 3:2 and 2:2 counters on bit planes; v_bitop3_b32 evaluates any 3-input boolean function in one op.
__device__ __forceinline__ void full_add(uint32_t a, uint32_t b, uint32_t c, uint32_t &sum, uint32_t &carry)
{
    sum = __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);   // a ^ b ^ c
    carry = __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); // majority(a, b, c)
}
__device__ __forceinline__ void half_add(uint32_t a, uint32_t b, uint32_t &sum, uint32_t &carry)
{
    sum = a ^ b;
    carry = a & b;
}

// For the lane's 32 candidates (planes c[]) and one guide scan word gw: the plane of candidates whose
// mismatch count over the 16 positions is <= THR (THR = 0..4 compiled in; THR < 0: runtime `thr`, any value).
// Position p mismatches iff low or high bit differs: (c[p] ^ G0p) | (c[16+p] ^ G1p) with the guide's bits
// broadcast to all-zero / all-one scalars (isslScoreOfftargets.cpp:376-380 in transposed form); the 16 mismatch
// planes are then counted with a carry-save adder tree.
// `keep`: the lane's candidates that belong to the item (the others are padding or a neighbouring group's); folded into
// the last operation of the count, which has an operand to spare for every compiled threshold but 1.
template <int THR>
__device__ __forceinline__ uint32_t count_near(const uint32_t (&m)[16], uint32_t thr, uint32_t keep);

template <int THR>
__device__ __forceinline__ uint32_t near_plane(const uint32_t (&c)[kPlanes], uint32_t gw, uint32_t thr, uint32_t keep)
{
    uint32_t m[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const uint32_t g0 = 0u - ((gw >> p) & 1u);
        const uint32_t g1 = 0u - ((gw >> (16 + p)) & 1u);
        m[p] = (c[p] ^ g0) | (c[16 + p] ^ g1);
    }
    return count_near<THR>(m, thr, keep);
}

// The same test in a SHORT unit: every plane register holds the lane's 16 / 8 candidates two / four times over, and
// mask word p (32 words in LDS, made by the wave itself: short_unit_masks) carries bit p of two / four guides' scan
// words, each spread over its field: one pass, two / four guides.  The masks arrive as wave-uniform VGPRs (every lane
// reads the same 16 bytes), four positions at a time so that they never occupy more than a handful of registers.
template <int THR>
__device__ __forceinline__ uint32_t near_plane_masks(const uint32_t (&c)[kPlanes], const uint4 *gm /*LDS, 8 x uint4*/,
                                                     uint32_t thr, uint32_t keep)
{
    uint32_t m[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint4 lo = gm[q], hi = gm[4 + q];
        asm volatile("" ::: "memory"); // (keeps the next loads behind these: at most two steps' masks are live)
        m[4 * q + 0] = (c[4 * q + 0] ^ lo.x) | (c[16 + 4 * q + 0] ^ hi.x);
        m[4 * q + 1] = (c[4 * q + 1] ^ lo.y) | (c[16 + 4 * q + 1] ^ hi.y);
        m[4 * q + 2] = (c[4 * q + 2] ^ lo.z) | (c[16 + 4 * q + 2] ^ hi.z);
        m[4 * q + 3] = (c[4 * q + 3] ^ lo.w) | (c[16 + 4 * q + 3] ^ hi.w);
    }
    return count_near<THR>(m, thr, keep);
}

// Masks of up to 8 passes of a short unit (guide slots g0 .. g0 + 8 * per - 1) into the wave's own 1 KiB of LDS: lane
// (pass i, quarter q) makes the four words 4q .. 4q + 3 of pass i: bit p of each of the pass's `per` guide words, spread
// over that guide's field of `shape` bits.
__device__ __forceinline__ void short_unit_masks(const uint32_t *__restrict__ gword_stream, uint32_t g0, uint32_t shape,
                                                 uint32_t lane, uint4 *lds_masks)
{
    const uint32_t per = 32u / shape, field = shape == 16u ? 0xFFFFu : 0xFFu;
    const uint32_t i = lane >> 3, q = lane & 7u;
    uint32_t gw[4];
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) gw[j] = j < per ? gword_stream[g0 + i * per + j] : 0u;
    uint32_t m[4];
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r) {
        const uint32_t p = 4u * q + r;
        uint32_t x = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) x |= (0u - ((gw[j] >> p) & 1u)) & (field << ((j * shape) & 31u)) & (j < per ? ~0u : 0u);
        m[r] = x;
    }
    lds_masks[i * 8u + q] = make_uint4(m[0], m[1], m[2], m[3]);
}

// The planes of candidates whose mismatch planes m[0..15] count up to at most THR.
template <int THR>
__device__ __forceinline__ uint32_t count_near(const uint32_t (&m)[16], uint32_t thr, uint32_t keep)
{
    uint32_t s0, s1, s2, s3, s4, t, u, n0, n1, n2;
    uint32_t k2[8], k4[4], k8a, k8b;
    // weight 1: 16 planes
    full_add(m[0], m[1], m[2], s0, k2[0]);
    full_add(m[3], m[4], m[5], s1, k2[1]);
    full_add(m[6], m[7], m[8], s2, k2[2]);
    full_add(m[9], m[10], m[11], s3, k2[3]);
    full_add(m[12], m[13], m[14], s4, k2[4]);
    full_add(s0, s1, s2, t, k2[5]);
    full_add(s3, s4, m[15], u, k2[6]);
    half_add(t, u, n0, k2[7]);
    // weight 2: 8 planes
    uint32_t a2, b2, c2;
    full_add(k2[0], k2[1], k2[2], a2, k4[0]);
    full_add(k2[3], k2[4], k2[5], b2, k4[1]);
    full_add(k2[6], k2[7], a2, c2, k4[2]);
    half_add(b2, c2, n1, k4[3]);
    // weight 4: 4 planes k4[0..3], S4 = how many of them are set.  count = 4 S4 + 2 n1 + n0, so the compiled thresholds
    // need S4 only as "none", "at least one", "at least two" -- cheaper than adding the four planes up:
    //   count <= 3  <=>  S4 == 0;   count <= 4  <=>  S4 == 0 or (S4 == 1 and n1 == n0 == 0)
    // with o = k4[0] | k4[1] | k4[2] and p = majority(k4[0], k4[1], k4[2]):  S4 >= 1 = o | k4[3],
    // S4 >= 2 = p | (o & k4[3]), and  S4 >= 2 or (S4 >= 1 and w)  =  p | majority(o, k4[3], w).
    if (THR >= 0 && THR <= 4) {
        const uint32_t o = __builtin_amdgcn_bitop3_b32(k4[0], k4[1], k4[2], 0xFE); // a | b | c
        // (the last operation of each case is ~(a | b) & keep as one bitop3: table 0x02)
        if (THR == 0) return __builtin_amdgcn_bitop3_b32(o, __builtin_amdgcn_bitop3_b32(k4[3], n1, n0, 0xFE), keep, 0x02);
        if (THR == 1) return __builtin_amdgcn_bitop3_b32(o, k4[3], n1, 0x01) & keep;     // ~(a | b | c)
        if (THR == 2) return __builtin_amdgcn_bitop3_b32(o, __builtin_amdgcn_bitop3_b32(k4[3], n1, n0, 0xF8), keep, 0x02); // a | (b & c)
        if (THR == 3) return __builtin_amdgcn_bitop3_b32(o, k4[3], keep, 0x02);
        const uint32_t p = __builtin_amdgcn_bitop3_b32(k4[0], k4[1], k4[2], 0xE8);            // majority
        const uint32_t z = __builtin_amdgcn_bitop3_b32(o, k4[3], n1 | n0, 0xE8);
        return __builtin_amdgcn_bitop3_b32(p, z, keep, 0x02);
    }
    uint32_t a4;
    full_add(k4[0], k4[1], k4[2], a4, k8a);
    half_add(a4, k4[3], n2, k8b);
    // generic threshold: count = n0 + 2 n1 + 4 n2 + 8 n3 + 16 n4, compared MSB first with the uniform thr
    const uint32_t n[5] = {n0, n1, n2, k8a ^ k8b, k8a & k8b};
    uint32_t gt = 0u, eq = ~0u;
#pragma unroll
    for (int b = 4; b >= 0; --b) {
        if ((thr >> b) & 1u) {
            eq &= n[b];
        } else {
            gt |= eq & n[b];
            eq &= ~n[b];
        }
    }
    return keep & ~gt;
}

// ---- the pruned scan's test: 12 positions (fine_word) -------------------------------------------------------------
// The planes of candidates whose 12 mismatch planes m[] (+ the class plane f when EXTRA) count up to at most B
// (B = 0..4 compiled in; B < 0: runtime `thr`).  count = n0 + 2 n1 + 4 S4 with S4 the number of set planes among the
// three of weight 4 (k4[0], k4[1], a2 & b2), so the compiled budgets need the weight-2 sums a2, b2 only through a handful
// of three-input functions: 41 vector operations per pass for B = 3 (the class-1 guides of max_dist 4: twelve of
// thirteen), 46 for B = 4, against 62 for the 16-position test.
// `prev` (out): three planes whose OR is "some position of the previous slice mismatches" -- m[0..3] are that slice's
// (fine_order), and m[0] | m[1] | m[2] = sum | carry of the first adder.  Only the cold block looks at them (fine_dup).
struct PrevSlice {
    uint32_t a, b, c;
};
template <int B, bool EXTRA>
__device__ __forceinline__ uint32_t count_near12(const uint32_t (&m)[12], uint32_t f, uint32_t thr, uint32_t keep, PrevSlice &prev)
{
    uint32_t s0, s1, s2, s3, t, n0, a2, b2, k2[6], k40, k41;
    full_add(m[0], m[1], m[2], s0, k2[0]);
    prev.a = s0; prev.b = k2[0]; prev.c = m[3];
    full_add(m[3], m[4], m[5], s1, k2[1]);
    full_add(m[6], m[7], m[8], s2, k2[2]);
    full_add(m[9], m[10], m[11], s3, k2[3]);
    full_add(s0, s1, s2, t, k2[4]);
    if (EXTRA) full_add(t, s3, f, n0, k2[5]);
    else half_add(t, s3, n0, k2[5]);
    full_add(k2[0], k2[1], k2[2], a2, k40);
    full_add(k2[3], k2[4], k2[5], b2, k41);
    // bitop3 tables: bit (4a + 2b + c) of the constant is f(a, b, c)
    if (B >= 0 && B <= 3) {
        const uint32_t r1 = __builtin_amdgcn_bitop3_b32(k40, k41, keep, 0x02);              // ~(a | b) & c: no weight-4 plane among the first two
        if (B == 3) return __builtin_amdgcn_bitop3_b32(r1, a2, b2, 0x70);                   // a & ~(b & c): S4 == 0
        if (B == 1) return __builtin_amdgcn_bitop3_b32(r1, a2, b2, 0x10);                   // a & ~(b | c): S4 == 0 and n1 == 0
        if (B == 0) return __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_bitop3_b32(r1, a2, b2, 0x10), n0, n0, 0x30); // ... and n0 == 0
        const uint32_t r2 = __builtin_amdgcn_bitop3_b32(r1, a2, b2, 0x70);
        return __builtin_amdgcn_bitop3_b32(r2, a2 ^ b2, n0, 0x70);                          // B == 2: S4 == 0 and not (n1 and n0)
    }
    const uint32_t k42 = a2 & b2;
    if (B == 4) { // count <= 4  <=>  not (S4 >= 2 or (S4 >= 1 and (n1 or n0)))
        const uint32_t w = __builtin_amdgcn_bitop3_b32(a2, b2, n0, 0xBE);                   // (a ^ b) | c
        const uint32_t p = __builtin_amdgcn_bitop3_b32(k40, k41, k42, 0xE8);                // majority
        const uint32_t o = __builtin_amdgcn_bitop3_b32(k40, k41, k42, 0xFE);                // a | b | c
        const uint32_t x = __builtin_amdgcn_bitop3_b32(o, w, keep, 0x2A);                   // ~(a & b) & c
        return __builtin_amdgcn_bitop3_b32(x, p, p, 0x30);                                  // a & ~b
    }
    // runtime budget: count = n0 + 2 n1 + 4 n2 + 8 n3 (<= 13), compared MSB first with the uniform thr (< 16)
    uint32_t n2, n3;
    full_add(k40, k41, k42, n2, n3);
    const uint32_t n[4] = {n0, a2 ^ b2, n2, n3};
    uint32_t gt = 0u, eq = ~0u;
#pragma unroll
    for (int b = 3; b >= 0; --b) {
        if ((thr >> b) & 1u) {
            eq &= n[b];
        } else {
            gt |= eq & n[b];
            eq &= ~n[b];
        }
    }
    return keep & ~gt;
}

// Full unit of the pruned scan: the lane's 32 candidates (c[0..11] low, c[12..23] high code bits of the 12 positions)
// against one guide word of the pruned plan (fine_word), budget B.
template <int B>
__device__ __forceinline__ uint32_t near_plane12(const uint32_t (&c)[24], uint32_t gw, uint32_t thr, uint32_t keep, PrevSlice &prev)
{
    uint32_t m[12];
#pragma unroll
    for (int p = 0; p < 12; ++p) {
        const uint32_t g0 = 0u - ((gw >> p) & 1u);
        const uint32_t g1 = 0u - ((gw >> (12 + p)) & 1u);
        m[p] = (c[p] ^ g0) | (c[12 + p] ^ g1);
    }
    return count_near12<B, false>(m, 0u, thr, keep, prev);
}

// Short unit of the pruned scan: two / four guides per pass, the masks from the wave's LDS (short_unit_masks: word p of a
// pass = bit p of its guides' words, spread over their fields; word 24 = the class bits: a plane of weight one).
template <int B>
__device__ __forceinline__ uint32_t near_plane12_masks(const uint32_t (&c)[24], const uint4 *gm /*LDS, 8 x uint4*/,
                                                       uint32_t thr, uint32_t keep, PrevSlice &prev)
{
    uint32_t m[12];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const uint4 lo = gm[q], hi = gm[3 + q];
        asm volatile("" ::: "memory"); // (keeps the next loads behind these: at most two steps' masks are live)
        m[4 * q + 0] = (c[4 * q + 0] ^ lo.x) | (c[12 + 4 * q + 0] ^ hi.x);
        m[4 * q + 1] = (c[4 * q + 1] ^ lo.y) | (c[12 + 4 * q + 1] ^ hi.y);
        m[4 * q + 2] = (c[4 * q + 2] ^ lo.z) | (c[12 + 4 * q + 2] ^ hi.z);
        m[4 * q + 3] = (c[4 * q + 3] ^ lo.w) | (c[12 + 4 * q + 3] ^ hi.w);
    }
    const uint32_t f = reinterpret_cast<const uint32_t *>(gm)[24];
    return count_near12<B, true>(m, f, thr, keep, prev);
}

// Guide words that start at any slot (a class boundary is no multiple of 8): a scalar load needs its address dword-aligned only.
// A candidate that also matches the guide exactly in the slice BEFORE the bucket's own is met in that slice's bucket too --
// in the group of the guide's own successor byte, which every guide is placed in -- and the smaller slice reports it
// (k_verify's reporter rule): the record this unit would note is one k_verify reads 16 random bytes for and
// throws away.  58 % of the duplicate records of a hit at distance 4 are of this kind (enumerated in
// tests/test_oracle_golden.py); dropping them here costs two operations in the cold block.
__device__ __forceinline__ uint32_t fine_dup(uint32_t ok, const PrevSlice &prev, uint32_t dup_filter)
{
    const uint32_t prev_mismatch = __builtin_amdgcn_bitop3_b32(prev.a, prev.b, prev.c, 0xFE); // a | b | c
    return ok & (prev_mismatch | ~dup_filter);
}

struct alignas(4) GuideGroupAny {
    uint32_t w[kGuideGroup];
};

// Cold block of the scan: the wave knows that SOME lane has a candidate within thr of a guide.  `ok` = this lane's
// plane of such candidates; bit q of it is candidate q % (1 << w_log) of the lane -- which sits at offset off0 + that of
// `tile` (the lane's own: a window of the pruned scan straddles two tiles) -- against the guide in slot gslot + (q >>
// w_log) (full units: w_log = 5, one guide per pass).
__device__ __forceinline__ void note_candidates(uint32_t ok, uint32_t gslot, uint32_t w_log, uint32_t tile, uint32_t off0,
                                                uint32_t lane, RawWriter &w, uint64_t *raw, uint32_t *raw_used,
                                                uint32_t max_chunks, Counters *counters)
{
    uint64_t who = __ballot(ok != 0u);
    if (who == 0ull) return;
    // One round per candidate of the lane that has most -- almost always ONE: the loop is laid out with its first round as the
    // straight path (a taken branch drains the wave's instruction buffer, and more than half of the passes come through
    // here: k_scan<4> -3 % same-box, profiles/r05_ab_scan_peel_lanes3.log).
    do {
        const uint32_t n = static_cast<uint32_t>(__builtin_popcountll(who));
        if (__builtin_expect(w.fill + n > kChunkRecs, 0)) {
            raw_retire(w, lane, raw, raw_used);
            raw_acquire(w, raw, max_chunks, counters, lane);
        }
        if (ok != 0u) {
            const uint32_t q = static_cast<uint32_t>(__builtin_ctz(ok));
            ok &= ok - 1u;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(who >> 32),
                                                            __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(who), 0u));
            w.chunk[w.fill + rank] = raw_record(gslot + (q >> w_log), tile, off0 + (q & ((1u << w_log) - 1u)));
        }
        w.fill += n;
        who = __ballot(ok != 0u);
    } while (__builtin_expect(who != 0ull, 0));
}

// Scan kernel.  A workgroup of 16 waves (two per CU = 8 waves per SIMD) owns one equal-cost range of the work
// and its waves share the tiles of that range through a ticket counter in LDS: the hardware favours the older
// waves of a SIMD, so waves with equal static shares finish anywhere between 30 % and 100 % of the kernel time
// (measured with the scan_stamps knob) and the SIMDs run half empty for the second half; with the LDS tickets all
// waves of a workgroup stop within one tile of each other.  (A device-wide ticket counter would serialise at ~12 ns
// per ticket; an LDS atomic costs a few hundred cycles and no global traffic.)
// Per tile a wave keeps the 2048 candidates in registers (32 bit planes per lane) while the guide words of the
// item stream through scalar registers, 8 per scalar load.  (Two tiles per wave -- the scalar work of a guide shared by
// 4096 candidates, 4 waves per SIMD -- was measured in round 2: 15 % slower, profiles/r02_ab_scan_tiles_*.log; the
// scalar pipe is not what limits the loop, tools/ubench_issue.hip.)
// The streams the hot loop reads (scan planes, tile table, items, guide words, plan) are separate
// `const __restrict__` kernel arguments: they are never written by this kernel, which lets the compiler fetch the
// wave-uniform ones through the scalar cache.
// The work of one scan workgroup.  FINE: the plan of the pruned scan (single-tile items numbered like the units).  A
// function template instantiated once per plan inside k_scan, so that each copy reads its item list and guide words
// through the kernel's own `__restrict__` arguments (a pointer chosen at run time loses the scalar loads).
template <int THR, bool FINE>
__device__ __forceinline__ void scan_range(const uint32_t *__restrict__ scan_stream, const ScanItem *__restrict__ items,
                                           const PlanInfo *__restrict__ plan, const RangeStart *__restrict__ range_start,
                                           const uint32_t *__restrict__ gword_stream, uint4 *wave_masks,
                                           uint64_t *raw, uint32_t *raw_used, uint32_t max_chunks,
                                           Counters *counters, uint32_t thr, unsigned long long *stamps,
                                           uint64_t *__restrict__ scan_count, uint32_t *next_unit_p, uint32_t *waves_done_p,
                                           unsigned long long *wg_compared_p, unsigned long long t_start,
                                           unsigned long long *span, uint32_t n_tiles, uint32_t slice_bits)
{
    uint32_t &next_unit = *next_unit_p;
    uint32_t &waves_done = *waves_done_p;
    unsigned long long &wg_compared = *wg_compared_p;
    constexpr bool fine = FINE;
    uint32_t units_done = 0;
    unsigned long long plane_wait = 0ull; // stamps only: 100 MHz ticks this wave spent waiting for tile planes to arrive
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_id = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    // Raw records: the wave's first chunk is the one with its own number (no atomic); k_guide_hist cleared its header.
    RawWriter w;
    const bool no_own_chunk = wave_id >= max_chunks; // buffer smaller than the wave count: spare chunk + overflow flag
    w.chunk = raw + static_cast<uint64_t>(no_own_chunk ? max_chunks : wave_id) * kChunkRecs;
    w.fill = 1;
    w.left = 0;
    w.reserve = 1;
    bool own_chunk = false;
    unsigned long long compared = 0ull; // (real candidate, real guide) pairs this wave has compared; wave-uniform

    // Work of this workgroup: from `first` up to (not including) `last`; a position is (item, tile of the item,
    // guide offset inside the item in multiples of 8).  Units = tiles, numbered from the first one; the first and
    // the last tile may be shared with the neighbouring workgroups (guide offsets).
    const RangeStart first = range_start[blockIdx.x];
    const RangeStart last = range_start[blockIdx.x + 1];
    const uint32_t tile_begin = items[first.item].tile0 + first.tile;
    const uint32_t tile_last = items[last.item].tile0 + last.tile; // partly ours when last.goff > 0
    const uint32_t n_units = tile_last - tile_begin + (last.goff ? 1u : 0u);
    uint32_t it = first.item;
    ScanItem cur = items[it];

    while (true) {
        uint32_t u = 0;
        if (lane == 0) u = atomicAdd(&next_unit, 1u);
        u = __builtin_amdgcn_readfirstlane(u);
        if (u >= n_units) break;
        ++units_done;
        const uint32_t gt = tile_begin + u;            // tile number in item order
        if (fine) { it = gt; cur = items[gt]; }        // single-tile items: numbered like the units, no search
        else while (gt >= cur.tile0 + cur.n_tiles) cur = items[++it]; // tickets only grow: the cursor moves forward
        const uint32_t k = gt - cur.tile0;
        const uint32_t g_begin = cur.g0 + (u == 0 ? first.goff : 0u);
        const uint32_t g_end = (gt == tile_last) ? cur.g0 + last.goff : cur.g1;

        if constexpr (FINE) {
            // The planes the unit needs: 12 of the 16 positions (fine_word) -- the successor slice's four sit in plane quads
            // sq and 4 + sq of the tile and stay in memory.
            const uint32_t slice_of = cur.bucket >> (8u + slice_bits); // bucket << 8 | successor byte; slice = bucket >> slice width
            const uint32_t q0 = fine_order(slice_of, 0u, slice_bits), q1 = fine_order(slice_of, 1u, slice_bits), q2 = fine_order(slice_of, 2u, slice_bits);
            const uint32_t dup_filter = slice_of != 0u ? ~0u : 0u; // (fine_dup: positions 0..3 are the previous slice's)
            if (cur.shape != 32u) {
                // ---- a SHORT unit: the last 64 * shape candidates of a successor-byte group, 32 / shape guides per pass ----
                // Lane l takes candidates [l * shape, (l + 1) * shape) of the window, i.e. field l % per of lane group
                // first + l / per: the same 16-byte loads, then one byte permute per plane spreads the field over the
                // whole register.
                const uint32_t shape = cur.shape, per = 32u / shape, w_log = shape == 16u ? 4u : 3u;
                const uint32_t glane = cur.group_abs + lane / per;
                uint32_t tile = glane >> 6;
                const uint32_t grp = glane & 63u, sub = lane & (per - 1u);
                if (tile >= n_tiles) tile = n_tiles - 1u;
                compared += static_cast<unsigned long long>(cur.last_cands) * (g_end - g_begin);
                const uint4 *__restrict__ src =
                    reinterpret_cast<const uint4 *>(scan_stream + static_cast<uint64_t>(tile) * kTileCands) + grp;
                const uint32_t sel = shape == 16u ? (sub ? 0x03020302u : 0x01000100u) : sub * 0x01010101u;
                uint32_t c[24];
                {
                    const uint4 a0 = src[q0 * 64u], a1 = src[q1 * 64u], a2 = src[q2 * 64u];
                    const uint4 b0 = src[(4u + q0) * 64u], b1 = src[(4u + q1) * 64u], b2 = src[(4u + q2) * 64u];
                    const uint4 t6[6] = {a0, a1, a2, b0, b1, b2};
#pragma unroll
                    for (int q = 0; q < 6; ++q) {
                        c[4 * q + 0] = __builtin_amdgcn_perm(0u, t6[q].x, sel); c[4 * q + 1] = __builtin_amdgcn_perm(0u, t6[q].y, sel);
                        c[4 * q + 2] = __builtin_amdgcn_perm(0u, t6[q].z, sel); c[4 * q + 3] = __builtin_amdgcn_perm(0u, t6[q].w, sel);
                    }
                }
                // the lane's candidates that are the item's: window offsets [lo, hi), the same for every guide field
                const int lo = static_cast<int>(cur.window & 0xFFFFu), hi = static_cast<int>(cur.window >> 16);
                const int below = lo - static_cast<int>(lane * shape), upto = hi - static_cast<int>(lane * shape);
                const uint32_t field = shape == 16u ? 0xFFFFu : 0xFFu;
                const uint32_t mine = (below <= 0 ? field : below >= static_cast<int>(shape) ? 0u : (field << below) & field) &
                                      (upto >= static_cast<int>(shape) ? field : upto <= 0 ? 0u : field >> (shape - upto));
                const uint32_t keep = mine * (shape == 16u ? 0x00010001u : 0x01010101u);
                const uint32_t off0 = grp * 32u + sub * shape;
                for (uint32_t gb = g_begin; gb < g_end; gb += 8u * per) { // 8 passes' masks at a time
                    __builtin_amdgcn_wave_barrier();
                    short_unit_masks(gword_stream, gb, shape, lane, wave_masks);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const uint32_t passes = (g_end - gb + per - 1u) / per;
                    for (uint32_t i = 0; i < (passes < 8u ? passes : 8u); ++i) {
                        PrevSlice prev;
                        const uint32_t ok = near_plane12_masks<THR>(c, wave_masks + i * 8u, thr, keep, prev);
                        if (__ballot(ok != 0u) != 0ull) {
                            note_candidates(fine_dup(ok, prev, dup_filter), gb + i * per, w_log, tile, off0, lane, w, raw, raw_used, max_chunks, counters);
                            own_chunk = true;
                        }
                    }
                }
                continue;
            }
            // ---- a full unit: 2048 consecutive candidates of the bucket from lane group cur.group_abs on; lane l takes the
            // 32 of group (first + l).  A window may start on any lane group and then straddles two tiles -- the same
            // 16-byte loads per lane, from two places.  (A window at the very end of the stream would reach past it:
            // those lanes read the last tile instead, and `keep` hides them.)
            const uint32_t glane = cur.group_abs + lane;
            uint32_t tile = glane >> 6;
            const uint32_t grp = glane & 63u;
            if (tile >= n_tiles) tile = n_tiles - 1u;
            compared += static_cast<unsigned long long>(cur.last_cands) * (g_end - g_begin);
            const uint4 *__restrict__ src =
                reinterpret_cast<const uint4 *>(scan_stream + static_cast<uint64_t>(tile) * kTileCands) + grp;
            uint32_t c[24];
            {
                const uint4 a0 = src[q0 * 64u], a1 = src[q1 * 64u], a2 = src[q2 * 64u];
                const uint4 b0 = src[(4u + q0) * 64u], b1 = src[(4u + q1) * 64u], b2 = src[(4u + q2) * 64u];
                c[0] = a0.x; c[1] = a0.y; c[2] = a0.z; c[3] = a0.w; c[4] = a1.x; c[5] = a1.y; c[6] = a1.z; c[7] = a1.w;
                c[8] = a2.x; c[9] = a2.y; c[10] = a2.z; c[11] = a2.w;
                c[12] = b0.x; c[13] = b0.y; c[14] = b0.z; c[15] = b0.w; c[16] = b1.x; c[17] = b1.y; c[18] = b1.z; c[19] = b1.w;
                c[20] = b2.x; c[21] = b2.y; c[22] = b2.z; c[23] = b2.w;
            }
            // the lane's candidates that are the item's: offsets [lo, hi) of the unit
            const int lo = static_cast<int>(cur.window & 0xFFFFu), hi = static_cast<int>(cur.window >> 16);
            const int below = lo - static_cast<int>(lane * 32u), upto = hi - static_cast<int>(lane * 32u);
            const uint32_t keep = (below <= 0 ? ~0u : below >= 32 ? 0u : ~0u << below) &
                                  (upto >= 32 ? ~0u : upto <= 0 ? 0u : ~0u >> (32 - upto));
            if (stamps) { // diagnostics: how long the planes take to arrive once they are requested
                const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                plane_wait += __builtin_amdgcn_s_memrealtime() - t1;
            }
            if constexpr (THR < 0) {
                // The runtime-threshold build (max_dist 5, the scan_generic knob): one loop, every guide's budget from the class
                // bits of its word -- thr minus the 0, 1 or 2 mismatches it has in the successor slice.
                for (uint32_t g = g_begin; g < g_end; g += kGuideGroup) {
                    const GuideGroup gg = *reinterpret_cast<const GuideGroup *>(gword_stream + g);
#pragma unroll
                    for (uint32_t uu = 0; uu < kGuideGroup; ++uu) {
                        if (g + uu >= g_end) break;
                        const uint32_t cls = (gg.w[uu] >> 24) & 3u;
                        if (cls > thr) continue;
                        PrevSlice prev;
                        const uint32_t ok = near_plane12<-1>(c, gg.w[uu], thr - cls, keep, prev);
                        if (__ballot(ok != 0u) != 0ull) {
                            note_candidates(fine_dup(ok, prev, dup_filter), g + uu, 5u, tile, grp * 32u, lane, w, raw, raw_used, max_chunks, counters);
                            own_chunk = true;
                        }
                    }
                }
                continue;
            }
            // class 1 (one mismatch in the successor slice): budget THR - 1 over the 12 positions; then class 0: budget THR
            const uint32_t gmid = cur.gmid < g_begin ? g_begin : cur.gmid > g_end ? g_end : cur.gmid;
            if constexpr (THR >= 1) {
                {
                    for (uint32_t g = g_begin; g < gmid; g += kGuideGroup) {
                        const GuideGroupAny gg = *reinterpret_cast<const GuideGroupAny *>(gword_stream + g);
#pragma unroll
                        for (uint32_t uu = 0; uu < kGuideGroup; ++uu) {
                            if (g + uu >= gmid) break;
                            PrevSlice prev;
                            const uint32_t ok = near_plane12<(THR < 1 ? 0 : THR - 1)>(c, gg.w[uu], thr - 1u, keep, prev);
                            if (__ballot(ok != 0u) != 0ull) {
                                note_candidates(fine_dup(ok, prev, dup_filter), g + uu, 5u, tile, grp * 32u, lane, w, raw, raw_used, max_chunks, counters);
                                own_chunk = true;
                            }
                        }
                    }
                }
            }
            for (uint32_t g = gmid; g < g_end; g += kGuideGroup) {
                const GuideGroupAny gg = *reinterpret_cast<const GuideGroupAny *>(gword_stream + g);
#pragma unroll
                for (uint32_t uu = 0; uu < kGuideGroup; ++uu) {
                    if (g + uu >= g_end) break;
                    PrevSlice prev;
                    const uint32_t ok = near_plane12<THR>(c, gg.w[uu], thr, keep, prev);
                    if (__ballot(ok != 0u) != 0ull) {
                        note_candidates(fine_dup(ok, prev, dup_filter), g + uu, 5u, tile, grp * 32u, lane, w, raw, raw_used, max_chunks, counters);
                        own_chunk = true;
                    }
                }
            }
            continue;
        }

        // ---- one tile: 2048 candidates, tile k of the item, guide slots [g_begin, g_end) -----------
        // The unit: 2048 consecutive candidates of the bucket from lane group cur.group_abs + 64 k on; lane l takes the
        // 32 of group (first + l).  Bucket-level items start on a tile; a window of the pruned scan may start on any lane
        // group and then straddles two tiles -- the same eight 16-byte loads per lane, from two places.  (A window at the
        // very end of the stream would reach past it: those lanes read the last tile instead, and `keep` hides them.)
        const uint32_t glane = cur.group_abs + (k << 6) + lane;
        const uint32_t tile = __builtin_amdgcn_readfirstlane(glane >> 6), grp = lane; // (whole tiles: uniform)
        // comparisons made here: the unit's real candidates (only a bucket's last one is short) x the real guides
        compared += static_cast<unsigned long long>((k + 1u == cur.n_tiles) ? cur.last_cands : kTileCands) * (g_end - g_begin);
        const uint4 *__restrict__ src =
            reinterpret_cast<const uint4 *>(scan_stream + static_cast<uint64_t>(tile) * kTileCands) + grp;
        uint32_t c[kPlanes];
#pragma unroll
        for (int q = 0; q < kPlanes / 4; ++q) {
            const uint4 t4 = src[q * 64];
            c[4 * q + 0] = t4.x; c[4 * q + 1] = t4.y; c[4 * q + 2] = t4.z; c[4 * q + 3] = t4.w;
        }
        // the lane's candidates that are the item's: offsets [lo, hi) of the tile
        const int lo = (k == 0u) ? static_cast<int>(cur.window & 0xFFFFu) : 0;
        const int hi = (k + 1u == cur.n_tiles) ? static_cast<int>(cur.window >> 16) : static_cast<int>(kTileCands);
        const int below = lo - static_cast<int>(lane * 32u), upto = hi - static_cast<int>(lane * 32u);
        const uint32_t keep = (below <= 0 ? ~0u : below >= 32 ? 0u : ~0u << below) &
                              (upto >= 32 ? ~0u : upto <= 0 ? 0u : ~0u >> (32 - upto));
        if (stamps) { // diagnostics: how long the planes take to arrive once they are requested
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            plane_wait += __builtin_amdgcn_s_memrealtime() - t1;
        }
        // Guide slots are padded to groups of 8 with a word (all T); the padding slots behind the last guide are skipped
        // below, one that does come near a real candidate would be dropped by k_verify.
        for (uint32_t g = g_begin; g < g_end; g += kGuideGroup) {
            const GuideGroup gg = *reinterpret_cast<const GuideGroup *>(gword_stream + g);
#pragma unroll
            for (uint32_t uu = 0; uu < kGuideGroup; ++uu) {
                if (g + uu >= g_end) break; // padding slots of the bucket's last group (scalar test, not taken: free)
                const uint32_t ok = near_plane<THR>(c, gg.w[uu], thr, keep);
                if (__ballot(ok != 0u) != 0ull) { // ~4 % of the (guide, tile) pairs on random data
                    note_candidates(ok, g + uu, 5u, tile, grp * 32u, lane, w, raw, raw_used, max_chunks, counters);
                    own_chunk = true;
                }
            }
        }
    }
    if (own_chunk) {
        raw_retire(w, lane, raw, raw_used);
        if (no_own_chunk && lane == 0) counters->raw_overflow = 1u;
    }
    // comparisons of the workgroup: summed in LDS, stored (not added: no reset needed) by its last wave
    if (lane == 0) {
        atomicAdd(&wg_compared, compared);
        if (atomicAdd(&waves_done, 1u) == (blockDim.x >> 6) - 1u) {
            scan_count[blockIdx.x] = atomicAdd(&wg_compared, 0ull);
            atomicMax(span + 1, static_cast<unsigned long long>(__builtin_amdgcn_s_memrealtime()));
        }
    }
    if (stamps && lane == 0) {
        stamps[4 * wave_id] = t_start;
        stamps[4 * wave_id + 1] = __builtin_amdgcn_s_memrealtime();
        stamps[4 * wave_id + 2] = (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11))) << 32) |
                                  __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); // XCC_ID, HW_ID
        stamps[4 * wave_id + 3] = units_done | (plane_wait << 32);
    }
}

template <int THR>
__global__ __launch_bounds__(1024, 8) void k_scan(const uint32_t *__restrict__ scan_stream,
                                                  const ScanItem *__restrict__ items_full,
                                                  const ScanItem *__restrict__ items_fine,
                                                  const PlanInfo *__restrict__ plan,
                                                  const RangeStart *__restrict__ range_start,
                                                  const uint32_t *__restrict__ gword_full,
                                                  const uint32_t *__restrict__ gword_fine, uint64_t *raw,
                                                  uint32_t *raw_used, uint32_t max_chunks, Counters *counters, uint32_t thr,
                                                  unsigned long long *stamps, uint64_t *__restrict__ scan_count,
                                                  unsigned long long *span, uint32_t n_tiles, uint32_t slice_bits)
{
    __shared__ uint32_t next_unit;
    __shared__ uint32_t waves_done;
    __shared__ unsigned long long wg_compared;
    __shared__ uint4 tail_masks[16][64]; // per wave: the guide masks of 8 passes of a short unit (short_unit_masks)
    // stamps (diagnostics, normally null): per wave {start, end} in 100 MHz ticks, {XCC_ID, HW_ID} and the number of
    // tiles it took; nothing else reads them
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { next_unit = 0; waves_done = 0; wg_compared = 0ull; }
    __syncthreads();
    if (blockIdx.x >= plan->n_ranges) {
        if (threadIdx.x == 0) scan_count[blockIdx.x] = 0ull;
        return;
    }
    if (threadIdx.x == 0) atomicMin(span, t_start); // the launch's own span: first workgroup in, last one out
    // the plan of this batch: bucket-level items, or the successor-byte groups of the pruned scan (k_fine_plan)
    if (plan->fine != 0u)
        scan_range<THR, true>(scan_stream, items_fine, plan, range_start, gword_fine, tail_masks[threadIdx.x >> 6], raw, raw_used, max_chunks, counters, thr, stamps,
                              scan_count, &next_unit, &waves_done, &wg_compared, t_start, span, n_tiles, slice_bits);
    else
        scan_range<THR, false>(scan_stream, items_full, plan, range_start, gword_full, tail_masks[threadIdx.x >> 6], raw, raw_used, max_chunks, counters, thr, stamps,
                               scan_count, &next_unit, &waves_done, &wg_compared, t_start, span, n_tiles, slice_bits);
}

// precalculatedScores[mask] with operator[] semantics: a missing mask contributes 0.0 (:394).
// Reference-built tables hold masks with flags on even bits below bit 40 only; for those the image carries a
// dense 2^20-entry table indexed by the 20 flags (one load instead of a 13-step search).
__device__ inline double mit_lookup(const ImageView &v, uint64_t mask)
{
    if (v.mit_dense) {
        if (mask >> 40) return 0.0;
        const uint32_t idx = gather_even16(static_cast<uint32_t>(mask)) |
                             (gather_even16(static_cast<uint32_t>(mask >> 32)) << 16);
        return v.mit_dense[idx];
    }
    uint32_t lo = 0, hi = v.n_scores;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint64_t m = v.score_mask[mid];
        if (m == mask) return v.score_val[mid];
        if (m < mask) lo = mid + 1; else hi = mid;
    }
    return 0.0;
}

// MIT and CFD terms of one scored off-target (isslScoreOfftargets.cpp:392-460) from the two signatures and the
// occurrence count.  The CFD product multiplies the penalties of the mismatching positions in position order, as the
// reference's loop over all 20 positions does (:399-460); the walk over the set flags visits the same positions.
__device__ inline void score_terms(const ImageView &v, uint64_t gsig, uint64_t ot, uint32_t occ, bool calc_mit, bool calc_cfd,
                                   double &mit_term, double &cfd_term, int &dist_out)
{
    mit_term = 0.0;
    cfd_term = 0.0;
    const uint64_t mm = mismatch_mask(gsig, ot);
    const int dist = __builtin_popcountll(mm);
    dist_out = dist;
    if (calc_mit && dist > 0) mit_term = mit_lookup(v, mm) * static_cast<double>(occ); // :394
    if (calc_cfd) {                                                                    // :399-460
        double cfd;
        if (dist == 0) {
            cfd = 1.0;
        } else {
            cfd = issl_cfd_pam[10];
            for (uint64_t left = mm; left != 0ull; left &= left - 1ull) { // the mismatching positions, ascending (<= max_dist)
                const uint32_t q = static_cast<uint32_t>(__builtin_ctzll(left)) >> 1;
                const uint32_t gb = static_cast<uint32_t>(gsig >> (2 * q)) & 3u;
                const uint32_t ob = static_cast<uint32_t>(ot >> (2 * q)) & 3u;
                cfd *= issl_cfd_pos[(q << 4) | (gb << 2) | (ob ^ 3u)];
            }
        }
        cfd_term = cfd * static_cast<double>(occ);
    }
}

// Workgroups of the two passes over the raw chunks (one chunk per workgroup and step): enough of them that the
// ~16 k first chunks of the scan waves are all in flight at once -- the passes are chains of dependent loads.
constexpr uint32_t kTailGrid = 16384;
constexpr uint32_t kReplayLds = 512;  // guides with up to this many hits: one wave each (k_replay)
static_assert(kReplayLds <= 512, "k_replay sorts (key, 9-bit index) pairs");
constexpr uint32_t kMidHits = 2048;   // ... up to this many: one 256-thread workgroup each (k_replay_mid), terms from k_verify;
                                      // beyond: k_replay_big (1024 threads, slice by slice, terms worked out as it walks)

// Exact check of the raw records, IN PLACE: one thread per record, one chunk per 128-thread workgroup.
// A record that survives becomes a hit: key guide<<37 | slice<<32 | site id (list-order layouts: position in the
// bucket's list), rank inside its guide from the per-guide counter, MIT / CFD terms.  With hit slots the first
// ws.slot_hits hits of a guide are written to its slots; what lies beyond overwrites the record with its key for the
// grouping pass; every other slot of the chunk becomes kDeadKey.
__global__ __launch_bounds__(kChunkRecs, 8) void k_verify(ImageView v, Workspace ws, const uint64_t *__restrict__ guides,
                                                       ScoreParams p)
{
    short_kernel_priority();
    const int max_dist = p.max_dist;
    const bool calc_mit = p.method == ISSL_METHOD_MIT || p.method == ISSL_METHOD_AND || p.method == ISSL_METHOD_OR ||
                          p.method == ISSL_METHOD_AVG;
    const bool calc_cfd = p.method == ISSL_METHOD_CFD || p.method == ISSL_METHOD_AND || p.method == ISSL_METHOD_OR ||
                          p.method == ISSL_METHOD_AVG;
    const uint32_t prune_mode = ws.plan->fine; // what the scan of this batch worked through
    uint32_t n_chunks = ws.counters->raw_chunks;
    if (blockIdx.x == 0 && threadIdx.x == 0) { // what the host needs to know after any number of batches
        if (ws.counters->raw_overflow) atomicOr(&ws.sticky[0], 1u);
        atomicMax(&ws.sticky[1], n_chunks);
        if (ws.plan->error) atomicOr(&ws.sticky[2], ws.plan->error);
    }
    if (n_chunks > ws.cap_chunks) n_chunks = static_cast<uint32_t>(ws.cap_chunks);
    const uint64_t low = (1ull << v.slice_width) - 1ull;
    for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        // The passes behind the scan are chains of dependent loads at full occupancy: what they cost is the number of
        // links.  Header and record of the chunk are asked for together (a chunk always has its 128 slots), then the
        // stream record and the guide slot's 16 bytes together, then -- nothing more before the exact test.
        uint64_t *recs = ws.raw + static_cast<uint64_t>(chunk) * kChunkRecs;
        const uint32_t t = threadIdx.x + 1u;
        const uint64_t rec_any = recs[t < kChunkRecs ? t : 0u];
        const uint32_t used = ws.raw_used[chunk];
        const bool in_use = t < used && t < kChunkRecs; // every lane stays: the counting below is done by the wave
        const uint64_t rec = in_use ? rec_any : 0ull;
        uint64_t key = kDeadKey;
        double mit_term = 0.0, cfd_term = 0.0; // of a record that survives: computed here, one thread per hit, so that the
                                               // replay (one wave per guide, a chain of dependent steps) only adds them up
        uint64_t hit_gsig = 0, hit_ot = 0;     // ... from these
        uint32_t hit_occ = 0;
        const uint32_t offset = static_cast<uint32_t>(rec) & (kTileCands - 1u);
        const uint32_t tile = static_cast<uint32_t>(rec >> 11) & 0x3FFFFFFu;
        const uint32_t gslot = static_cast<uint32_t>(rec >> 37);
        // sorted layouts: what the stream holds at the record's place, asked for before anything else is known about it
        const bool by_id = v.srec || v.sid; // the scoring order is (slice, site id): ImageHeader
        StreamRec sr_early{};
        if (in_use && v.srec) sr_early = v.srec[static_cast<uint64_t>(tile) * kTileCands + offset];
        else if (in_use && v.sid) sr_early.id = v.sid[static_cast<uint64_t>(tile) * kTileCands + offset];
        // the guide slot knows its guide and its bucket (pruned scan: and its successor-byte group, and the guide's
        // signature): no search for the tile's
        uint32_t guide = kNoGuide, where = 0;
        uint64_t gsig = 0;
        if (in_use) {
            if (prune_mode) { const FineMeta m = ws.fmeta[gslot]; guide = m.guide; where = m.where; gsig = m.gsig; }
            else { guide = ws.gidx[gslot]; where = ws.gbucket[gslot]; }
        }
        if (guide != kNoGuide) {
            const uint32_t bucket = prune_mode ? where >> 8 : where;
            const uint32_t slice = bucket >> v.slice_width;
            // Is the candidate the item's?  The scan notes only candidates of the item's own window (`keep`: not the zero
            // padding behind a bucket, not the neighbouring group that shares the tile), so on the sorted layouts, which need
            // nothing else from the bucket tables, the question is not asked again.  The list-order layouts find their list
            // entry through the bucket's start and check on the way.
            uint64_t start = 0, pos = 0;
            bool mine = true;
            if (!by_id) {
                start = v.bucket_start[bucket];
                pos = static_cast<uint64_t>(tile - v.tile_first[bucket]) * kTileCands + offset; // in the stream
                mine = pos < v.bucket_start[bucket + 1] - start;
            }
            if (mine) {
                if (!prune_mode) gsig = guides[guide];
                // sorted layouts: signature, site id (and a 24-bit copy of the count) come in one stream-order record, or --
                // compact -- the id alone, with the signature behind it in the site table
                StreamRec sr = sr_early;
                if (v.sid) sr.sig = v.sites[sr.id]; // (the site table of a sorted layout: signature | 24-bit count << 40, like a stream record)
                const uint64_t ot = by_id    ? sr.sig & kSigMask
                                    : v.esig ? v.esig[start + pos]
                                    : v.occ8 ? candidate_signature(v, bucket, tile, offset) // cold sections in host memory
                                             : v.sites[v.entries[start + pos] & 0xFFFFFFFFull];
                if (__builtin_popcountll(mismatch_mask(gsig, ot)) <= max_dist) { // exact, full signatures (:376-382)
                    // First-matching-slice rule (equivalent of the seen bitmap, isslScoreOfftargets.cpp:385-390,463):
                    // the site was already met iff an earlier slice of the XOR is all zero.
                    const uint64_t x = gsig ^ ot;
                    if (!prune_mode) {
                        bool earlier = false;
                        for (uint32_t j = 0; j < slice; ++j)
                            if (((x >> (v.slice_width * j)) & low) == 0) earlier = true;
                        if (!earlier) {
                            // list-order layouts: the key carries the position in the bucket's list, which is the stream
                            // position; sorted layouts: the site id (lists ascend by id, so the order is the same)
                            const uint64_t lp = by_id ? sr.id : pos;
                            key = (static_cast<uint64_t>(guide) << kKeyGuideShift) | (static_cast<uint64_t>(slice) << kKeySliceShift) | lp;
                        }
                    } else {
                        // Pruned scan: the guide meets this site once in every exactly matching slice whose successor
                        // slice has at most `tol` mismatches (k_fine_count); the smallest such slice reports it, under
                        // the slice the reference would meet it in first.
                        const uint32_t tol = prune_mode - 1u; // 0, 1, 2 mismatches allowed in the successor slice
                        const uint64_t mm = mismatch_mask(gsig, ot);
                        uint32_t first = slice, reporter = slice;
                        for (uint32_t j = slice; j-- > 0;) {
                            if (((x >> (v.slice_width * j)) & low) != 0) continue;
                            first = j;
                            if (static_cast<uint32_t>(__builtin_popcount(succ_byte(mm, j, v.slice_width))) <= tol) reporter = j; // (the successor unit's flags)
                        }
                        if (reporter == slice)
                            key = (static_cast<uint64_t>(guide) << kKeyGuideShift) | (static_cast<uint64_t>(first) << kKeySliceShift) | sr.id;
                    }
                    if (key != kDeadKey) { // the hit will be scored: what its terms are made of (:348)
                        uint32_t occ;
                        if (by_id) {
                            occ = static_cast<uint32_t>(sr.sig >> 40);
                            if (occ == kOccSaturated) occ = v.site_occ[sr.id];
                        } else if (v.occ8) {
                            occ = v.occ8[start + pos];
                            if (occ == 255u) occ = static_cast<uint32_t>(v.entries[start + pos] >> 32); // (host memory)
                        } else {
                            occ = static_cast<uint32_t>(v.entries[start + pos] >> 32);
                        }
                        hit_gsig = gsig; hit_ot = ot; hit_occ = occ;
                    }
                }
            }
        }
        // Count the hit for its guide.  The count doubles as the hit's place in the guide's segment, so that the grouping
        // pass scatters without a second atomic (ws.rank, by raw-record slot).  The hits of a guide in one unit lie side by
        // side in the chunk (the scan wave notes them guide by guide): every RUN of neighbouring lanes with the same guide
        // takes one atomic, issued by its first lane -- no loop, every run of the wave in the same instruction.  (The
        // atomics are half of this kernel's time on a skewed index: profiles/r03_ablation_verify.log.)
        const bool live = key != kDeadKey;
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t prev_guide = static_cast<uint32_t>(__shfl_up(static_cast<int>(live ? guide : kNoGuide), 1, 64));
        const bool continues = live && lane != 0u && prev_guide == guide;    // (a dead lane never equals a live one: kNoGuide)
        const uint64_t starts = __ballot(!continues);                        // first lanes of runs, and the dead lanes
        uint32_t rank = 0;
        if (starts == ~0ull) { // (uniform over the wave) no runs, the usual case on an even index: everybody for itself
            if (live) rank = atomicAdd(&ws.gcount[guide], 1u);
        } else {
            const uint64_t upto = (lane == 63u ? 0ull : (~0ull << (lane + 1u))); // the lanes above this one
            const uint32_t head = 63u - static_cast<uint32_t>(__builtin_clzll(starts & ~upto)); // lane 0 always starts: never empty
            const uint64_t later = starts & upto;
            const uint32_t next = later ? static_cast<uint32_t>(__builtin_ctzll(later)) : 64u;
            uint32_t base = 0;
            if (live && !continues) base = atomicAdd(&ws.gcount[guide], next - lane); // the run is [lane, next)
            rank = static_cast<uint32_t>(__shfl(static_cast<int>(base), static_cast<int>(head), 64)) + (lane - head);
        }
        if (live && rank == kReplayLds) atomicAdd(&ws.counters->overflowed, 1u); // the guide's first hit beyond what k_replay takes
        if (live && rank < ws.slot_hits) {
            // hit slots (Workspace): the hit goes to its final place at once and takes no part in the grouping pass
            int dist;
            score_terms(v, hit_gsig, hit_ot, hit_occ, calc_mit, calc_cfd, mit_term, cfd_term, dist);
            const uint64_t at = static_cast<uint64_t>(guide) * ws.slot_hits + rank;
            SlotRec r;
            r.mit = mit_term; r.cfd = cfd_term; r.key = key; r.pad = 0;
            ws.slots[at] = r;
            key = kDeadKey;
        } else if (live) {
            const uint64_t slot = static_cast<uint64_t>(chunk) * (kChunkRecs - 1u) + (t - 1u); // < cap_chunks * 127 <= cap_hits
            ws.rank[slot] = rank;
            // The terms (:392-460) -- unless the guide already has more hits than the replays that read them take
            // (k_replay, k_replay_mid): the many-hit replay works out the terms of the hits it walks by itself, and on
            // skewed data most hits belong to such guides and lie behind their early exit.
            if (rank < kMidHits) {
                int dist;
                score_terms(v, hit_gsig, hit_ot, hit_occ, calc_mit, calc_cfd, mit_term, cfd_term, dist);
                reinterpret_cast<double2 *>(ws.pay)[slot] = make_double2(mit_term, cfd_term);
            }
        }
        if (!in_use) continue;
        // (a lean batch has no grouping pass to read the keys back -- every hit went to its guide's slots, or the batch is run
        // again in full: 8 bytes per record that need not be written)
        if (!ws.lean_tail) recs[t] = key;
    }
}

template <int THR>
static void launch_scan_thr(const ImageView &v, const Workspace &ws, const Tuning &tn, uint32_t thr, uint32_t prune_mode,
                            hipStream_t stream)
{
    // pruned scan: the items and guide words grouped by (bucket, successor byte); the plan says which list counts
    hipLaunchKernelGGL(k_scan<THR>, dim3(tn.scan_blocks), dim3(tn.scan_threads), 0, stream, v.scan, ws.items,
                       prune_mode ? ws.fitems : ws.items, ws.plan, ws.range_start, ws.gword, prune_mode ? ws.fword : ws.gword,
                       ws.raw, ws.raw_used, static_cast<uint32_t>(ws.cap_chunks), ws.counters, thr, ws.stamps, ws.scan_count,
                       ws.scan_span + 2u * ws.span_slot, v.n_tiles, v.slice_width);
}

void launch_scan(const ImageView &v, const Workspace &ws, const Tuning &tn, const uint64_t *d_guides, uint32_t n,
                 int max_dist, uint32_t prune_mode, void *stream_)
{
    (void)n;
    (void)d_guides;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (max_dist < 0) { // isslScoreOfftargets.cpp:382: no distance satisfies 0 <= dist <= maxDist -- nothing is compared
        (void)hipMemsetAsync(ws.scan_count, 0, 8ull * tn.scan_blocks, stream);
        return;
    }
    const uint32_t thr = max_dist > 31 ? 31u : static_cast<uint32_t>(max_dist);
    // the runtime-threshold build serves max_dist > 4 (and, forced by the scan_generic knob, the tests of that build)
    if (tn.scan_generic || thr > 4) launch_scan_thr<-1>(v, ws, tn, thr, prune_mode, stream);
    else if (thr == 0) launch_scan_thr<0>(v, ws, tn, thr, prune_mode, stream);
    else if (thr == 1) launch_scan_thr<1>(v, ws, tn, thr, prune_mode, stream);
    else if (thr == 2) launch_scan_thr<2>(v, ws, tn, thr, prune_mode, stream);
    else if (thr == 3) launch_scan_thr<3>(v, ws, tn, thr, prune_mode, stream);
    else launch_scan_thr<4>(v, ws, tn, thr, prune_mode, stream);
}

void launch_verify(const ImageView &v, const Workspace &ws, const uint64_t *d_guides, uint32_t n, const ScoreParams &p, void *stream)
{
    if (p.max_dist < 0) return;
    // (the kernel strides over the chunks the scan used: a small batch's few thousand need no 16 384 workgroups to start and leave)
    const uint32_t grid = n < 256u ? std::max<uint32_t>(2048u, 64u * n) : kTailGrid;
    hipLaunchKernelGGL(k_verify, dim3(std::min(grid, kTailGrid)), dim3(kChunkRecs), 0, static_cast<hipStream_t>(stream), v, ws, d_guides, p);
}

// ------------------------------------------------------------------------------------------------
// hit grouping: counting sort of the keys by guide
// ------------------------------------------------------------------------------------------------

constexpr uint32_t kScanChunk = 2048; // elements per block in the device-wide prefix sum
constexpr uint32_t kBigLds = 7680;    // hits per slice k_replay_big sorts in LDS (2 x 30 KiB); longer slices are sorted in HBM

// What a guide's hits take in the grouped arrays: nothing when they all sit in its hit slots (Workspace::slot_hits).
// (with slots of any width every guide of the many-hit replays keeps a segment for ALL its hits: k_replay_big copies the keys in
// the slots in front of the rest, and k_replay_mid may hand a guide on to it)
__device__ __forceinline__ uint32_t grouped_hits(uint32_t count, uint32_t slot_hits)
{
    return count <= (slot_hits ? kSlotHits : 0u) ? 0u : count;
}

__global__ __launch_bounds__(256) void k_prefix_block_sums(const uint32_t *__restrict__ in, uint32_t n,
                                                           uint32_t *__restrict__ sums, uint32_t slot_hits)
{
    short_kernel_priority();
    __shared__ uint64_t lds[256];
    const uint32_t base = blockIdx.x * kScanChunk + threadIdx.x * 8u;
    uint64_t s = 0;
    for (uint32_t i = 0; i < 8; ++i)
        if (base + i < n) s += grouped_hits(in[base + i], slot_hits);
    uint64_t total;
    (void)block_exclusive_scan(s, lds, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = static_cast<uint32_t>(total);
}

__global__ __launch_bounds__(256) void k_prefix_of_sums(uint32_t *__restrict__ sums, uint32_t n_blocks)
{
    short_kernel_priority();
    __shared__ uint64_t lds[256];
    uint64_t carry = 0;
    for (uint32_t base = 0; base < n_blocks; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t val = i < n_blocks ? sums[i] : 0;
        uint64_t total;
        const uint64_t ex = block_exclusive_scan(val, lds, &total);
        if (i < n_blocks) sums[i] = static_cast<uint32_t>(carry + ex);
        carry += total;
    }
}

__global__ __launch_bounds__(256) void k_prefix_apply(const uint32_t *__restrict__ in, uint32_t n,
                                                      const uint32_t *__restrict__ sums, uint32_t *__restrict__ out,
                                                      uint32_t *__restrict__ big, Counters *__restrict__ counters,
                                                      uint32_t slot_hits)
{
    short_kernel_priority();
    __shared__ uint64_t lds[256];
    const uint32_t base = blockIdx.x * kScanChunk + threadIdx.x * 8u;
    uint32_t val[8];
    uint64_t s = 0;
    uint32_t nb = 0;
    for (uint32_t i = 0; i < 8; ++i) {
        const uint32_t c = (base + i < n) ? in[base + i] : 0u;
        nb += c > kReplayLds;
        val[i] = grouped_hits(c, slot_hits);
        s += val[i];
    }
    if (nb != 0u) { // guides for k_replay_mid / k_replay_big: one reservation per thread
        uint32_t at = atomicAdd(&counters->n_big, nb);
        for (uint32_t i = 0; i < 8; ++i)
            if (base + i < n && in[base + i] > kReplayLds) big[at++] = base + i;
    }
    uint64_t run = block_exclusive_scan(s, lds, nullptr) + sums[blockIdx.x];
    for (uint32_t i = 0; i < 8; ++i) {
        if (base + i < n) out[base + i] = static_cast<uint32_t>(run);
        run += val[i];
    }
}

// Whole prefix sum in one workgroup (used while n is moderate; saves two launches): every thread sums its own run of
// consecutive counts (16-byte loads), ONE scan over the 1024 run totals, then every thread writes its run's prefixes --
// two barriers in all, where a loop over chunks of 4096 counts paid three per chunk (0.06 ms at 100 k guides).  The list
// of guides with more than kReplayLds hits comes out of the same scan (their number rides in a second scanned word), in
// guide order and without an atomic: on indexes where most guides are such (skewed genomes, the 3 G-line index) one
// returning atomic per guide from a single workgroup cost more than the rest of the grouping (2 ms per 100 k guides).
// With hit slots only those guides have anything in the grouped arrays (grouped_hits), and when k_verify saw no guide
// outgrow its slots there is nothing to do at all.
__global__ __launch_bounds__(1024) void k_prefix_single(const uint32_t *__restrict__ in, uint32_t n,
                                                        uint32_t *__restrict__ out, uint32_t *__restrict__ big,
                                                        Counters *__restrict__ counters, uint32_t slot_hits)
{
    short_kernel_priority();
    if (slot_hits >= kReplayLds && counters->overflowed == 0u) return; // (n_big stays 0: k_group_scatter and the replays of many-hit guides return at once)
    __shared__ uint32_t wave_sum[16], wave_big[16];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t per = ((n + 1023u) / 1024u + 3u) & ~3u; // counts per thread, a multiple of 4: the runs start 16-byte aligned
    const uint32_t i0 = threadIdx.x * per;
    uint32_t s = 0, b = 0;
    for (uint32_t k = 0; k < per; k += 4) {
        const uint32_t i = i0 + k;
        uint4 q = make_uint4(0, 0, 0, 0);
        if (i + 3 < n) q = *reinterpret_cast<const uint4 *>(in + i);
        else { if (i < n) q.x = in[i]; if (i + 1 < n) q.y = in[i + 1]; if (i + 2 < n) q.z = in[i + 2]; }
        s += grouped_hits(q.x, slot_hits) + grouped_hits(q.y, slot_hits) + grouped_hits(q.z, slot_hits) + grouped_hits(q.w, slot_hits);
        b += (q.x > kReplayLds) + (q.y > kReplayLds) + (q.z > kReplayLds) + (q.w > kReplayLds);
    }
    uint32_t x = s, xb = b; // inclusive scans of s and b inside the wave
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64), yb = __shfl_up(xb, d, 64);
        if (lane >= d) { x += y; xb += yb; }
    }
    if (lane == 63) { wave_sum[wave] = x; wave_big[wave] = xb; }
    __syncthreads();
    uint32_t run = x - s, brun = xb - b;
    for (uint32_t wv = 0; wv < wave; ++wv) { run += wave_sum[wv]; brun += wave_big[wv]; }
    if (threadIdx.x == 1023u) counters->n_big = brun + b;
    for (uint32_t k = 0; k < per; k += 4) {
        const uint32_t i = i0 + k;
        if (i >= n) break;
        uint4 q = make_uint4(0, 0, 0, 0);
        if (i + 3 < n) q = *reinterpret_cast<const uint4 *>(in + i);
        else { q.x = in[i]; if (i + 1 < n) q.y = in[i + 1]; if (i + 2 < n) q.z = in[i + 2]; }
        const uint32_t ex = grouped_hits(q.x, slot_hits), ey = grouped_hits(q.y, slot_hits), ez = grouped_hits(q.z, slot_hits),
                       ew = grouped_hits(q.w, slot_hits);
        const uint4 o = make_uint4(run, run + ex, run + ex + ey, run + ex + ey + ez);
        if (i + 3 < n) *reinterpret_cast<uint4 *>(out + i) = o;
        else { out[i] = o.x; if (i + 1 < n) out[i + 1] = o.y; if (i + 2 < n) out[i + 2] = o.z; }
        run += ex + ey + ez + ew;
        if (q.x > kReplayLds) big[brun++] = i; // guides for k_replay_mid / k_replay_big
        if (q.y > kReplayLds) big[brun++] = i + 1;
        if (q.z > kReplayLds) big[brun++] = i + 2;
        if (q.w > kReplayLds) big[brun++] = i + 3;
    }
}

// (One 32-byte record {key, terms, rank} per hit instead of the three arrays -- written by k_verify, moved by this
// kernel, read by the replay -- was measured in round 3: verify +7 %, this kernel +35 %: the passes are bound by the
// bytes they move, not by the number of streams; profiles/r03_ab_hit_records.log.)
__global__ __launch_bounds__(kChunkRecs) void k_group_scatter(const uint64_t *__restrict__ raw,
                                                              const uint32_t *__restrict__ raw_used,
                                                              const Counters *__restrict__ counters, uint32_t cap_chunks,
                                                              const uint32_t *__restrict__ gcount,
                                                              const uint32_t *__restrict__ goff,
                                                              const uint32_t *__restrict__ rank,
                                                              const double2 *__restrict__ pay,
                                                              uint64_t *__restrict__ sorted, double2 *__restrict__ terms,
                                                              uint32_t slot_hits)
{
    short_kernel_priority();
    // hit slots: only guides with more hits than fit their slots left anything to group -- on an even index none
    if (slot_hits >= kReplayLds && counters->n_big == 0u) return;
    uint32_t n_chunks = counters->raw_chunks;
    if (n_chunks > cap_chunks) n_chunks = cap_chunks;
    for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        // (everything that does not depend on the key is asked for at once: the pass is a chain of dependent loads)
        const uint64_t *recs = raw + static_cast<uint64_t>(chunk) * kChunkRecs;
        const uint32_t t = threadIdx.x + 1u;
        const uint64_t slot = static_cast<uint64_t>(chunk) * (kChunkRecs - 1u) + (threadIdx.x < kChunkRecs - 1u ? threadIdx.x : kChunkRecs - 2u);
        const uint64_t key = recs[t < kChunkRecs ? t : 0u];
        const uint32_t my_rank = rank[slot];
        const double2 my_pay = pay[slot];
        const uint32_t used = raw_used[chunk];
        if (t >= used || t >= kChunkRecs || key == kDeadKey) continue;
        const uint32_t guide = static_cast<uint32_t>(key >> kKeyGuideShift);
        const uint32_t to = goff[guide] + my_rank; // rank: k_verify's
        sorted[to] = key;
        if (gcount[guide] <= kMidHits) terms[to] = my_pay; // (the many-hit replay makes its own)
    }
}

void launch_group_hits(const Workspace &ws, uint32_t n, void *stream_)
{
    if (ws.lean_tail) return; // (predicted: no guide beyond its hit slots, nothing to group; k_replay checks)
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const uint32_t m = n + 1; // gcount[n] = 0 so that goff[n] = total
    if (m <= (1u << 18)) {
        hipLaunchKernelGGL(k_prefix_single, dim3(1), dim3(1024), 0, stream, ws.gcount, m, ws.goff, ws.gcur_big,
                           ws.counters, ws.slot_hits);
    } else {
        const uint32_t blocks = (m + kScanChunk - 1) / kScanChunk;
        hipLaunchKernelGGL(k_prefix_block_sums, dim3(blocks), dim3(256), 0, stream, ws.gcount, m, ws.blocksum, ws.slot_hits);
        hipLaunchKernelGGL(k_prefix_of_sums, dim3(1), dim3(256), 0, stream, ws.blocksum, blocks);
        hipLaunchKernelGGL(k_prefix_apply, dim3(blocks), dim3(256), 0, stream, ws.gcount, m, ws.blocksum, ws.goff,
                           ws.gcur_big, ws.counters, ws.slot_hits);
    }
    hipLaunchKernelGGL(k_group_scatter, dim3(kTailGrid), dim3(kChunkRecs), 0, stream, ws.raw, ws.raw_used, ws.counters,
                       static_cast<uint32_t>(ws.cap_chunks), ws.gcount, ws.goff, ws.rank, reinterpret_cast<const double2 *>(ws.pay),
                       ws.sorted, reinterpret_cast<double2 *>(ws.terms), ws.slot_hits);
}

// ------------------------------------------------------------------------------------------------
// replay: ordered MIT/CFD accumulation, one wave per guide
// ------------------------------------------------------------------------------------------------


// Ascending sort of data[0..n) by the whole workgroup.  Bitonic network with every comparator
// ascending; comparators that touch an index >= n are no-ops (virtual +inf padding).
__device__ inline void wave_sort(uint64_t *data, uint32_t n)
{
    if (n < 2) return;
    uint32_t np = 1;
    while (np < n) np <<= 1;
    for (uint32_t k = 2; k <= np; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < (np >> 1); t += blockDim.x) {
                const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)); // bit log2(j) of i is 0
                const uint32_t l = (j == (k >> 1)) ? (i ^ (k - 1u)) : (i | j);
                if (l < n) {
                    const uint64_t a = data[i], b = data[l];
                    if (a > b) { data[i] = b; data[l] = a; }
                }
            }
            __syncthreads();
        }
    }
}

__device__ inline double bcast_f64(double x, int lane)
{
    const uint64_t u = __double_as_longlong(x);
    const uint32_t lo = __builtin_amdgcn_readlane(static_cast<uint32_t>(u), lane);
    const uint32_t hi = __builtin_amdgcn_readlane(static_cast<uint32_t>(u >> 32), lane);
    return __longlong_as_double((static_cast<uint64_t>(hi) << 32) | lo);
}

// Adds the terms of a chunk of cnt <= 64 hits to the running totals in walking order (:394, :460) and applies the exit test
// of :467-496 after every hit.  `chunk_lds`: the chunk's terms in LDS, {mit, cfd} per hit in walking order, zeros behind the
// last hit up to a multiple of 8; lane l also holds the terms of hit l (0.0 beyond cnt).  The sums are a serial chain of
// f64 additions -- the order is part of the result -- read from LDS, every lane the same 16 bytes: one load and two
// additions per hit (passing them from lane to lane through scalar registers cost four readlanes more; the walk shares
// its SIMD with seven other waves, what it costs is instructions: replay 0.37 -> 0.30 ms, 4.2 -> 3.5 on the skewed index).
// The exit test is not part of the chain: every lane keeps the totals as they stood after ITS hit, the tests run side by
// side afterwards, and the first lane that passes decides where the walk stops.  (Testing inside the chain costs a
// compare, a branch and their latencies per hit: ~200 cycles against ~50.)
// Returns true when the walk stops; `kept` counts the hits that were scored, the totals are those at that point.
__device__ __forceinline__ bool accumulate_chunk(double mit_term, double cfd_term, uint32_t cnt, const ScoreParams &p,
                                                 uint32_t lane, double &tot_mit, double &tot_cfd, uint32_t &kept,
                                                 const double2 *chunk_lds)
{
    auto passes = [&](double m, double c) {
        if (p.method == ISSL_METHOD_AND) return m > p.maximum_sum && c > p.maximum_sum;
        if (p.method == ISSL_METHOD_OR) return m > p.maximum_sum || c > p.maximum_sum;
        if (p.method == ISSL_METHOD_AVG) return ((m + c) / 2.0) > p.maximum_sum;
        if (p.method == ISSL_METHOD_MIT) return m > p.maximum_sum;
        if (p.method == ISSL_METHOD_CFD) return c > p.maximum_sum;
        return false;
    };
    // First the totals behind the chunk alone (the same additions in the same order).  Terms are products of
    // non-negative table values and counts, so the totals only grow and every exit test is monotone in them: when the
    // totals behind the chunk do not pass, no hit inside it did, and the per-hit bookkeeping below is not needed.  (A
    // table with a negative entry, or a NaN, takes the careful pass.)
    {
        double tm = tot_mit, tc = tot_cfd;
        for (uint32_t l0 = 0; l0 < cnt; l0 += 8) {
#pragma unroll
            for (uint32_t l = 0; l < 8; ++l) { // x + 0.0 == x: the zeros behind the last hit change nothing
                const double2 t = chunk_lds[l0 + l];
                tm += t.x;
                tc += t.y;
            }
        }
        const bool grows = __ballot(lane < cnt && !(mit_term >= 0.0 && cfd_term >= 0.0)) == 0ull;
        if (grows && !passes(tm, tc)) {
            kept += cnt;
            tot_mit = tm;
            tot_cfd = tc;
            return false;
        }
    }
    double tm = tot_mit, tc = tot_cfd, mine_m = 0.0, mine_c = 0.0;
    for (uint32_t l0 = 0; l0 < cnt; l0 += 8) {
#pragma unroll
        for (uint32_t l = 0; l < 8; ++l) {
            const double2 t = chunk_lds[l0 + l];
            tm += t.x;
            tc += t.y;
            if (lane == l0 + l) { mine_m = tm; mine_c = tc; }
        }
    }
    const bool exit_here = passes(mine_m, mine_c);
    const uint64_t exits = __ballot(exit_here && lane < cnt);
    if (exits != 0ull) {
        const int first = __builtin_ctzll(exits);
        kept += static_cast<uint32_t>(first) + 1u;
        tot_mit = bcast_f64(mine_m, first);
        tot_cfd = bcast_f64(mine_c, first);
        return true;
    }
    kept += cnt;
    tot_mit = tm;
    tot_cfd = tc;
    return false;
}

// MIT and CFD terms of one scored off-target (isslScoreOfftargets.cpp:392-460) and its record.
struct HitTerms {
    double mit, cfd;
    issl_hit rec;
};

__device__ inline HitTerms hit_terms(const ImageView &v, uint64_t gsig, uint32_t g, uint64_t key, bool calc_mit,
                                     bool calc_cfd, bool want_id)
{
    HitTerms t;
    t.mit = 0.0;
    t.cfd = 0.0;
    const uint64_t low = (1ull << v.slice_width) - 1ull;
    const uint32_t slice = static_cast<uint32_t>(key >> kKeySliceShift) & kKeySliceMask;
    uint32_t pos = static_cast<uint32_t>(key);
    const uint32_t bucket = (slice << v.slice_width) + static_cast<uint32_t>((gsig >> (v.slice_width * slice)) & low);
    const uint64_t at = v.bucket_start[bucket] + pos;
    uint32_t id = 0, occ;
    uint64_t ot;
    if (v.srec || v.sid) {
        // sorted layouts: the key's low word is the site id; issl_dump_hits also wants the position in the bucket's list
        // (:344): the lists ascend by id, so a binary search finds it (in host memory when the lists live there)
        id = pos;
        const uint64_t site = v.sites[id]; // signature | min(count, kOccSaturated) << 40 (k_tag_sites)
        ot = site & kSigMask;
        occ = static_cast<uint32_t>(site >> 40);
        if (occ == kOccSaturated) occ = v.site_occ[id];
        pos = 0;
        if (want_id && v.entries) {
            const uint64_t *list = v.entries + v.bucket_start[bucket];
            uint64_t lo = 0, hi = v.bucket_start[bucket + 1] - v.bucket_start[bucket];
            while (lo < hi) {
                const uint64_t mid = (lo + hi) >> 1;
                if (static_cast<uint32_t>(list[mid]) < id) lo = mid + 1; else hi = mid;
            }
            pos = static_cast<uint32_t>(lo);
        } else if (want_id) {
            // An image without slice lists (ImageHeader::lists_absent): the position in the bucket's list is the number of
            // the bucket's sites with a smaller id.  The stream holds the bucket's ids, ascending inside each of its 256
            // successor-byte groups: one binary search per group.
            const uint32_t *ss = v.sub_start + static_cast<uint64_t>(bucket) * 257u;
            const uint64_t first = static_cast<uint64_t>(v.tile_first[bucket]) * kTileCands;
            for (uint32_t w = 0; w < 256u; ++w) {
                uint32_t lo = ss[w], hi = ss[w + 1];
                const uint32_t s0 = lo;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    const uint32_t there = v.srec ? v.srec[first + mid].id : v.sid[first + mid];
                    if (there < id) lo = mid + 1; else hi = mid;
                }
                pos += lo - s0;
            }
        }
    } else if (v.occ8) {
        // cold sections in host memory: signature from the scan planes, occurrences from the byte copy in HBM; the list
        // entry itself (PCIe) only for counts that do not fit a byte and for the site id of issl_dump_hits
        ot = candidate_signature(v, bucket, v.tile_first[bucket] + (pos >> 11), pos & (kTileCands - 1u));
        occ = v.occ8[at];
        if (occ == 255u || want_id) {
            const uint64_t e = v.entries[at];
            id = static_cast<uint32_t>(e);
            occ = static_cast<uint32_t>(e >> 32);
        }
    } else {
        const uint64_t e = v.entries[at];
        id = static_cast<uint32_t>(e);
        occ = static_cast<uint32_t>(e >> 32);
        ot = v.esig ? v.esig[at] : v.sites[id]; // independent of `e` when the in-list copy exists
    }
    int dist;
    score_terms(v, gsig, ot, occ, calc_mit, calc_cfd, t.mit, t.cfd, dist);
    t.rec.guide = g; t.rec.slice = slice; t.rec.pos = pos; t.rec.id = id;
    t.rec.dist = static_cast<uint32_t>(dist); t.rec.occ = occ;
    return t;
}

__global__ __launch_bounds__(64, 8) void k_replay(ImageView v, Workspace ws, const uint64_t *__restrict__ guides,
                                               uint32_t n, ScoreParams p, double *__restrict__ out_mit,
                                               double *__restrict__ out_cfd, uint32_t *__restrict__ out_kept,
                                               issl_hit *__restrict__ out_hits)
{
    short_kernel_priority();
    __shared__ uint64_t keys[kReplayLds];
    __shared__ __attribute__((aligned(16))) double2 ord[64]; // the terms of the chunk being walked, in key order
    const bool calc_mit = p.method == ISSL_METHOD_MIT || p.method == ISSL_METHOD_AND || p.method == ISSL_METHOD_OR ||
                          p.method == ISSL_METHOD_AVG;
    const bool calc_cfd = p.method == ISSL_METHOD_CFD || p.method == ISSL_METHOD_AND || p.method == ISSL_METHOD_OR ||
                          p.method == ISSL_METHOD_AVG;
    const uint32_t lane = threadIdx.x;
    // A guide beyond its hit slots in this batch: the lane's next batches get the whole tail; a batch that was enqueued
    // WITHOUT it (Workspace::lean_tail) is run again.
    if (blockIdx.x == 0 && lane == 0 && ws.counters->overflowed != 0u) atomicOr(&ws.sticky[0], ws.lean_tail ? 6u : 4u);

    for (uint32_t g = blockIdx.x; g < n; g += gridDim.x) {
        const uint32_t h = ws.gcount[g];
        if (h > kReplayLds) continue; // k_replay_mid's, k_replay_big's
        // the guide's keys and terms: in its hit slots, or (no slots: issl_dump_hits, ...) its segment of the grouped arrays
        const bool slots = ws.slot_hits >= kReplayLds;
        const uint32_t h0 = slots ? 0u : ws.goff[g];
        const SlotRec *__restrict__ srec = ws.slots + static_cast<uint64_t>(g) * ws.slot_hits;
        const uint64_t *__restrict__ skeys = ws.sorted + h0;
        const double2 *__restrict__ sterms = reinterpret_cast<const double2 *>(ws.terms) + h0;
        auto key_of = [&](uint32_t i) { return slots ? srec[i].key : skeys[i]; };
        auto terms_of = [&](uint32_t i) { return slots ? *reinterpret_cast<const double2 *>(&srec[i].mit) : sterms[i]; };
        const uint64_t gsig = guides[g];
        double tot_mit = 0.0, tot_cfd = 0.0;
        uint32_t kept = 0;
        bool stop = false;

        // Running totals in key order, same operations as the reference's (:394,:460), early exit of :467-496.
        auto accumulate = [&](double mit_term, double cfd_term, uint32_t cnt) {
            stop = accumulate_chunk(mit_term, cfd_term, cnt, p, lane, tot_mit, tot_cfd, kept, ord);
        };

        // The terms of every hit were computed by k_verify and sit next to the keys (key_of / terms_of above);
        // what is left is putting them in key order and adding them up.  issl_dump_hits also wants the expanded
        // records: those are looked up here (hit_terms), the totals still come from the stored terms.
        if (h <= 64) {
            // Common case: no sort.  Lane l takes key l and its terms, finds the rank of its key among the h keys by
            // counting, and drops the terms at that rank; lane r then owns the r-th hit in key order.
            uint64_t key = ~0ull;
            double2 mine = make_double2(0.0, 0.0);
            issl_hit rec{};
            if (lane < h) {
                key = key_of(lane);
                mine = terms_of(lane);
                if (out_hits) rec = hit_terms(v, gsig, g, key, calc_mit, calc_cfd, true).rec;
            }
            uint32_t rank = 0;
            for (uint32_t j = 0; j < h; ++j) {
                const uint32_t klo = __builtin_amdgcn_readlane(static_cast<uint32_t>(key), static_cast<int>(j));
                const uint32_t khi = __builtin_amdgcn_readlane(static_cast<uint32_t>(key >> 32), static_cast<int>(j));
                const uint64_t other = (static_cast<uint64_t>(khi) << 32) | klo;
                rank += (other < key) ? 1u : 0u;
            }
            if (lane < h) {
                ord[rank] = mine;
                if (out_hits) out_hits[h0 + rank] = rec;
            } else {
                ord[lane] = make_double2(0.0, 0.0); // (ranks are below h: nobody else writes here)
            }
            __syncthreads();
            const double2 t = ord[lane];
            accumulate(t.x, t.y, h);
        } else {
            // (slice, position) of every key with the key's index behind it, sorted in LDS; the terms follow by index
            uint64_t *data = keys;
            for (uint32_t i = lane; i < h; i += 64) keys[i] = ((key_of(i) & ((1ull << kKeyGuideShift) - 1ull)) << 9) | i; // h <= 512
            __syncthreads();
            wave_sort(data, h);
            __syncthreads();
            for (uint32_t base = 0; base < h && !stop; base += 64) {
                const uint32_t idx = base + lane;
                double2 mine = make_double2(0.0, 0.0);
                if (idx < h) {
                    const uint64_t sv = data[idx];
                    mine = terms_of(static_cast<uint32_t>(sv & 511ull));
                    if (out_hits)
                        out_hits[h0 + idx] = hit_terms(v, gsig, g, (static_cast<uint64_t>(g) << kKeyGuideShift) | (sv >> 9), calc_mit, calc_cfd, true).rec;
                }
                __syncthreads(); // (the walk of the chunk before has read `ord`)
                ord[lane] = mine;
                __syncthreads();
                accumulate(mine.x, mine.y, (h - base < 64u) ? h - base : 64u);
            }
        }
        if (lane == 0) {
            out_mit[g] = 10000.0 / (100.0 + tot_mit); // :505
            out_cfd[g] = 10000.0 / (100.0 + tot_cfd); // :506
            if (out_kept) out_kept[g] = kept;
        }
        __syncthreads();
    }
}

// Guides with kReplayLds < hits <= kMidHits (on skewed data four guides in ten): one 256-thread workgroup each, the terms
// k_verify left fetched by the hit's index.  One slice at a time, as the reference walks them (:330): the slice's keys are
// gathered into LDS, ranked by counting (no barrier inside: a bitonic network over 2048 keys costs 66 barrier-separated
// stages, 170 us per guide), their terms fetched -- all of the slice's at once -- and dropped at their ranks; wave 0 then
// walks the terms in LDS.  Such a guide usually leaves through the early exit (:467-496) inside its first slice (median:
// 295 hits walked of 1024 found), and the slices behind the exit are never touched.  A guide with a slice of more than
// kMidSlice hits is handed on to k_replay_big (second list).
// Round 4, later: (a) a slice of more than kMidDirect hits is ranked INSIDE 256 groups of the range its ids span (one
// counting pass in LDS puts the ids in group order first): len * len / 256 comparisons on evenly spread ids instead of
// len * len -- the kernel was bound by the vector instructions of the all-against-all count (0.67 G of them per 100 k guides
// of the skewed index); (b) the workgroups take the entries of the guide list one at a time from a device-wide ticket
// (asked for one guide ahead), not every gridDim-th entry: the 2048 workgroups are not all resident (7 per CU), and the
// stragglers of a static split ran alone on an empty chip for a quarter of the launch.
constexpr uint32_t kMidSlice = 1024;
constexpr uint32_t kBigSmall = 16384; // up to this many hits of a guide: the 256-thread build of k_replay_big
constexpr uint32_t kMidDirect = 256;  // up to this many hits in a slice: ranked against all of them, one per thread
// The next entry of the many-hit guide list for this workgroup (`which`: Counters::replay_next), handed to all its threads
// through LDS; the ticket after it is asked for at once, so that its round trip runs beside the guide's work.  (Two LDS
// words, used in turn: a wave that is late reading this guide's entry must not find the next one's there.)
struct ReplayTicket {
    uint32_t next = 0, turn = 0;
};
__device__ __forceinline__ uint32_t replay_take(ReplayTicket &t, uint32_t *cur /*LDS[2]*/, Counters *counters, uint32_t which,
                                                bool first, uint32_t n_entries)
{
    // no more entries than workgroups (a small batch, a few many-hit guides): one each, no ticket, no barrier -- the kernel then
    // lasts as long as its slowest guide, and the round trip of the atomic is part of that
    if (n_entries <= gridDim.x) return first ? blockIdx.x : n_entries;
    if (threadIdx.x == 0) {
        if (first) t.next = atomicAdd(&counters->replay_next[which], 1u);
        cur[t.turn] = t.next;
    }
    __syncthreads();
    const uint32_t b = cur[t.turn];
    t.turn ^= 1u;
    if (threadIdx.x == 0) t.next = atomicAdd(&counters->replay_next[which], 1u);
    return b;
}

__global__ __launch_bounds__(256, 6) void k_replay_mid(ImageView v, Workspace ws, const uint64_t *__restrict__ guides,
                                                    ScoreParams p, double *__restrict__ out_mit,
                                                    double *__restrict__ out_cfd, uint32_t *__restrict__ out_kept,
                                                    issl_hit *__restrict__ out_hits)
{
    short_kernel_priority();
    __shared__ __attribute__((aligned(16))) uint32_t head[kMidSlice]; // the slice's keys: site ids or positions (distinct) ...
    __shared__ uint16_t head_idx[kMidSlice];                           // ... and the index of the hit each belongs to
    __shared__ __attribute__((aligned(16))) double2 tmc[kMidSlice];    // its terms {mit, cfd} in key order (before that: the
                                                                       // slice's ids and hit indexes in group order)
    __shared__ uint32_t slice_cnt[kMaxSlices];
    __shared__ uint32_t group_at[257], group_cur[256], id_min, id_max;
    __shared__ uint32_t head_fill, stopped_s, carry_kept, cur_entry[2];
    __shared__ double carry_mit, carry_cfd;
    const bool calc_mit = p.method == ISSL_METHOD_MIT || p.method == ISSL_METHOD_AND || p.method == ISSL_METHOD_OR ||
                          p.method == ISSL_METHOD_AVG;
    const bool calc_cfd = p.method == ISSL_METHOD_CFD || p.method == ISSL_METHOD_AND || p.method == ISSL_METHOD_OR ||
                          p.method == ISSL_METHOD_AVG;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_big = ws.counters->n_big;
    const double2 *__restrict__ terms2 = reinterpret_cast<const double2 *>(ws.terms);
    if (blockIdx.x >= n_big) return; // (more workgroups than entries -- on an even index there are none: no ticket is taken)
    ReplayTicket ticket;
    for (bool first = true;; first = false) {
        const uint32_t b = replay_take(ticket, cur_entry, ws.counters, 0u, first, n_big);
        if (b >= n_big) break;
        const uint32_t g = ws.gcur_big[b];
        const uint32_t h = ws.gcount[g];
        if (h > kMidHits) { // (uniform) k_replay_big's: onto the list of its 256-thread build, or -- from the far end of the same array -- of the other
            if (threadIdx.x == 0) {
                if (h <= kBigSmall) ws.gcur_big2[atomicAdd(&ws.counters->n_big2, 1u)] = g;
                else ws.gcur_big2[static_cast<uint32_t>(ws.cap_guides) - atomicAdd(&ws.counters->n_big3, 1u)] = g;
            }
            continue;
        }
        const uint32_t h0 = ws.goff[g];
        const uint64_t gsig = guides[g];
        // hit i of the guide: in its hit slots below slot_hits, in its segment of the grouped arrays from there on
        const uint32_t in_slots = ws.slot_hits;
        const SlotRec *__restrict__ srec = ws.slots + static_cast<uint64_t>(g) * in_slots;
        const uint64_t *__restrict__ gkeys = ws.sorted + h0;
        // diagnostics (ISSL_SCAN_STAMPS): phase clocks of the first 4096 listed guides, like k_replay_big's
        unsigned long long *st = (ws.stamps && b < 4096u) ? ws.stamps + kStampsMid + 16u * b : nullptr;
        if (st && threadIdx.x == 0) { st[0] = __builtin_amdgcn_s_memrealtime(); st[1] = h; st[15] = blockIdx.x; st[14] = 1; }
        if (threadIdx.x < kMaxSlices) slice_cnt[threadIdx.x] = 0;
        if (threadIdx.x == 0) { stopped_s = 0; carry_mit = 0.0; carry_cfd = 0.0; carry_kept = 0; }
        __syncthreads();
        // the guide's keys, eight per thread, all asked for at once and kept in registers: every later phase works from
        // them (a phase that goes back to memory costs a round trip of microseconds, and a guide is a chain of phases)
        uint64_t mykey[kMidHits / 256];
#pragma unroll
        for (uint32_t k = 0; k < kMidHits / 256; ++k) {
            const uint32_t i = k * 256u + threadIdx.x;
            mykey[k] = i < h ? (i < in_slots ? srec[i].key : gkeys[i]) : ~0ull;
        }
#pragma unroll
        for (uint32_t k = 0; k < kMidHits / 256; ++k) { // hits per slice (one LDS atomic per wave and slice present)
            if (k * 256u >= h) break;
            const uint32_t sl = mykey[k] != ~0ull ? static_cast<uint32_t>(mykey[k] >> kKeySliceShift) & kKeySliceMask : kKeySliceMask;
            for (uint32_t s2 = 0; s2 < v.n_slices; ++s2) {
                const uint64_t mm = __ballot(sl == s2);
                if (mm != 0ull && lane == 0) atomicAdd(&slice_cnt[s2], static_cast<uint32_t>(__builtin_popcountll(mm)));
            }
        }
        __syncthreads();
        uint32_t longest = 0;
        for (uint32_t s2 = 0; s2 < v.n_slices; ++s2) longest = slice_cnt[s2] > longest ? slice_cnt[s2] : longest;
        if (longest > kMidSlice) { // (uniform) a slice that does not fit: the slice-by-slice kernel with the larger buffers
            if (threadIdx.x == 0) ws.gcur_big2[atomicAdd(&ws.counters->n_big2, 1u)] = g;
            __syncthreads(); // (the next guide's reset of slice_cnt must not overtake a wave that is still reading it)
            continue;
        }
        if (st && threadIdx.x == 0) st[2] = __builtin_amdgcn_s_memrealtime();
        uint32_t walked = 0; // hits of the slices done so far (= where this slice's hits start in scoring order)
        for (uint32_t s2 = 0; s2 < v.n_slices; ++s2) {
            const uint32_t len = slice_cnt[s2];
            if (len == 0) continue;
            const bool grouped_rank = len > kMidDirect; // (uniform)
            if (threadIdx.x == 0) { head_fill = 0; id_min = 0xFFFFFFFFu; id_max = 0u; }
            group_cur[threadIdx.x] = 0;
            __syncthreads();
            uint32_t mn = 0xFFFFFFFFu, mx = 0u;
#pragma unroll
            for (uint32_t k = 0; k < kMidHits / 256; ++k) { // gather the slice's keys (one cursor bump per wave)
                if (k * 256u >= h) break;
                const uint64_t key = mykey[k];
                const bool mine = key != ~0ull && (static_cast<uint32_t>(key >> kKeySliceShift) & kKeySliceMask) == s2;
                const uint64_t mm = __ballot(mine);
                if (mm == 0ull) continue;
                uint32_t at = 0;
                if (lane == 0) at = atomicAdd(&head_fill, static_cast<uint32_t>(__builtin_popcountll(mm)));
                at = __builtin_amdgcn_readfirstlane(at);
                if (mine) {
                    const uint32_t to = at + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mm >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mm), 0u));
                    const uint32_t id = static_cast<uint32_t>(key);
                    head[to] = id;
                    head_idx[to] = static_cast<uint16_t>(k * 256u + threadIdx.x);
                    mn = id < mn ? id : mn;
                    mx = id > mx ? id : mx;
                }
            }
            if (grouped_rank && mn <= mx) { atomicMin(&id_min, mn); atomicMax(&id_max, mx); } // the range the slice's ids span
            __syncthreads();
            // Rank by counting against the WHOLE slice (the keys are distinct: ids / positions are), K keys per thread, the
            // slice's ids read four at a time by every thread at once; then the terms to their ranks.
            auto rank_against_all = [&](auto k_tag) {
                constexpr uint32_t K = decltype(k_tag)::value;
                if (threadIdx.x < 4u && len + threadIdx.x < ((len + 3u) & ~3u)) head[len + threadIdx.x] = 0xFFFFFFFFu; // (read four at a time)
                __syncthreads();
                uint32_t mine[K], rk[K];
#pragma unroll
                for (uint32_t k = 0; k < K; ++k) { mine[k] = threadIdx.x + k * 256u < len ? head[threadIdx.x + k * 256u] : 0xFFFFFFFFu; rk[k] = 0; }
                const uint4 *quads = reinterpret_cast<const uint4 *>(head);
#pragma unroll 4
                for (uint32_t j = 0; j < (len + 3u) / 4u; ++j) {
                    const uint4 q = quads[j];
#pragma unroll
                    for (uint32_t k = 0; k < K; ++k)
                        rk[k] += (q.x < mine[k] ? 1u : 0u) + (q.y < mine[k] ? 1u : 0u) + (q.z < mine[k] ? 1u : 0u) + (q.w < mine[k] ? 1u : 0u);
                }
#pragma unroll
                for (uint32_t k = 0; k < K; ++k) {
                    if (threadIdx.x + k * 256u >= len) continue;
                    const uint32_t idx = head_idx[threadIdx.x + k * 256u];
                    const double2 t2 = idx < in_slots ? *reinterpret_cast<const double2 *>(&srec[idx].mit) : terms2[h0 + idx];
                    tmc[rk[k]] = t2;
                    if (out_hits)
                        out_hits[h0 + walked + rk[k]] = hit_terms(v, gsig, g, (static_cast<uint64_t>(g) << kKeyGuideShift) |
                                                                      (static_cast<uint64_t>(s2) << kKeySliceShift) | mine[k],
                                                                  calc_mit, calc_cfd, true).rec;
                }
            };
            if (!grouped_rank) {
                rank_against_all(std::integral_constant<uint32_t, 1u>{});
            } else {
                // Ranked inside 256 groups of the range the ids span (the hits of a guide in one slice share the slice's bases:
                // on a text-sorted index their ids lie in a narrow range far from zero): group sizes, their prefix, the ids and
                // hit indexes in group order (in the memory the terms will take), then every id against its own group only.
                // Ids that pile up in one group -- a repeat family: neighbours in the text-sorted site table -- leave nothing to
                // gain there: a slice whose largest group holds more than a quarter of it is ranked against all of it.
                const uint32_t low = id_min, top = id_max - low;
                const uint32_t shift = top < 256u ? 0u : 24u - static_cast<uint32_t>(__builtin_clz(top)); // (id - low) >> shift < 256
#pragma unroll
                for (uint32_t k = 0; k < kMidSlice / 256; ++k)
                    if (threadIdx.x + k * 256u < len) atomicAdd(&group_cur[(head[threadIdx.x + k * 256u] - low) >> shift], 1u);
                __syncthreads();
                if (threadIdx.x < 64) { // exclusive scan of the 256 group sizes by one wave, 4 per lane; the cursors start there
                    uint32_t v4[4], sum = 0, big = 0;
                    for (uint32_t k = 0; k < 4; ++k) { v4[k] = group_cur[threadIdx.x * 4 + k]; sum += v4[k]; big = v4[k] > big ? v4[k] : big; }
                    uint32_t x = sum;
                    for (uint32_t d = 1; d < 64; d <<= 1) {
                        const uint32_t y = __shfl_up(x, d, 64);
                        if (threadIdx.x >= d) x += y;
                    }
                    for (uint32_t d = 32; d > 0; d >>= 1) { const uint32_t y = __shfl_xor(big, d, 64); big = y > big ? y : big; }
                    uint32_t run = x - sum;
                    for (uint32_t k = 0; k < 4; ++k) { group_at[threadIdx.x * 4 + k] = run; group_cur[threadIdx.x * 4 + k] = run; run += v4[k]; }
                    if (threadIdx.x == 63) group_at[256] = run;
                    if (threadIdx.x == 0) id_max = big; // (the range is in registers: the word now says how large the largest group is)
                }
                __syncthreads();
                if (id_max * 4u > len) { // (uniform)
                    if (len <= 512u) rank_against_all(std::integral_constant<uint32_t, 2u>{});
                    else rank_against_all(std::integral_constant<uint32_t, 4u>{});
                } else {
                uint32_t *ids2 = reinterpret_cast<uint32_t *>(tmc);                 // [kMidSlice]
                uint16_t *idx2 = reinterpret_cast<uint16_t *>(ids2 + kMidSlice);    // [kMidSlice]
#pragma unroll
                for (uint32_t k = 0; k < kMidSlice / 256; ++k) {
                    const uint32_t i = threadIdx.x + k * 256u;
                    if (i >= len) continue;
                    const uint32_t id = head[i];
                    const uint32_t to = atomicAdd(&group_cur[(id - low) >> shift], 1u);
                    ids2[to] = id;
                    idx2[to] = head_idx[i];
                }
                if (threadIdx.x < 4u && len + threadIdx.x < ((len + 3u) & ~3u)) ids2[len + threadIdx.x] = 0xFFFFFFFFu; // (read four at a time)
                __syncthreads();
                // every id against its own group, four ids per LDS read: the quads that cover the group also hold ids of the
                // groups around it -- smaller ones below (each counts: the rank starts at the quad, not at the group), larger
                // ones and the padding above (none counts); id and hit index go to the id's rank (`head`, `head_idx`: their
                // first contents are in group order now), so that nothing is carried across the barrier but the arrays
                const uint4 *quads2 = reinterpret_cast<const uint4 *>(ids2);
#pragma unroll
                for (uint32_t k = 0; k < kMidSlice / 256; ++k) {
                    const uint32_t i = threadIdx.x + k * 256u;
                    if (i < len) {
                        const uint32_t id = ids2[i];
                        const uint32_t g0 = group_at[(id - low) >> shift], g1 = group_at[((id - low) >> shift) + 1u];
                        uint32_t r = g0 & ~3u;
                        for (uint32_t j = g0 >> 2; j < (g1 + 3u) >> 2; ++j) {
                            const uint4 q = quads2[j];
                            r += (q.x < id ? 1u : 0u) + (q.y < id ? 1u : 0u) + (q.z < id ? 1u : 0u) + (q.w < id ? 1u : 0u);
                        }
                        head[r] = id;
                        head_idx[r] = idx2[i];
                    }
                }
                __syncthreads(); // (everybody has read the ids in group order: the terms may land on them)
#pragma unroll
                for (uint32_t k = 0; k < kMidSlice / 256; ++k) {
                    const uint32_t i = threadIdx.x + k * 256u;
                    if (i < len) {
                        const uint32_t idx = head_idx[i];
                        tmc[i] = idx < in_slots ? *reinterpret_cast<const double2 *>(&srec[idx].mit) : terms2[h0 + idx];
                        if (out_hits)
                            out_hits[h0 + walked + i] = hit_terms(v, gsig, g, (static_cast<uint64_t>(g) << kKeyGuideShift) |
                                                                      (static_cast<uint64_t>(s2) << kKeySliceShift) | head[i],
                                                                  calc_mit, calc_cfd, true).rec;
                    }
                }
                }
            }
            if (threadIdx.x < 8u && len + threadIdx.x < ((len + 7u) & ~7u)) tmc[len + threadIdx.x] = make_double2(0.0, 0.0); // (walked eight at a time)
            __syncthreads();
            if (st && threadIdx.x == 0 && walked == 0) { st[3] = __builtin_amdgcn_s_memrealtime(); st[4] = len; }
            if (threadIdx.x < 64) { // wave 0 walks the slice
                double tot_mit = carry_mit, tot_cfd = carry_cfd;
                uint32_t kept = carry_kept;
                bool stop = false;
                for (uint32_t base = 0; base < len && !stop; base += 64) {
                    const uint32_t idx = base + lane;
                    const double2 mine2 = idx < len ? tmc[idx] : make_double2(0.0, 0.0);
                    stop = accumulate_chunk(mine2.x, mine2.y, (len - base < 64u) ? len - base : 64u, p, lane, tot_mit, tot_cfd, kept,
                                            tmc + base);
                }
                if (lane == 0) { carry_mit = tot_mit; carry_cfd = tot_cfd; carry_kept = kept; stopped_s = stop ? 1u : 0u; }
            }
            __syncthreads();
            walked += len;
            if (stopped_s != 0u) break; // (uniform) the slices behind the exit are never touched
        }
        if (threadIdx.x == 0) {
            out_mit[g] = 10000.0 / (100.0 + carry_mit); // :505
            out_cfd[g] = 10000.0 / (100.0 + carry_cfd); // :506
            if (out_kept) out_kept[g] = carry_kept;
            if (st) { st[7] = __builtin_amdgcn_s_memrealtime(); st[8] = carry_kept; st[9] = walked; }
        }
        __syncthreads();
    }
}

// Sort by counting for the slices of a big guide: positions (low 32 key bits; guide and slice are common to the
// slice) sit in LDS, every thread finds the rank of its K positions by comparing them with all `len` of them
// (broadcast reads, no barrier inside) and writes the full keys to their final places.  A bitonic network over the same
// keys costs ~80-90 workgroup barriers; positions are distinct, so the ranks are a permutation.
template <uint32_t K, uint32_t THREADS>
__device__ __forceinline__ void rank_sort_slice(const uint32_t *pos_lds, uint32_t len, uint64_t high_bits,
                                                uint64_t *__restrict__ dst)
{
    uint32_t mine[K], rk[K];
#pragma unroll
    for (uint32_t k = 0; k < K; ++k) {
        const uint32_t idx = threadIdx.x + k * THREADS;
        mine[k] = idx < len ? pos_lds[idx] : 0xFFFFFFFFu;
        rk[k] = 0;
    }
    // four positions per LDS read (the array is padded with 0xFFFFFFFF, which is below nothing), several reads in
    // flight: the loop is bound by LDS latency otherwise
    const uint4 *quads = reinterpret_cast<const uint4 *>(pos_lds);
    const uint32_t n_quads = (len + 3u) >> 2;
#pragma unroll 4
    for (uint32_t j = 0; j < n_quads; ++j) {
        const uint4 q = quads[j];
#pragma unroll
        for (uint32_t k = 0; k < K; ++k)
            rk[k] += (q.x < mine[k] ? 1u : 0u) + (q.y < mine[k] ? 1u : 0u) + (q.z < mine[k] ? 1u : 0u) +
                     (q.w < mine[k] ? 1u : 0u);
    }
#pragma unroll
    for (uint32_t k = 0; k < K; ++k)
        if (threadIdx.x + k * THREADS < len) dst[rk[k]] = high_bits | mine[k];
}

// Longer slices: the same ranking, but only against the positions that share the top 8 bits (of the slice's largest
// position): one counting pass groups the positions by those bits in `grouped`, then every position is ranked inside
// its group -- len * len / 256 comparisons on evenly spread positions instead of len * len.
__device__ __forceinline__ void rank_sort_slice_grouped(const uint32_t *pos_lds, uint32_t *grouped, uint32_t *group_at /*[257]*/,
                                                        uint32_t *group_cur /*[256]*/, uint32_t *max_pos, uint32_t *min_pos,
                                                        uint32_t len, uint64_t high_bits, uint64_t *__restrict__ dst)
{
    if (threadIdx.x < 256) group_cur[threadIdx.x] = 0;
    if (threadIdx.x == 0) { *max_pos = 0; *min_pos = 0xFFFFFFFFu; }
    __syncthreads();
    uint32_t m = 0, mn = 0xFFFFFFFFu;
    for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) { const uint32_t q = pos_lds[i]; m = q > m ? q : m; mn = q < mn ? q : mn; }
    atomicMax(max_pos, m);
    atomicMin(min_pos, mn);
    __syncthreads();
    // (the hits of a guide in one slice share the slice's bases: on a text-sorted index their ids lie in a narrow range far
    // from zero, so the groups divide the range they span, not the values)
    const uint32_t low = *min_pos, top = *max_pos - low;
    const uint32_t shift = top < 256u ? 0u : 24u - static_cast<uint32_t>(__builtin_clz(top)); // group = (pos - low) >> shift < 256
    for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) atomicAdd(&group_cur[(pos_lds[i] - low) >> shift], 1u);
    __syncthreads();
    if (threadIdx.x < 64) { // exclusive scan of the 256 group sizes by one wave, 4 per lane
        uint32_t v4[4], sum = 0;
        for (uint32_t k = 0; k < 4; ++k) { v4[k] = group_cur[threadIdx.x * 4 + k]; sum += v4[k]; }
        uint32_t x = sum;
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d, 64);
            if (threadIdx.x >= d) x += y;
        }
        uint32_t run = x - sum;
        for (uint32_t k = 0; k < 4; ++k) { group_at[threadIdx.x * 4 + k] = run; run += v4[k]; }
        if (threadIdx.x == 63) group_at[256] = run;
    }
    __syncthreads();
    if (threadIdx.x < 256) group_cur[threadIdx.x] = group_at[threadIdx.x];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) {
        const uint32_t pos = pos_lds[i];
        grouped[atomicAdd(&group_cur[(pos - low) >> shift], 1u)] = pos;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) {
        const uint32_t pos = grouped[i];
        const uint32_t g0 = group_at[(pos - low) >> shift], g1 = group_at[((pos - low) >> shift) + 1u];
        uint32_t rk = g0;
        for (uint32_t j = g0; j < g1; ++j) rk += grouped[j] < pos ? 1u : 0u;
        dst[rk] = high_bits | pos;
    }
}

// Guides with many hits (dense neighbourhoods, repeats): one 1024-thread workgroup each, one slice at a time: sort
// the slice's keys (by counting in LDS up to 8192 per slice, else a bitonic network in HBM), compute the terms of its
// hits in parallel, let wave 0 add them up in key order with the reference's running totals and early exit.
template <uint32_t THREADS, uint32_t LDS_HITS>
__global__ __launch_bounds__(THREADS, THREADS < 1024u ? 6 : 4) void k_replay_big(ImageView v, Workspace ws, const uint64_t *__restrict__ guides,
                                                     ScoreParams p, double *__restrict__ out_mit,
                                                     double *__restrict__ out_cfd, uint32_t *__restrict__ out_kept,
                                                     issl_hit *__restrict__ out_hits)
{
    short_kernel_priority();
    __shared__ __attribute__((aligned(16))) uint32_t pos_lds[LDS_HITS];
    __shared__ uint32_t grouped[LDS_HITS];
    __shared__ uint32_t group_at[257], group_cur[256], max_pos, min_pos;
    __shared__ uint32_t outer_at[257]; // a slice beyond the LDS: where its 256 id groups start once it is in group order
    __shared__ uint32_t slice_cnt[kMaxSlices], slice_off[kMaxSlices + 1], slice_cur[kMaxSlices];
    __shared__ uint32_t walk_stopped, head_groups, head_count, head_fill;
    __shared__ __attribute__((aligned(16))) double2 walk_terms[64]; // wave 0: the terms of the 64 hits it is adding up
    const bool calc_mit = p.method == ISSL_METHOD_MIT || p.method == ISSL_METHOD_AND || p.method == ISSL_METHOD_OR ||
                          p.method == ISSL_METHOD_AVG;
    const bool calc_cfd = p.method == ISSL_METHOD_CFD || p.method == ISSL_METHOD_AND || p.method == ISSL_METHOD_OR ||
                          p.method == ISSL_METHOD_AVG;
    const uint32_t lane = threadIdx.x & 63u;
    // The guides of this build: a list k_replay_mid has made while it went through the shared one (more than kMidHits hits, or
    // handed on; the 1024-thread build's from the far end of the array).  (Each build used to walk the whole shared list and skip
    // what was not its own: 40 k entries of three dependent loads each, for the few dozen guides of the 1024-thread build.)
    const uint32_t n_mine = THREADS < 1024u ? ws.counters->n_big2 : ws.counters->n_big3;
    __shared__ uint32_t cur_entry[2];
    if (blockIdx.x >= n_mine) return; // (more workgroups than entries: no ticket is taken)
    ReplayTicket ticket;
    for (bool first = true;; first = false) { // (entries by ticket, as in k_replay_mid)
        const uint32_t b = replay_take(ticket, cur_entry, ws.counters, THREADS < 1024u ? 1u : 2u, first, n_mine);
        if (b >= n_mine) break;
        const uint32_t g = THREADS < 1024u ? ws.gcur_big2[b] : ws.gcur_big2[static_cast<uint32_t>(ws.cap_guides) - b];
        const uint32_t h = ws.gcount[g];
        const uint32_t h0 = ws.goff[g];
        // k_replay_mid's, or the other build's: up to kBigSmall hits a 256-thread workgroup with 2048 hits per slice in
        // LDS (eight per CU: what such a guide costs is a chain of barriers and memory round trips, and what counts is
        // how many are in flight), beyond that 1024 threads with 7680 (two per CU).  (uniform over the workgroup)
        const uint64_t gsig = guides[g];
        uint64_t *seg = ws.sorted + h0;
        uint64_t *tmp = ws.raw + h0; // the raw records are dead once they are grouped; the buffer holds >= all hits
        // diagnostics (ISSL_SCAN_STAMPS, tools/replay_stamps.py): phase clocks of the first 4096 big guides
        unsigned long long *st = (ws.stamps && b < 4096u) ? ws.stamps + (THREADS < 1024u ? kStampsBig256 : kStampsBig1024) + 16u * b : nullptr; // (behind k_replay_mid's)
        if (st && threadIdx.x == 0) { st[0] = __builtin_amdgcn_s_memrealtime(); st[1] = h; st[15] = blockIdx.x; }

        // The scoring order is (slice, position in bucket) and the walk usually ends inside the first slice (the
        // totals pass the threshold, :467-496): split the keys by slice (bits 32..34) and sort and walk one slice at
        // a time -- a fifth of the sorting work per step, in LDS up to 8192 hits PER SLICE, and none at all for the
        // slices behind the exit.
        if (threadIdx.x < kMaxSlices) { slice_cnt[threadIdx.x] = 0; slice_cur[threadIdx.x] = 0; }
        // hit slots: the guide's first slot_hits keys join the rest in its segment (h > slot_hits here: all of them are there)
        for (uint32_t i = threadIdx.x; i < (ws.slot_hits < h ? ws.slot_hits : h); i += blockDim.x) seg[i] = ws.slots[static_cast<uint64_t>(g) * ws.slot_hits + i].key;
        __syncthreads();
        for (uint32_t base = 0; base < h; base += blockDim.x) {
            const uint32_t i = base + threadIdx.x;
            const uint32_t sl = i < h ? static_cast<uint32_t>(seg[i] >> kKeySliceShift) & kKeySliceMask : kKeySliceMask;
            for (uint32_t s2 = 0; s2 < v.n_slices; ++s2) {
                const uint64_t m = __ballot(sl == s2);
                if (m != 0ull && lane == 0) atomicAdd(&slice_cnt[s2], static_cast<uint32_t>(__builtin_popcountll(m)));
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t run = 0;
            for (uint32_t s2 = 0; s2 < kMaxSlices; ++s2) { slice_off[s2] = run; run += slice_cnt[s2]; }
            slice_off[kMaxSlices] = run;
        }
        __syncthreads();
        for (uint32_t base = 0; base < h; base += blockDim.x) {
            const uint32_t i = base + threadIdx.x;
            const uint64_t key = i < h ? seg[i] : 0ull;
            const uint32_t sl = i < h ? static_cast<uint32_t>(key >> kKeySliceShift) & kKeySliceMask : kKeySliceMask;
            for (uint32_t s2 = 0; s2 < v.n_slices; ++s2) {
                const uint64_t m = __ballot(sl == s2);
                if (m == 0ull) continue;
                uint32_t at = 0;
                if (lane == 0) at = atomicAdd(&slice_cur[s2], static_cast<uint32_t>(__builtin_popcountll(m)));
                at = __builtin_amdgcn_readfirstlane(at);
                if (sl == s2)
                    tmp[slice_off[s2] + at + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                                                        __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u))] = key;
            }
        }
        __syncthreads();

        if (st && threadIdx.x == 0) st[2] = __builtin_amdgcn_s_memrealtime();
        double tot_mit = 0.0, tot_cfd = 0.0;
        uint32_t kept = 0;
        bool stop = false;
        auto accumulate = [&](double mit_term, double cfd_term, uint32_t cnt) { // (wave 0 only)
            __builtin_amdgcn_wave_barrier();
            walk_terms[lane] = make_double2(mit_term, cfd_term);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            stop = accumulate_chunk(mit_term, cfd_term, cnt, p, lane, tot_mit, tot_cfd, kept, walk_terms);
        };
        for (uint32_t s2 = 0; s2 < v.n_slices; ++s2) {
            const uint32_t off = slice_off[s2], len = slice_cnt[s2];
            if (len == 0) continue; // uniform over the workgroup
            if (st && threadIdx.x == 0 && s2 == 0) { st[3] = __builtin_amdgcn_s_memrealtime(); st[4] = len; }
            uint64_t *dst = seg + off;
            const uint64_t high_bits = (static_cast<uint64_t>(g) << kKeyGuideShift) | (static_cast<uint64_t>(s2) << kKeySliceShift);
            // Terms of hits [from, to) of the slice (dst[] holds them in key order) by the whole workgroup, a block of
            // blockDim.x at a time, each block added up in key order by wave 0 before the next one is worked out: a
            // guide like this usually leaves through the early exit within its first hits (:467-496), and the terms
            // cost two or three random reads each.  Leaves walk_stopped set when the exit was taken.
            auto walk = [&](uint32_t from, uint32_t to) {
                for (uint32_t blk = from; blk < to; blk += blockDim.x) {
                    const uint32_t i = blk + threadIdx.x;
                    if (i < to) {
                        const HitTerms t = hit_terms(v, gsig, g, dst[i], calc_mit, calc_cfd, out_hits != nullptr);
                        ws.terms[2ull * (h0 + off + i)] = t.mit;
                        ws.terms[2ull * (h0 + off + i) + 1] = t.cfd;
                        if (out_hits) out_hits[h0 + off + i] = t.rec;
                    }
                    __syncthreads();
                    if (threadIdx.x < 64) {
                        const uint32_t end = (to - blk < blockDim.x) ? to : blk + blockDim.x;
                        for (uint32_t base = blk; base < end && !stop; base += 64) {
                            const uint32_t idx = base + lane;
                            const double mit_term = idx < end ? ws.terms[2ull * (h0 + off + idx)] : 0.0;
                            const double cfd_term = idx < end ? ws.terms[2ull * (h0 + off + idx) + 1] : 0.0;
                            accumulate(mit_term, cfd_term, (end - base < 64u) ? end - base : 64u);
                        }
                        if (lane == 0) walk_stopped = stop ? 1u : 0u;
                    }
                    __syncthreads();
                    if (walk_stopped != 0u) break; // uniform
                }
            };
            auto sort_in_lds = [&](uint32_t n, uint64_t *out) { // pos_lds[0 .. n) -> out[0 .. n) in key order
                if (n <= THREADS) rank_sort_slice<1, THREADS>(pos_lds, n, high_bits, out);
                else rank_sort_slice_grouped(pos_lds, grouped, group_at, group_cur, &max_pos, &min_pos, n, high_bits, out);
                __syncthreads();
            };
            uint32_t walked = 0;
            uint32_t g_low = 0, g_shift = 0, g_done = 0; // the id groups of the head pass: (id - g_low) >> g_shift; the first g_done are walked
            if (threadIdx.x == 0) walk_stopped = 0;
            __syncthreads();
            if (len > THREADS) {
                // A guide like this all but always leaves through the early exit within the hits with the smallest ids of
                // its first slice (median: ~250 hits walked of thousands found), so those are tried first: the slice's
                // positions are counted by their top eight bits, the leading groups that hold at least 384 of them are
                // gathered, ranked by counting and walked; only a guide that survives them pays for the order of the whole
                // slice (a ranking inside unevenly filled groups in LDS; beyond the LDS a bitonic network in HBM of ~140
                // stages of memory round trips, which used to set the kernel's duration).
                if (threadIdx.x < 256) group_cur[threadIdx.x] = 0;
                if (threadIdx.x == 0) { max_pos = 0; min_pos = 0xFFFFFFFFu; head_fill = 0; }
                __syncthreads();
                uint32_t m = 0, mn = 0xFFFFFFFFu;
                for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) { const uint32_t q = static_cast<uint32_t>(tmp[off + i]); m = q > m ? q : m; mn = q < mn ? q : mn; }
                atomicMax(&max_pos, m);
                atomicMin(&min_pos, mn);
                __syncthreads();
                const uint32_t low = min_pos, top = max_pos - low; // (groups of the RANGE the slice's ids span: see rank_sort_slice_grouped)
                const uint32_t shift = top < 256u ? 0u : 24u - static_cast<uint32_t>(__builtin_clz(top));
                for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) atomicAdd(&group_cur[(static_cast<uint32_t>(tmp[off + i]) - low) >> shift], 1u);
                __syncthreads();
                if (threadIdx.x == 0) {
                    uint32_t run = 0, nb = 0;
                    while (nb < 256u && run < 384u) run += group_cur[nb++];
                    head_groups = nb;
                    head_count = run;
                }
                __syncthreads();
                const uint32_t nb = head_groups, cnt = head_count;
                g_low = low; g_shift = shift;
                constexpr uint32_t kHeadMax = 4u * THREADS < LDS_HITS ? 4u * THREADS : LDS_HITS; // what rank_sort_slice<4> takes
                if (cnt <= kHeadMax && cnt < len) {
                    for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) {
                        const uint32_t q = static_cast<uint32_t>(tmp[off + i]);
                        if (((q - low) >> shift) < nb) pos_lds[atomicAdd(&head_fill, 1u)] = q;
                    }
                    __syncthreads();
                    for (uint32_t i = cnt + threadIdx.x; i < ((cnt + 3u) & ~3u); i += blockDim.x) pos_lds[i] = 0xFFFFFFFFu;
                    __syncthreads();
                    if (cnt <= THREADS) rank_sort_slice<1, THREADS>(pos_lds, cnt, high_bits, dst);
                    else rank_sort_slice<4, THREADS>(pos_lds, cnt, high_bits, dst);
                    __syncthreads();
                    walk(0u, cnt);
                    walked = cnt;
                    g_done = nb;
                }
            }
            if (walk_stopped == 0u && walked < len) { // (uniform) the whole slice in key order
                if (len <= LDS_HITS) {
                    for (uint32_t i = threadIdx.x; i < ((len + 3u) & ~3u); i += blockDim.x)
                        pos_lds[i] = i < len ? static_cast<uint32_t>(tmp[off + i]) : 0xFFFFFFFFu;
                    __syncthreads();
                    sort_in_lds(len, dst);
                } else {
                    // A slice beyond the LDS.  Its keys go into group order first -- the 256 groups of the range its ids span
                    // that the head pass counted (group_cur), one pass with the cursors in LDS --, then run after run of
                    // consecutive groups that fit the LDS is sorted there and walked: the exit (:467-496) usually comes before
                    // the second run, and no run costs more than a slice that fits.  (A bitonic network over the whole slice in
                    // HBM, ~140 stages of memory round trips, set the duration of this kernel before: 0.5 ms per batch on the
                    // skewed index for a few dozen guides.)  Only a single group beyond the LDS -- ids piled up in 1/256 of the
                    // range -- still takes the network, alone.
                    if (threadIdx.x < 64) { // exclusive scan of the 256 group sizes by one wave, 4 per lane; the cursors start there
                        uint32_t v4[4], sum = 0;
                        for (uint32_t k = 0; k < 4; ++k) { v4[k] = group_cur[threadIdx.x * 4 + k]; sum += v4[k]; }
                        uint32_t x = sum;
                        for (uint32_t d = 1; d < 64; d <<= 1) {
                            const uint32_t y = __shfl_up(x, d, 64);
                            if (threadIdx.x >= d) x += y;
                        }
                        uint32_t run = x - sum;
                        for (uint32_t k = 0; k < 4; ++k) { outer_at[threadIdx.x * 4 + k] = run; group_cur[threadIdx.x * 4 + k] = run; run += v4[k]; }
                        if (threadIdx.x == 63) outer_at[256] = run;
                    }
                    __syncthreads();
                    for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) {
                        const uint64_t key = tmp[off + i];
                        dst[atomicAdd(&group_cur[(static_cast<uint32_t>(key) - g_low) >> g_shift], 1u)] = key;
                    }
                    __syncthreads();
                    for (uint32_t g_lo = g_done; g_lo < 256u;) { // (uniform: every thread reads the same LDS words)
                        const uint32_t start = outer_at[g_lo];
                        uint32_t g_hi = g_lo + 1u;
                        while (g_hi < 256u && outer_at[g_hi + 1u] - start <= LDS_HITS) ++g_hi;
                        const uint32_t n = outer_at[g_hi] - start;
                        g_lo = g_hi;
                        if (n == 0u) continue;
                        if (n <= LDS_HITS) {
                            for (uint32_t i = threadIdx.x; i < ((n + 3u) & ~3u); i += blockDim.x)
                                pos_lds[i] = i < n ? static_cast<uint32_t>(dst[start + i]) : 0xFFFFFFFFu;
                            __syncthreads();
                            sort_in_lds(n, dst + start);
                        } else {
                            wave_sort(dst + start, n);
                            __syncthreads();
                        }
                        walk(start, start + n);
                        if (walk_stopped != 0u) break;
                    }
                    walked = len; // (nothing is left for the walk below)
                }
            }
            if (st && threadIdx.x == 0 && s2 == 0) st[5] = __builtin_amdgcn_s_memrealtime();
            if (walk_stopped == 0u) walk(walked, len);
            if (walk_stopped != 0u) break; // uniform: the slices behind the exit are never sorted
        }
        if (threadIdx.x == 0) {
            out_mit[g] = 10000.0 / (100.0 + tot_mit); // :505
            out_cfd[g] = 10000.0 / (100.0 + tot_cfd); // :506
            if (out_kept) out_kept[g] = kept;
            if (st) { st[7] = __builtin_amdgcn_s_memrealtime(); st[8] = kept; }
        }
        __syncthreads();
    }
}

void launch_replay(const ImageView &v, const Workspace &ws, const uint64_t *d_guides, uint32_t n,
                   const ScoreParams &p, double *d_mit, double *d_cfd, uint32_t *d_kept, issl_hit *d_hitrec,
                   void *stream)
{
    if (n == 0) return;
    const uint32_t grid = n < 65536u ? n : 65536u;
    hipLaunchKernelGGL(k_replay, dim3(grid), dim3(64), 0, static_cast<hipStream_t>(stream), v, ws, d_guides, n, p,
                       d_mit, d_cfd, d_kept, d_hitrec);
    if (ws.lean_tail) return; // (predicted: no guide with more than kReplayLds hits; k_replay checks)
    hipLaunchKernelGGL(k_replay_mid, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(stream), v, ws, d_guides, p,
                       d_mit, d_cfd, d_kept, d_hitrec);
    hipLaunchKernelGGL((k_replay_big<256, 2048>), dim3(2048), dim3(256), 0, static_cast<hipStream_t>(stream), v, ws, d_guides, p,
                       d_mit, d_cfd, d_kept, d_hitrec);
    hipLaunchKernelGGL((k_replay_big<1024, kBigLds>), dim3(512), dim3(1024), 0, static_cast<hipStream_t>(stream), v, ws, d_guides, p,
                       d_mit, d_cfd, d_kept, d_hitrec);
}

} // namespace issl
