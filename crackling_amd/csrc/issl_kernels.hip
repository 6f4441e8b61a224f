// HIP kernels of the ISSL off-target scorer for gfx950 (MI355X).  Wave = 64 lanes.
//
// Pipeline for one batch of guides (reference: src/ISSL/isslScoreOfftargets.cpp:307-511):
//   bin_*      group the guides of the batch by (slice, slice value) = by index bucket       (A3)
//   scan       every bucket tile against every guide of that bucket: XOR, fold, popcount     (A4-A6)
//              -> append (guide, slice, position) keys of candidates within max_dist that were
//                 not already met in an earlier slice (replaces the seen-bitmap, A7)
//   group_*    counting sort of the keys by guide
//   replay     per guide: order its keys by (slice, position) = the reference's scan order, then
//              accumulate MIT / CFD sequentially with the reference's early exit            (A8-A11)
//
// The scan is the bandwidth/ALU-critical kernel.  It streams 4 B per candidate (the signature
// with the bucket's own slice removed, stored bit-sliced: 32 candidates per 32-bit plane) and keeps
// a tile of 2048 candidates in registers while the guide words of the bucket arrive through scalar
// loads, so one HBM read of a bucket tile serves every guide of the batch that falls into this bucket.
#include <hip/hip_runtime.h>

#include "issl_device.hpp"

namespace issl {

// CFD penalty tables (cfdPenalties.h:1-346) live in the code object's constant segment.
#define ISSL_CFD_QUAL __constant__
#include "cfd_tables.inc"

// ------------------------------------------------------------------------------------------------
// bit helpers
// ------------------------------------------------------------------------------------------------

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
__host__ __device__ inline uint32_t scan_word(uint64_t sig, uint32_t s)
{
    const uint32_t sh = 8u * s;
    const uint64_t low = sig & ((1ull << sh) - 1ull);
    const uint64_t high = (sig >> (sh + 8u)) << sh;
    const uint32_t rem = static_cast<uint32_t>(low | high);
    return gather_even16(rem) | (gather_even16(rem >> 1) << 16);
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
__host__ __device__ inline uint32_t plane_word(uint32_t r, uint32_t group)
{
    return ((r >> 2) * 64u + group) * 4u + (r & 3u);
}

// ------------------------------------------------------------------------------------------------
// upload: build the scan stream from sites + bucket entries
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_pack_scan_stream(ImageView v, uint32_t *__restrict__ scan_out,
                                                          uint32_t *__restrict__ error_flag)
{
    for (uint32_t t = blockIdx.x; t < v.n_tiles; t += gridDim.x) {
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
            if (pos < len) {
                const uint64_t e = v.entries[start + pos];
                const uint64_t id = e & 0xFFFFFFFFull;
                if (id < v.n_sites) w = scan_word(v.sites[id], slice);
                else atomicOr(error_flag, 1u);
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

void launch_pack_scan_stream(const ImageView &v, uint32_t *scan_out, uint32_t *error_flag, void *stream)
{
    if (v.n_tiles == 0) return;
    const uint32_t grid = v.n_tiles < 65536u ? v.n_tiles : 65536u;
    hipLaunchKernelGGL(k_pack_scan_stream, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), v,
                       scan_out, error_flag);
}

// ------------------------------------------------------------------------------------------------
// guide binning
// ------------------------------------------------------------------------------------------------

constexpr uint32_t kMaxBuckets = 2048;

__global__ __launch_bounds__(256) void k_guide_hist(const uint64_t *__restrict__ guides, uint32_t n,
                                                    uint32_t slice_width, uint32_t n_slices, uint32_t n_buckets,
                                                    uint32_t *__restrict__ ng)
{
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
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < n_buckets; b += 256)
        if (hist[b]) atomicAdd(&ng[b], hist[b]);
}

// Exclusive scan of one uint64 per thread over a 256-thread block.
__device__ inline uint64_t block_exclusive_scan(uint64_t v, uint64_t *lds /*[256]*/, uint64_t *total)
{
    const uint32_t t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        const uint64_t add = (t >= d) ? lds[t - d] : 0;
        __syncthreads();
        lds[t] += add;
        __syncthreads();
    }
    const uint64_t incl = lds[t];
    if (total) *total = lds[255];
    __syncthreads();
    return incl - v;
}

// One block: lay out the bucket-sorted guide arrays and the list of scan items.
__global__ __launch_bounds__(256) void k_plan(ImageView v, const uint32_t *__restrict__ ng,
                                              uint32_t *__restrict__ gstart, ScanItem *__restrict__ items,
                                              uint32_t cap_items, PlanInfo *__restrict__ plan)
{
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
            const uint32_t k = (g + kItemGuides - 1u) / kItemGuides;
            n_it += k;
            cost += static_cast<uint64_t>(nt) * (static_cast<uint64_t>(g) + static_cast<uint64_t>(k) * kTileFixedCost);
            cand += (v.bucket_start[b + 1] - v.bucket_start[b]) * g;
            wtiles += static_cast<uint64_t>(nt) * k;
        }
    }
    uint64_t tot_slots, tot_items, tot_cost, tot_cand, tot_tiles;
    uint64_t slot_at = block_exclusive_scan(slots, lds, &tot_slots);
    uint64_t item_at = block_exclusive_scan(n_it, lds, &tot_items);
    uint64_t cost_at = block_exclusive_scan(cost, lds, &tot_cost);
    (void)block_exclusive_scan(cand, lds, &tot_cand);
    (void)block_exclusive_scan(wtiles, lds, &tot_tiles);

    const bool overflow = tot_items > cap_items;
    for (uint32_t b = b0; b < b1; ++b) {
        const uint32_t g = ng[b];
        const uint32_t nt = v.tile_first[b + 1] - v.tile_first[b];
        gstart[b] = static_cast<uint32_t>(slot_at);
        if (g && nt && !overflow) {
            for (uint32_t done = 0; done < g; done += kItemGuides) {
                const uint32_t len = (g - done < kItemGuides) ? g - done : kItemGuides;
                ScanItem it;
                it.bucket = b;
                it.g0 = static_cast<uint32_t>(slot_at) + done;
                it.g1 = it.g0 + len;
                it.n_tiles = nt;
                it.cost0 = cost_at;
                items[item_at++] = it;
                cost_at += static_cast<uint64_t>(nt) * (len + kTileFixedCost);
            }
        }
        slot_at += (g + kGuideGroup - 1u) / kGuideGroup * kGuideGroup;
    }
    if (threadIdx.x == 255) {
        gstart[nb] = static_cast<uint32_t>(tot_slots);
        if (!overflow) {
            ScanItem end;
            end.bucket = 0; end.g0 = 0; end.g1 = 0; end.n_tiles = 0; end.cost0 = tot_cost;
            items[tot_items] = end;
        }
        plan->n_items = overflow ? 0u : static_cast<uint32_t>(tot_items);
        plan->total_cost = overflow ? 0ull : tot_cost;
        plan->candidates = tot_cand;
        // one range per tile while that keeps the ticket traffic low, else cost-balanced ranges
        const uint64_t max_ranges = static_cast<uint64_t>(kScanGridBlocks) * 4u * kScanRangesPerWave;
        plan->n_ranges = overflow ? 0u : static_cast<uint32_t>(tot_tiles < max_ranges ? tot_tiles : max_ranges);
        plan->error = overflow ? 2u : 0u;
    }
}

// Scatter every guide into its bucket's range of (gword, gidx), once per slice.
__global__ __launch_bounds__(256) void k_guide_scatter(const uint64_t *__restrict__ guides, uint32_t n,
                                                       uint32_t slice_width, uint32_t n_slices,
                                                       uint32_t n_buckets, const uint32_t *__restrict__ gstart,
                                                       uint32_t *__restrict__ gfill, uint32_t *__restrict__ gword,
                                                       uint32_t *__restrict__ gidx)
{
    __shared__ uint32_t hist[kMaxBuckets];
    __shared__ uint32_t base[kMaxBuckets];
    for (uint32_t b = threadIdx.x; b < n_buckets; b += 256) hist[b] = 0;
    __syncthreads();
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    const uint32_t low = (1u << slice_width) - 1u;
    uint64_t sig = 0;
    uint32_t rank[8];
    if (g < n) {
        sig = guides[g];
#pragma unroll
        for (uint32_t s = 0; s < 8; ++s) {
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
        for (uint32_t s = 0; s < 8; ++s) {
            if (s < n_slices) {
                const uint32_t key = static_cast<uint32_t>(sig >> (slice_width * s)) & low;
                const uint32_t b = (s << slice_width) + key;
                const uint32_t slot = gstart[b] + base[b] + rank[s];
                gword[slot] = scan_word(sig, s);
                gidx[slot] = g;
            }
        }
    }
}

void launch_bin_guides(const ImageView &v, const Workspace &ws, const uint64_t *d_guides, uint32_t n, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const uint32_t nb = v.n_buckets;
    (void)hipMemsetAsync(ws.ng, 0, sizeof(uint32_t) * nb, stream);
    (void)hipMemsetAsync(ws.gfill, 0, sizeof(uint32_t) * nb, stream);
    (void)hipMemsetAsync(ws.plan, 0, sizeof(PlanInfo), stream);
    (void)hipMemsetAsync(ws.counters, 0, sizeof(Counters), stream);
    (void)hipMemsetAsync(ws.gidx, 0xFF, sizeof(uint32_t) * ws.cap_gslots, stream);
    (void)hipMemsetAsync(ws.gword, 0, sizeof(uint32_t) * ws.cap_gslots, stream);
    (void)hipMemsetAsync(ws.gcount, 0, sizeof(uint32_t) * (static_cast<size_t>(n) + 1), stream);
    const uint32_t blocks = (n + 255u) / 256u;
    hipLaunchKernelGGL(k_guide_hist, dim3(blocks), dim3(256), 0, stream, d_guides, n, v.slice_width, v.n_slices, nb,
                       ws.ng);
    hipLaunchKernelGGL(k_plan, dim3(1), dim3(256), 0, stream, v, ws.ng, ws.gstart, ws.items,
                       static_cast<uint32_t>(ws.cap_items), ws.plan);
    hipLaunchKernelGGL(k_guide_scatter, dim3(blocks), dim3(256), 0, stream, d_guides, n, v.slice_width, v.n_slices,
                       nb, ws.gstart, ws.gfill, ws.gword, ws.gidx);
}

// ------------------------------------------------------------------------------------------------
// scan
// ------------------------------------------------------------------------------------------------

struct alignas(4 * kGuideGroup) GuideGroup {
    uint32_t w[kGuideGroup];
};

constexpr int kPlanes = 32; // planes per lane = VGPRs holding the lane's 32 candidates

// ---- raw records ------------------------------------------------------------------------------
// A candidate that the scan finds within max_dist of a guide is only NOTED by the scan kernel, as an
// 8-byte record (guide slot, tile, offset in tile), with plain stores into a chunk of the raw buffer
// that the wave owns -- no dependent load, no returning atomic on the hot path (one hit per ~50k
// comparisons is frequent enough that a latency chain per hit would dominate the kernel).
// k_verify then checks every record exactly, applies the first-matching-slice rule and appends the
// final keys.  Chunk = kChunkRecs slots of 8 bytes, slot 0 = number of used slots (header included).
__device__ __forceinline__ uint64_t raw_record(uint32_t gslot, uint32_t tile, uint32_t offset)
{
    return (static_cast<uint64_t>(gslot) << 37) | (static_cast<uint64_t>(tile) << 11) | offset;
}

struct RawWriter {
    uint64_t *chunk; // current chunk of this wave (wave-uniform)
    uint32_t fill;   // used slots of the current chunk, header included
};

__device__ __forceinline__ void raw_retire(const RawWriter &w, uint32_t lane)
{
    if (lane == 0) w.chunk[0] = w.fill;
}

__device__ __forceinline__ void raw_acquire(RawWriter &w, uint64_t *raw, uint32_t max_chunks, Counters *counters,
                                            uint32_t lane)
{
    uint32_t idx = 0;
    if (lane == 0) idx = atomicAdd(&counters->raw_chunks, 1u);
    idx = __builtin_amdgcn_readfirstlane(idx);
    if (idx >= max_chunks) { // buffer exhausted: write into the spare chunk, the host grows the buffer and re-runs
        idx = max_chunks;
        if (lane == 0) counters->raw_overflow = 1u;
    }
    w.chunk = raw + static_cast<uint64_t>(idx) * kChunkRecs;
    w.fill = 1;
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
template <int THR>
__device__ __forceinline__ uint32_t near_plane(const uint32_t (&c)[kPlanes], uint32_t gw, uint32_t thr)
{
    uint32_t m[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const uint32_t g0 = 0u - ((gw >> p) & 1u);
        const uint32_t g1 = 0u - ((gw >> (16 + p)) & 1u);
        m[p] = (c[p] ^ g0) | (c[16 + p] ^ g1);
    }
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
    // weight 4: 4 planes
    uint32_t a4;
    full_add(k4[0], k4[1], k4[2], a4, k8a);
    half_add(a4, k4[3], n2, k8b);
    if (THR == 0) return ~(k8a | k8b | n2 | n1 | n0);
    if (THR == 1) return ~(k8a | k8b | n2 | n1);
    if (THR == 2) return ~(k8a | k8b | n2 | (n1 & n0));
    if (THR == 3) return ~(k8a | k8b | n2);
    if (THR == 4) return ~(k8a | k8b | (n2 & (n1 | n0)));
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
    return ~gt;
}

// Cold block of the scan: the wave knows that SOME lane has a candidate within thr of the guide in slot
// gslot.  `ok` = this lane's plane of such candidates; candidate bit j of lane l sits at tile offset 32 l + j.
__device__ __forceinline__ void note_candidates(uint32_t ok, uint32_t gslot, uint32_t tile, uint32_t lane,
                                                RawWriter &w, uint64_t *raw, uint32_t max_chunks, Counters *counters)
{
    while (true) {
        const bool has = ok != 0u;
        const uint64_t who = __ballot(has);
        if (who == 0ull) break;
        const uint32_t n = static_cast<uint32_t>(__builtin_popcountll(who));
        if (w.fill + n > kChunkRecs) {
            raw_retire(w, lane);
            raw_acquire(w, raw, max_chunks, counters, lane);
        }
        if (has) {
            const uint32_t j = static_cast<uint32_t>(__builtin_ctz(ok));
            ok &= ok - 1u;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(who >> 32),
                                                            __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(who), 0u));
            w.chunk[w.fill + rank] = raw_record(gslot, tile, lane * 32u + j);
        }
        w.fill += n;
    }
}

// Scan kernel.  Every WAVE is an independent worker: it takes ranges of the cost axis (the first one
// by its global wave number, further ones from an atomic ticket), and for every tile of the range
// keeps the tile's 2048 candidates in registers (32 bit planes per lane) while the guide words of the
// item stream through scalar registers, 8 per scalar load.
// The streams the hot loop reads (scan planes, tile table, items, guide words, plan) are separate
// `const __restrict__` kernel arguments: they are never written by this kernel, which lets the
// compiler fetch the wave-uniform ones through the scalar cache.
template <int THR>
__global__ __launch_bounds__(256, 6) void k_scan(const uint32_t *__restrict__ scan_stream,
                                                 const uint32_t *__restrict__ tile_first,
                                                 const ScanItem *__restrict__ items,
                                                 const PlanInfo *__restrict__ plan,
                                                 const uint32_t *__restrict__ gword_stream, uint64_t *raw,
                                                 uint32_t max_chunks, Counters *counters, uint32_t thr)
{
    const uint32_t n_items = plan->n_items;
    const uint64_t total = plan->total_cost;
    const uint32_t n_ranges = plan->n_ranges;
    if (n_items == 0 || total == 0 || n_ranges == 0) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    uint32_t range = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    uint32_t tiles_done = 0;
    RawWriter w;
    w.chunk = raw + static_cast<uint64_t>(max_chunks) * kChunkRecs; // spare chunk until the first real one
    w.fill = kChunkRecs;                                             // "full": first note acquires a chunk
    bool own_chunk = false;

    while (range < n_ranges) {
        // Ranges cut the cost axis into equal parts; a tile belongs to the range that holds its
        // start cost.  No 128-bit intermediate: total < 2^50 and n_ranges <= 2^15.
        const uint64_t lo = total / n_ranges * range + (total % n_ranges) * range / n_ranges;
        const uint64_t hi = (range + 1 == n_ranges)
                                ? total
                                : total / n_ranges * (range + 1) + (total % n_ranges) * (range + 1) / n_ranges;
        if (hi > lo) {
            // item holding `lo`: last i with items[i].cost0 <= lo
            uint32_t a = 0, z = n_items;
            while (z - a > 1) {
                const uint32_t mid = (a + z) >> 1;
                if (items[mid].cost0 <= lo) a = mid; else z = mid;
            }
            uint32_t it = a;
            ScanItem cur = items[it];
            uint64_t tile_cost = static_cast<uint64_t>(cur.g1 - cur.g0) + kTileFixedCost;
            uint64_t k = (lo - cur.cost0 + tile_cost - 1) / tile_cost; // first tile starting at or after lo
            while (true) {
                if (k >= cur.n_tiles) {
                    ++it;
                    if (it >= n_items) break;
                    cur = items[it];
                    tile_cost = static_cast<uint64_t>(cur.g1 - cur.g0) + kTileFixedCost;
                    k = 0;
                }
                if (cur.cost0 + k * tile_cost >= hi) break;

                // ---- one tile: 2048 candidates of bucket cur.bucket, tile k ---------------------
                const uint32_t tile = tile_first[cur.bucket] + static_cast<uint32_t>(k);
                const uint4 *__restrict__ src =
                    reinterpret_cast<const uint4 *>(scan_stream + static_cast<uint64_t>(tile) * kTileCands);
                uint32_t c[kPlanes];
#pragma unroll
                for (int q = 0; q < kPlanes / 4; ++q) {
                    const uint4 t4 = src[q * 64 + lane];
                    c[4 * q + 0] = t4.x; c[4 * q + 1] = t4.y; c[4 * q + 2] = t4.z; c[4 * q + 3] = t4.w;
                }
                const uint32_t g_full = cur.g0 + ((cur.g1 - cur.g0) & ~(kGuideGroup - 1u));
                uint32_t g = cur.g0;
                for (; g < g_full; g += kGuideGroup) {
                    const GuideGroup gg = *reinterpret_cast<const GuideGroup *>(gword_stream + g);
                    uint32_t flagged = 0; // bit u: a lane has a candidate within thr of guide g+u
#pragma unroll
                    for (uint32_t u = 0; u < kGuideGroup; ++u) {
                        const uint32_t ok = near_plane<THR>(c, gg.w[u], thr);
                        if (__ballot(ok != 0u) != 0ull) flagged |= 1u << u;
                    }
                    while (flagged) { // ~4 % of the (guide, tile) pairs on random data
                        const uint32_t u = static_cast<uint32_t>(__builtin_ctz(flagged));
                        flagged &= flagged - 1u;
                        const uint32_t ok = near_plane<THR>(c, gword_stream[g + u], thr);
                        note_candidates(ok, g + u, tile, lane, w, raw, max_chunks, counters);
                        own_chunk = true;
                    }
                }
                for (; g < cur.g1; ++g) { // the (< 8) guides of the last, partial group
                    const uint32_t ok = near_plane<THR>(c, gword_stream[g], thr);
                    if (__ballot(ok != 0u) != 0ull) {
                        note_candidates(ok, g, tile, lane, w, raw, max_chunks, counters);
                        own_chunk = true;
                    }
                }
                ++tiles_done;
                ++k;
            }
        }
        // next range: the ticket counter continues after the statically assigned first round
        uint32_t ticket = 0;
        if (lane == 0) ticket = atomicAdd(&counters->next_range, 1u);
        range = n_waves + __builtin_amdgcn_readfirstlane(ticket);
    }
    if (own_chunk) raw_retire(w, lane);
    if (lane == 0 && tiles_done)
        atomicAdd(reinterpret_cast<unsigned long long *>(&counters->tiles), static_cast<unsigned long long>(tiles_done));
}

// Exact check of the raw records: one thread per record, one chunk per 128-thread workgroup.
__global__ __launch_bounds__(kChunkRecs) void k_verify(ImageView v, Workspace ws, const uint64_t *__restrict__ guides,
                                                       int max_dist)
{
    uint32_t n_chunks = ws.counters->raw_chunks;
    if (n_chunks > ws.cap_chunks) n_chunks = static_cast<uint32_t>(ws.cap_chunks);
    const uint64_t low = (1ull << v.slice_width) - 1ull;
    for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const uint64_t *recs = ws.raw + static_cast<uint64_t>(chunk) * kChunkRecs;
        const uint32_t used = static_cast<uint32_t>(recs[0]);
        const uint32_t t = threadIdx.x + 1u;
        if (t >= used || t >= kChunkRecs) continue;
        const uint64_t rec = recs[t];
        const uint32_t offset = static_cast<uint32_t>(rec) & (kTileCands - 1u);
        const uint32_t tile = static_cast<uint32_t>(rec >> 11) & 0x3FFFFFFu;
        const uint32_t gslot = static_cast<uint32_t>(rec >> 37);
        const uint32_t guide = ws.gidx[gslot];
        if (guide == kNoGuide) continue;
        // bucket of the tile: last b with tile_first[b] <= tile
        uint32_t lo = 0, hi = v.n_buckets;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (v.tile_first[mid] <= tile) lo = mid; else hi = mid;
        }
        const uint32_t bucket = lo;
        const uint32_t slice = bucket >> v.slice_width;
        const uint64_t start = v.bucket_start[bucket];
        const uint64_t len = v.bucket_start[bucket + 1] - start;
        const uint64_t pos = static_cast<uint64_t>(tile - v.tile_first[bucket]) * kTileCands + offset;
        if (pos >= len) continue; // zero padding of the bucket's last tile
        const uint64_t gsig = guides[guide];
        const uint64_t entry = v.entries[start + pos];
        const uint64_t ot = v.sites[entry & 0xFFFFFFFFull];
        if (__builtin_popcountll(mismatch_mask(gsig, ot)) > max_dist) continue; // exact, full signatures (:376-382)
        // First-matching-slice rule (equivalent of the seen bitmap, isslScoreOfftargets.cpp:385-390,463):
        // the site was already met iff an earlier slice of the XOR is all zero.
        const uint64_t x = gsig ^ ot;
        bool earlier = false;
        for (uint32_t j = 0; j < slice; ++j)
            if (((x >> (v.slice_width * j)) & low) == 0) earlier = true;
        if (earlier) continue;
        const uint32_t slot = atomicAdd(&ws.counters->n_hits, 1u);
        atomicAdd(&ws.gcount[guide], 1u);
        if (slot < ws.cap_hits)
            ws.hits[slot] = (static_cast<uint64_t>(guide) << 35) | (static_cast<uint64_t>(slice) << 32) | pos;
    }
}

template <int THR>
static void launch_scan_thr(const ImageView &v, const Workspace &ws, uint32_t thr, hipStream_t stream)
{
    hipLaunchKernelGGL(k_scan<THR>, dim3(kScanGridBlocks), dim3(256), 0, stream, v.scan, v.tile_first, ws.items, ws.plan,
                       ws.gword, ws.raw, static_cast<uint32_t>(ws.cap_chunks), ws.counters, thr);
}

void launch_scan(const ImageView &v, const Workspace &ws, const uint64_t *d_guides, uint32_t n, int max_dist,
                 void *stream_)
{
    (void)n;
    (void)d_guides;
    if (max_dist < 0) return; // isslScoreOfftargets.cpp:382: no distance satisfies 0 <= dist <= maxDist
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const uint32_t thr = max_dist > 31 ? 31u : static_cast<uint32_t>(max_dist);
    // ISSL_SCAN_GENERIC=1 forces the runtime-threshold build of the kernel (A/B and test aid).
    const char *generic_env = getenv("ISSL_SCAN_GENERIC");
    const bool generic = generic_env && generic_env[0] == '1';
    if (generic || thr > 4) launch_scan_thr<-1>(v, ws, thr, stream);
    else if (thr == 0) launch_scan_thr<0>(v, ws, thr, stream);
    else if (thr == 1) launch_scan_thr<1>(v, ws, thr, stream);
    else if (thr == 2) launch_scan_thr<2>(v, ws, thr, stream);
    else if (thr == 3) launch_scan_thr<3>(v, ws, thr, stream);
    else launch_scan_thr<4>(v, ws, thr, stream);
}

void launch_verify(const ImageView &v, const Workspace &ws, const uint64_t *d_guides, int max_dist, void *stream)
{
    if (max_dist < 0) return;
    hipLaunchKernelGGL(k_verify, dim3(4096), dim3(kChunkRecs), 0, static_cast<hipStream_t>(stream), v, ws, d_guides,
                       max_dist);
}

// ------------------------------------------------------------------------------------------------
// hit grouping: counting sort of the keys by guide
// ------------------------------------------------------------------------------------------------

constexpr uint32_t kScanChunk = 2048; // elements per block in the device-wide prefix sum

__global__ __launch_bounds__(256) void k_prefix_block_sums(const uint32_t *__restrict__ in, uint32_t n,
                                                           uint32_t *__restrict__ sums)
{
    __shared__ uint64_t lds[256];
    const uint32_t base = blockIdx.x * kScanChunk + threadIdx.x * 8u;
    uint64_t s = 0;
    for (uint32_t i = 0; i < 8; ++i)
        if (base + i < n) s += in[base + i];
    uint64_t total;
    (void)block_exclusive_scan(s, lds, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = static_cast<uint32_t>(total);
}

__global__ __launch_bounds__(256) void k_prefix_of_sums(uint32_t *__restrict__ sums, uint32_t n_blocks)
{
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
                                                      const uint32_t *__restrict__ sums, uint32_t *__restrict__ out)
{
    __shared__ uint64_t lds[256];
    const uint32_t base = blockIdx.x * kScanChunk + threadIdx.x * 8u;
    uint32_t val[8];
    uint64_t s = 0;
    for (uint32_t i = 0; i < 8; ++i) {
        val[i] = (base + i < n) ? in[base + i] : 0u;
        s += val[i];
    }
    uint64_t run = block_exclusive_scan(s, lds, nullptr) + sums[blockIdx.x];
    for (uint32_t i = 0; i < 8; ++i) {
        if (base + i < n) out[base + i] = static_cast<uint32_t>(run);
        run += val[i];
    }
}

__global__ __launch_bounds__(256) void k_group_scatter(const uint64_t *__restrict__ hits,
                                                       const Counters *__restrict__ counters, uint32_t cap,
                                                       const uint32_t *__restrict__ goff, uint32_t *__restrict__ gcur,
                                                       uint64_t *__restrict__ sorted)
{
    uint32_t n = counters->n_hits;
    if (n > cap) n = cap;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint64_t key = hits[i];
        const uint32_t guide = static_cast<uint32_t>(key >> 35);
        const uint32_t slot = goff[guide] + atomicAdd(&gcur[guide], 1u);
        sorted[slot] = key;
    }
}

void launch_group_hits(const Workspace &ws, uint32_t n, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const uint32_t m = n + 1; // gcount[n] = 0 so that goff[n] = total
    const uint32_t blocks = (m + kScanChunk - 1) / kScanChunk;
    (void)hipMemsetAsync(ws.gcur, 0, sizeof(uint32_t) * n, stream);
    hipLaunchKernelGGL(k_prefix_block_sums, dim3(blocks), dim3(256), 0, stream, ws.gcount, m, ws.blocksum);
    hipLaunchKernelGGL(k_prefix_of_sums, dim3(1), dim3(256), 0, stream, ws.blocksum, blocks);
    hipLaunchKernelGGL(k_prefix_apply, dim3(blocks), dim3(256), 0, stream, ws.gcount, m, ws.blocksum, ws.goff);
    hipLaunchKernelGGL(k_group_scatter, dim3(1024), dim3(256), 0, stream, ws.hits, ws.counters,
                       static_cast<uint32_t>(ws.cap_hits), ws.goff, ws.gcur, ws.sorted);
}

// ------------------------------------------------------------------------------------------------
// replay: ordered MIT/CFD accumulation, one wave per guide
// ------------------------------------------------------------------------------------------------

constexpr uint32_t kReplayLds = 2048; // keys sorted in LDS; longer lists are sorted in place in HBM

// Ascending sort of data[0..n) by one wave (block = 64 threads).  Bitonic network with every
// comparator ascending; comparators that touch an index >= n are no-ops (virtual +inf padding).
__device__ inline void wave_sort(uint64_t *data, uint32_t n)
{
    if (n < 2) return;
    uint32_t np = 1;
    while (np < n) np <<= 1;
    for (uint32_t k = 2; k <= np; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < (np >> 1); t += 64) {
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

// precalculatedScores[mask] with operator[] semantics: a missing mask contributes 0.0 (:394).
__device__ inline double mit_lookup(const ImageView &v, uint64_t mask)
{
    uint32_t lo = 0, hi = v.n_scores;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint64_t m = v.score_mask[mid];
        if (m == mask) return v.score_val[mid];
        if (m < mask) lo = mid + 1; else hi = mid;
    }
    return 0.0;
}

__global__ __launch_bounds__(64) void k_replay(ImageView v, Workspace ws, const uint64_t *__restrict__ guides,
                                               uint32_t n, ScoreParams p, double *__restrict__ out_mit,
                                               double *__restrict__ out_cfd, uint32_t *__restrict__ out_kept,
                                               issl_hit *__restrict__ out_hits)
{
    __shared__ uint64_t keys[kReplayLds];
    const bool calc_mit = p.method == ISSL_METHOD_MIT || p.method == ISSL_METHOD_AND || p.method == ISSL_METHOD_OR ||
                          p.method == ISSL_METHOD_AVG;
    const bool calc_cfd = p.method == ISSL_METHOD_CFD || p.method == ISSL_METHOD_AND || p.method == ISSL_METHOD_OR ||
                          p.method == ISSL_METHOD_AVG;
    const uint32_t lane = threadIdx.x;
    const uint64_t low = (1ull << v.slice_width) - 1ull;

    for (uint32_t g = blockIdx.x; g < n; g += gridDim.x) {
        const uint32_t h0 = ws.goff[g];
        const uint32_t h = ws.goff[g + 1] - h0;
        uint64_t *data;
        if (h <= kReplayLds) {
            for (uint32_t i = lane; i < h; i += 64) keys[i] = ws.sorted[h0 + i];
            data = keys;
        } else {
            data = ws.sorted + h0;
        }
        __syncthreads();
        wave_sort(data, h);
        __syncthreads();

        const uint64_t gsig = guides[g];
        double tot_mit = 0.0, tot_cfd = 0.0;
        uint32_t kept = 0;
        bool stop = false;
        for (uint32_t base = 0; base < h && !stop; base += 64) {
            const uint32_t idx = base + lane;
            double mit_term = 0.0, cfd_term = 0.0;
            if (idx < h) {
                const uint64_t key = data[idx];
                const uint32_t slice = static_cast<uint32_t>(key >> 32) & 7u;
                const uint32_t pos = static_cast<uint32_t>(key);
                const uint32_t bucket =
                    (slice << v.slice_width) + static_cast<uint32_t>((gsig >> (v.slice_width * slice)) & low);
                const uint64_t e = v.entries[v.bucket_start[bucket] + pos];
                const uint32_t id = static_cast<uint32_t>(e);
                const uint32_t occ = static_cast<uint32_t>(e >> 32);
                const uint64_t ot = v.sites[id];
                const uint64_t mm = mismatch_mask(gsig, ot);
                const int dist = __builtin_popcountll(mm);
                if (calc_mit && dist > 0) mit_term = mit_lookup(v, mm) * static_cast<double>(occ); // :394
                if (calc_cfd) {                                                                   // :399-460
                    double cfd;
                    if (dist == 0) {
                        cfd = 1.0;
                    } else {
                        cfd = issl_cfd_pam[10];
                        for (uint32_t q = 0; q < 20; ++q) {
                            const uint32_t gb = static_cast<uint32_t>(gsig >> (2 * q)) & 3u;
                            const uint32_t ob = static_cast<uint32_t>(ot >> (2 * q)) & 3u;
                            if (gb != ob) cfd *= issl_cfd_pos[(q << 4) | (gb << 2) | (ob ^ 3u)];
                        }
                    }
                    cfd_term = cfd * static_cast<double>(occ);
                }
                if (out_hits) {
                    issl_hit rec;
                    rec.guide = g; rec.slice = slice; rec.pos = pos; rec.id = id;
                    rec.dist = static_cast<uint32_t>(dist); rec.occ = occ;
                    out_hits[h0 + idx] = rec;
                }
            }
            const uint32_t cnt = (h - base < 64u) ? h - base : 64u;
            for (uint32_t l = 0; l < cnt; ++l) {
                // same order and same operations as the reference's running totals (:394,:460)
                tot_mit += bcast_f64(mit_term, static_cast<int>(l));
                tot_cfd += bcast_f64(cfd_term, static_cast<int>(l));
                ++kept;
                bool exit_now = false;                                            // :467-496
                if (p.method == ISSL_METHOD_AND) exit_now = tot_mit > p.maximum_sum && tot_cfd > p.maximum_sum;
                else if (p.method == ISSL_METHOD_OR) exit_now = tot_mit > p.maximum_sum || tot_cfd > p.maximum_sum;
                else if (p.method == ISSL_METHOD_AVG) exit_now = ((tot_mit + tot_cfd) / 2.0) > p.maximum_sum;
                else if (p.method == ISSL_METHOD_MIT) exit_now = tot_mit > p.maximum_sum;
                else if (p.method == ISSL_METHOD_CFD) exit_now = tot_cfd > p.maximum_sum;
                if (exit_now) { stop = true; break; }
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

void launch_replay(const ImageView &v, const Workspace &ws, const uint64_t *d_guides, uint32_t n,
                   const ScoreParams &p, double *d_mit, double *d_cfd, uint32_t *d_kept, issl_hit *d_hitrec,
                   void *stream)
{
    if (n == 0) return;
    const uint32_t grid = n < 65536u ? n : 65536u;
    hipLaunchKernelGGL(k_replay, dim3(grid), dim3(64), 0, static_cast<hipStream_t>(stream), v, ws, d_guides, n, p,
                       d_mit, d_cfd, d_kept, d_hitrec);
}

} // namespace issl
