// Index construction on the device (second half): the slice lists of isslCreateIndex.cpp:218-234 as one
// stable 8-bit radix pass per slice over the site signatures that are already in the HBM image.
//
// The reference appends every site id, in ascending order, to the list of the bucket its slice value selects, one
// slice after the other; the .issl stores, per slice, the 2^w lists back to back (bucket order), each entry
// occurrences << 32 | id (isslCreateIndex.cpp:226-231,275-285).  That is exactly a stable counting sort of the ids by
// slice value: histogram per 4096-site block, one exclusive scan over (value, block), stable scatter
// (issl_radix.hpp).  The host never holds the 40 B/site of entries, which is what limits the size of an index that
// can be built through host arrays (48 B/site).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

#include "issl_device.hpp"
#include "issl_host.hpp"
#include "issl_radix.hpp"

namespace issl {

namespace {

// ... or the slice-list entry of site i (the pass then IS the builder's inner loop for one slice)
struct SliceEntry {
    const uint32_t *occ;
    __device__ uint64_t operator()(uint64_t, uint64_t i) const { return (static_cast<uint64_t>(occ[i]) << 32) | i; }
};

// ---- sorted layouts: every bucket's candidates ordered by the byte of the successor slice ------------------------
// One slice at a time (16 B per site of temporary memory): key = (slice << 16 | own value << 8 | successor byte) << 40 |
// index of the list entry (the successor byte: succ_byte -- the next slice of five 8-bit ones, the next two / four of ten 4-bit / twenty 2-bit ones).  A slice's lists are already grouped by own byte; two stable 8-bit passes (successor byte,
// own byte) leave them grouped and order every bucket by successor byte, ties in list order.
constexpr uint32_t kKeyShift = 40;

__global__ __launch_bounds__(256) void k_sort_keys(const uint64_t *__restrict__ sites, const uint64_t *__restrict__ list,
                                                   const uint64_t *__restrict__ bucket_start, uint64_t n_sites,
                                                   uint32_t slice_width, uint32_t slice, uint64_t *__restrict__ keys,
                                                   uint32_t *__restrict__ flag)
{
    const uint64_t e0 = static_cast<uint64_t>(slice) * n_sites; // every slice lists every site once
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; i < n_sites; i += static_cast<uint64_t>(gridDim.x) * 256) {
        const uint64_t e = e0 + i;
        const uint64_t id = list[i] & 0xFFFFFFFFull;
        if (id >= n_sites) atomicOr(flag, 1u); // reported as a format error, like the pack kernel does
        const uint64_t sig = id < n_sites ? sites[id] : 0ull;
        const uint32_t own = static_cast<uint32_t>(sig >> (slice_width * slice)) & ((1u << slice_width) - 1u);
        const uint32_t succ = succ_byte(sig, slice, slice_width);
        keys[i] = (static_cast<uint64_t>((slice << 16) | (own << 8) | succ) << kKeyShift) | e;
        // The scoring order of the sorted layouts is (slice, site id): every list must be ascending by id, as the builder
        // writes it (isslCreateIndex.cpp:218-234).  (An entry in the wrong bucket is k_fill_maps' to flag.)
        if (i > 0 && e > bucket_start[(slice << slice_width) | own]) {
            const uint64_t before = list[i - 1] & 0xFFFFFFFFull;
            if (before == id) atomicOr(flag, 4u);     // the same site twice: not an index
            else if (before > id) atomicOr(flag, 2u); // valid, but only in list order
        }
    }
}

__global__ __launch_bounds__(256) void k_fill_maps(const uint64_t *__restrict__ keys, const uint64_t *__restrict__ bucket_start,
                                                   const uint32_t *__restrict__ tile_first, const uint64_t *__restrict__ list, uint64_t n_sites, uint32_t slice,
                                                   uint32_t slice_width, const uint64_t *__restrict__ sites, StreamRec *__restrict__ srec,
                                                   uint32_t *__restrict__ sid, uint32_t *__restrict__ site_occ,
                                                   uint32_t *__restrict__ flag)
{
    const uint64_t t0 = static_cast<uint64_t>(slice) * n_sites;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; i < n_sites; i += static_cast<uint64_t>(gridDim.x) * 256) {
        const uint64_t key = keys[i];
        const uint64_t e = key & ((1ull << kKeyShift) - 1ull);
        const uint32_t so = static_cast<uint32_t>(key >> (kKeyShift + 8)); // slice << 8 | own value
        const uint32_t bucket = ((so >> 8) << slice_width) | (so & 0xFFu);
        // An entry listed in a bucket its signature does not belong to: not an index (see k_pack_scan_stream).
        if (e < bucket_start[bucket] || e >= bucket_start[bucket + 1]) { atomicOr(flag, 4u); continue; }
        const uint32_t p = static_cast<uint32_t>(e - bucket_start[bucket]);
        const uint64_t entry = list[e - t0];
        const uint32_t id = static_cast<uint32_t>(entry & 0xFFFFFFFFull);
        if (id >= n_sites) continue; // flagged by k_sort_keys: the upload fails with a format error
        const uint64_t occ = entry >> 32;
        // one count per site: the reference takes the count of the entry it meets first (:348); an index whose lists
        // disagree about a site keeps its list-order layout
        if (slice == 0) site_occ[id] = static_cast<uint32_t>(occ);
        else if (site_occ[id] != static_cast<uint32_t>(occ)) atomicOr(flag, 2u);
        // t0 + i = bucket_start[bucket] + stream position inside the bucket (the buckets keep their places); the maps are
        // indexed like the scan stream itself, where every bucket starts on a tile
        const uint64_t at = static_cast<uint64_t>(tile_first[bucket]) * kTileCands + (t0 + i - bucket_start[bucket]);
        if (srec) {
            StreamRec r;
            r.sig = (sites[id] & kSigMask) | ((occ < kOccSaturated ? occ : kOccSaturated) << 40);
            r.id = id; r.pos = p;
            srec[at] = r;
        } else {
            sid[at] = id;
        }
    }
}

__global__ __launch_bounds__(256) void k_sub_start(const uint64_t *__restrict__ keys, const uint64_t *__restrict__ bucket_start,
                                                   uint64_t n_sites, uint32_t slice, uint32_t buckets_per_slice,
                                                   uint32_t *__restrict__ sub_start)
{
    const uint32_t idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= buckets_per_slice * 257u) return;
    const uint32_t bucket = slice * buckets_per_slice + idx / 257u, w = idx % 257u;
    const uint64_t t0 = static_cast<uint64_t>(slice) * n_sites; // keys[] holds this slice only
    uint64_t lo = bucket_start[bucket] - t0, hi = bucket_start[bucket + 1] - t0;
    if (bucket_start[bucket] < t0 || hi > n_sites || lo > hi) { lo = 0; hi = 0; } // (inconsistent sizes: flagged by k_fill_maps)
    const uint64_t first = lo;
    while (lo < hi) { // first stream position of the bucket whose successor byte is >= w
        const uint64_t mid = (lo + hi) >> 1;
        if (((keys[mid] >> kKeyShift) & 0xFFull) < w) lo = mid + 1; else hi = mid;
    }
    sub_start[static_cast<uint64_t>(bucket) * 257u + w] = static_cast<uint32_t>(lo - first);
}

} // namespace

// Sorted layouts, last step of their construction: a 24-bit copy of every site's occurrence count into the 24 bits of its
// signature word that the 20-mer leaves free (the same packing as StreamRec::sig).  Whoever goes from a site id to the site
// -- k_verify on the compact layout, the many-hit replay on both -- then has signature and count in ONE random read instead
// of two; counts from kOccSaturated on are still looked up in site_occ.  Every reader of `sites` masks the upper bits.
__global__ __launch_bounds__(256) void k_tag_sites(uint64_t *__restrict__ sites, const uint32_t *__restrict__ site_occ, uint64_t n_sites)
{
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; i < n_sites; i += static_cast<uint64_t>(gridDim.x) * 256) {
        const uint64_t occ = site_occ[i];
        sites[i] = (sites[i] & kSigMask) | ((occ < kOccSaturated ? occ : kOccSaturated) << 40);
    }
}

void launch_tag_sites(uint64_t *d_sites, const uint32_t *d_site_occ, uint64_t n_sites)
{
    if (n_sites == 0) return;
    const uint32_t grid = static_cast<uint32_t>(std::min<uint64_t>((n_sites + 255) / 256, 1u << 16));
    hipLaunchKernelGGL(k_tag_sites, dim3(grid), dim3(256), 0, nullptr, d_sites, d_site_occ, n_sites);
}

int SortTemp::alloc(uint64_t n_sites, void *borrowed_keys)
{
    release();
    const uint32_t n_blocks = static_cast<uint32_t>((n_sites + 256ull * kSortItems - 1) / (256ull * kSortItems));
    if (borrowed_keys) { keys = static_cast<uint64_t *>(borrowed_keys); keys_borrowed = true; }
    else if (hipMalloc(reinterpret_cast<void **>(&keys), 8 * std::max<uint64_t>(n_sites, 1)) != hipSuccess) keys = nullptr;
    if (keys && hipMalloc(reinterpret_cast<void **>(&tmp), 8 * std::max<uint64_t>(n_sites, 1)) != hipSuccess) tmp = nullptr;
    if (tmp && hipMalloc(reinterpret_cast<void **>(&hist), 4ull * radix_hist_words(std::max(n_blocks, 1u))) != hipSuccess) hist = nullptr;
    if (!hist) {
        (void)hipGetLastError();
        release();
        return kSortNoRoom;
    }
    return ISSL_OK;
}

void SortTemp::release()
{
    if (keys && !keys_borrowed) (void)hipFree(keys);
    keys_borrowed = false;
    if (tmp) (void)hipFree(tmp);
    if (hist) (void)hipFree(hist);
    keys = tmp = nullptr;
    hist = nullptr;
}

int launch_sort_slice(SortTemp &t, const uint64_t *d_sites, const uint64_t *d_list, const uint64_t *d_bucket_start,
                      const uint32_t *d_tile_first, uint64_t n_sites, uint32_t n_slices, uint32_t n_buckets, uint32_t slice_width, uint32_t slice, uint32_t *d_sub_start,
                      StreamRec *d_srec, uint32_t *d_sid, uint32_t *d_site_occ, uint32_t *d_flag)
{
    if (!((slice_width == 8 && n_slices == 5) || (slice_width == 4 && n_slices == 10) || (slice_width == 2 && n_slices == 20)) || n_buckets != (n_slices << slice_width) ||
        n_sites >= (1ull << 32) || slice >= n_slices) {
        set_error("the sorted layout handles 20 positions in 8-, 4- or 2-bit slices and up to 2^32 - 1 sites");
        return ISSL_E_UNSUPPORTED;
    }
    const uint32_t per_slice = n_buckets / n_slices;
    if (n_sites == 0) {
        (void)hipMemsetAsync(d_sub_start + static_cast<uint64_t>(slice) * per_slice * 257u, 0, 4ull * per_slice * 257u, nullptr);
        return ISSL_OK;
    }
    const uint32_t n_blocks = static_cast<uint32_t>((n_sites + 256ull * kSortItems - 1) / (256ull * kSortItems));
    const uint32_t grid = static_cast<uint32_t>(std::min<uint64_t>((n_sites + 255) / 256, 1u << 20));
    hipLaunchKernelGGL(k_sort_keys, dim3(grid), dim3(256), 0, nullptr, d_sites, d_list, d_bucket_start, n_sites, slice_width, slice,
                       t.keys, d_flag);
    uint64_t *src = t.keys, *dst = t.tmp;
    for (uint32_t pass = 0; pass < 2; ++pass) { // successor byte, then own byte
        const uint32_t shift = kKeyShift + 8 * pass;
        hipLaunchKernelGGL(k_radix_hist, dim3(n_blocks), dim3(256), 0, nullptr, src, n_sites, shift, t.hist, n_blocks, 0xFFu);
        launch_radix_scan(t.hist, n_blocks, nullptr);
        hipLaunchKernelGGL(k_radix_scatter<KeyItself>, dim3(n_blocks), dim3(256), 0, nullptr, src, dst, n_sites, shift, t.hist,
                           n_blocks, KeyItself{}, 0xFFu);
        std::swap(src, dst);
    }
    hipLaunchKernelGGL(k_fill_maps, dim3(grid), dim3(256), 0, nullptr, src, d_bucket_start, d_tile_first, d_list, n_sites, slice, slice_width,
                       d_sites, d_srec, d_sid, d_site_occ, d_flag);
    hipLaunchKernelGGL(k_sub_start, dim3((per_slice * 257u + 255u) / 256u), dim3(256), 0, nullptr, src, d_bucket_start,
                       n_sites, slice, per_slice, d_sub_start);
    if (hipGetLastError() != hipSuccess) {
        set_error("HIP error launching the sorted-layout kernels");
        return ISSL_E_DEVICE;
    }
    return ISSL_OK;
}

int finish_sort(uint32_t *d_flag)
{
    hipError_t e = hipDeviceSynchronize();
    uint32_t flags = 0;
    if (e == hipSuccess) e = hipMemcpy(&flags, d_flag, 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        set_error(std::string("HIP error: ") + hipGetErrorString(e) + " (sorted layout)");
        return ISSL_E_DEVICE;
    }
    if (flags & 1u) {
        set_error("Error reading index: a slice entry refers to an off-target id beyond the site table");
        return ISSL_E_FORMAT;
    }
    if (flags & 4u) {
        set_error("Error reading index: a slice list holds an off-target in a bucket its signature does not select, or twice");
        return ISSL_E_FORMAT;
    }
    if (flags & 2u) { // cleared for the pack kernel, which shares the flag word
        (void)hipMemset(d_flag, 0, 4);
        return kSortNeedsListOrder;
    }
    return ISSL_OK;
}

namespace {
// (sizes[nb]: set when a signature carries bits above the n_slices * slice_width the geometry gives it -- the sorted layouts keep
// a site's count in those bits of the site table, k_tag_sites)
__global__ __launch_bounds__(256) void k_bucket_sizes(const uint64_t *__restrict__ sites, uint64_t n_sites, uint32_t slice_width,
                                                      uint32_t n_slices, unsigned long long *__restrict__ sizes)
{
    __shared__ uint32_t hist[kMaxSlices << 8]; // 20 KiB: widths up to 8 bits, up to 20 slices
    const uint32_t nb = n_slices << slice_width, low = (1u << slice_width) - 1u;
    // a workgroup's share stays below 2^32 sites: 32-bit counters in LDS
    for (uint32_t b = threadIdx.x; b < nb; b += 256) hist[b] = 0;
    __syncthreads();
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; i < n_sites; i += static_cast<uint64_t>(gridDim.x) * 256) {
        const uint64_t sig = sites[i];
        for (uint32_t s = 0; s < n_slices; ++s) atomicAdd(&hist[(s << slice_width) + (static_cast<uint32_t>(sig >> (slice_width * s)) & low)], 1u);
        if (n_slices * slice_width < 64u && (sig >> (n_slices * slice_width)) != 0ull) sizes[nb] = 1ull; // (every writer stores the same value)
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += 256)
        if (hist[b]) atomicAdd(&sizes[b], static_cast<unsigned long long>(hist[b]));
}
} // namespace

int launch_bucket_sizes(const uint64_t *d_sites, uint64_t n_sites, uint32_t slice_width, uint32_t n_slices, uint64_t *h_sizes)
{
    if (slice_width == 0 || slice_width > 8 || n_slices == 0 || n_slices > kMaxSlices) {
        set_error("bucket sizes on the device: slices of 1..8 bits, up to 20 of them");
        return ISSL_E_UNSUPPORTED;
    }
    const uint32_t nb = n_slices << slice_width;
    unsigned long long *d_sizes = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_sizes), 8ull * (nb + 1u));
    if (e == hipSuccess) e = hipMemset(d_sizes, 0, 8ull * (nb + 1u));
    unsigned long long stray = 0ull;
    if (e == hipSuccess) {
        const uint32_t grid = static_cast<uint32_t>(std::min<uint64_t>((n_sites + 255) / 256, 4096));
        hipLaunchKernelGGL(k_bucket_sizes, dim3(std::max(grid, 1u)), dim3(256), 0, nullptr, d_sites, n_sites, slice_width, n_slices, d_sizes);
        e = hipMemcpy(h_sizes, d_sizes, 8ull * nb, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(&stray, d_sizes + nb, 8, hipMemcpyDeviceToHost);
    }
    if (d_sizes) (void)hipFree(d_sizes);
    if (e == hipSuccess && stray) {
        set_error("site table on the device: a signature carries bits above the " + std::to_string(n_slices * slice_width) +
                  " its geometry gives it (2 bits per position, position 0 in the lowest)");
        return ISSL_E_ARG;
    }
    if (e != hipSuccess) {
        set_error(std::string("HIP error: ") + hipGetErrorString(e) + " (bucket sizes of a site table on the device)");
        return ISSL_E_DEVICE;
    }
    return ISSL_OK;
}

int launch_build_entries(const uint64_t *d_sites, const uint32_t *d_occ, uint64_t n_sites, uint32_t slice_begin,
                         uint32_t slice_end, uint32_t slice_width, uint64_t *d_entries)
{
    if (slice_width == 0 || slice_width > 8) { // one pass = one slice value of up to 8 bits
        set_error("the device-side builder handles slices of up to 8 bits");
        return ISSL_E_UNSUPPORTED;
    }
    const uint32_t mask = (1u << slice_width) - 1u;
    const uint32_t n_blocks = static_cast<uint32_t>((n_sites + 256ull * kSortItems - 1) / (256ull * kSortItems));
    uint32_t *d_hist = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_hist), 4ull * radix_hist_words(n_blocks));
    if (e != hipSuccess) {
        set_error(std::string("HIP error: ") + hipGetErrorString(e) + " (histograms of the device-side builder)");
        return ISSL_E_DEVICE;
    }
    for (uint32_t s = slice_begin; s < slice_end; ++s) {
        const uint32_t shift = slice_width * s;
        hipLaunchKernelGGL(k_radix_hist, dim3(n_blocks), dim3(256), 0, nullptr, d_sites, n_sites, shift, d_hist, n_blocks, mask);
        launch_radix_scan(d_hist, n_blocks, nullptr);
        hipLaunchKernelGGL(k_radix_scatter<SliceEntry>, dim3(n_blocks), dim3(256), 0, nullptr, d_sites,
                           d_entries + static_cast<uint64_t>(s - slice_begin) * n_sites, n_sites, shift, d_hist, n_blocks,
                           SliceEntry{d_occ}, mask);
    }
    e = hipDeviceSynchronize();
    (void)hipFree(d_hist);
    if (e != hipSuccess) {
        set_error(std::string("HIP error in the device-side builder: ") + hipGetErrorString(e));
        return ISSL_E_DEVICE;
    }
    return ISSL_OK;
}

} // namespace issl
