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

} // namespace

int launch_build_entries(const uint64_t *d_sites, const uint32_t *d_occ, uint64_t n_sites, uint32_t slice_begin,
                         uint32_t slice_end, uint32_t slice_width, uint64_t *d_entries)
{
    if (slice_width != 8) { // one pass = one byte of the signature
        set_error("the device-side builder handles 8-bit slices only");
        return ISSL_E_UNSUPPORTED;
    }
    const uint32_t n_blocks = static_cast<uint32_t>((n_sites + 256ull * kSortItems - 1) / (256ull * kSortItems));
    uint32_t *d_hist = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_hist), 4ull * 256 * n_blocks);
    if (e != hipSuccess) {
        set_error(std::string("HIP error: ") + hipGetErrorString(e) + " (histograms of the device-side builder)");
        return ISSL_E_DEVICE;
    }
    for (uint32_t s = slice_begin; s < slice_end; ++s) {
        const uint32_t shift = slice_width * s;
        hipLaunchKernelGGL(k_radix_hist, dim3(n_blocks), dim3(256), 0, nullptr, d_sites, n_sites, shift, d_hist, n_blocks);
        hipLaunchKernelGGL(k_radix_scan, dim3(1), dim3(1024), 0, nullptr, d_hist, 256ull * n_blocks);
        hipLaunchKernelGGL(k_radix_scatter<SliceEntry>, dim3(n_blocks), dim3(256), 0, nullptr, d_sites,
                           d_entries + static_cast<uint64_t>(s - slice_begin) * n_sites, n_sites, shift, d_hist, n_blocks,
                           SliceEntry{d_occ});
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
