// One stable radix pass of up to 8 bits (digit = key >> shift & mask) over 64-bit keys, shared by the off-target extraction (LSD sort of the site keys,
// issl_extract.hip) and the device-side index builder (one pass per slice, issl_build.hip).  Included by both; the
// kernels live in an anonymous namespace, one copy per translation unit.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace issl {
namespace {

constexpr uint32_t kSortItems = 16; // keys per thread and radix pass (4096 per 256-thread workgroup)

// ---- LSD radix sort of 64-bit keys, 8 bits per pass -----------------------------------------------------------

__global__ __launch_bounds__(256) void k_radix_hist(const uint64_t *__restrict__ keys, uint64_t n, uint32_t shift,
                                                    uint32_t *__restrict__ hist, uint32_t n_blocks, uint32_t mask)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t base = static_cast<uint64_t>(blockIdx.x) * (256 * kSortItems);
    for (uint32_t r = 0; r < kSortItems; ++r) {
        const uint64_t i = base + r * 256 + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    hist[static_cast<uint64_t>(threadIdx.x) * n_blocks + blockIdx.x] = h[threadIdx.x]; // digit-major for the scan
}

// Exclusive scan of the m = 256 * n_blocks histogram counters (digit-major), in three launches: sums of chunks of 4096
// counters, a scan of those sums by one workgroup, the chunks again with their bases.  (One workgroup walking over all
// 18 M counters of a 300 M-key pass took 10.9 ms -- three quarters of the pass.)
constexpr uint32_t kScanChunkWords = 4096;

__device__ __forceinline__ uint32_t block_scan_1024(uint32_t s, uint32_t *wave_sum /*[16]*/, uint32_t *total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t x = s;
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) wave_sum[wave] = x;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (uint32_t wv = 0; wv < 16; ++wv) {
        if (wv < wave) before += wave_sum[wv];
        all += wave_sum[wv];
    }
    if (total) *total = all;
    __syncthreads();
    return before + x - s; // exclusive prefix of this thread's s
}

__global__ __launch_bounds__(1024) void k_radix_scan_sums(const uint32_t *__restrict__ data, uint64_t m, uint32_t *__restrict__ sums)
{
    __shared__ uint32_t wave_sum[16];
    const uint64_t i0 = static_cast<uint64_t>(blockIdx.x) * kScanChunkWords + threadIdx.x * 4ull;
    uint32_t s = 0;
    for (uint32_t i = 0; i < 4; ++i) s += (i0 + i < m) ? data[i0 + i] : 0u;
    uint32_t total;
    (void)block_scan_1024(s, wave_sum, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void k_radix_scan_top(uint32_t *__restrict__ sums, uint32_t n)
{
    __shared__ uint32_t wave_sum[16];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n ? sums[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_scan_1024(v, wave_sum, &total);
        if (i < n) sums[i] = carry_s + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s += total;
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void k_radix_scan_apply(uint32_t *__restrict__ data, uint64_t m, const uint32_t *__restrict__ sums)
{
    __shared__ uint32_t wave_sum[16];
    const uint64_t i0 = static_cast<uint64_t>(blockIdx.x) * kScanChunkWords + threadIdx.x * 4ull;
    uint32_t val[4], s = 0;
    for (uint32_t i = 0; i < 4; ++i) {
        val[i] = (i0 + i < m) ? data[i0 + i] : 0u;
        s += val[i];
    }
    uint32_t run = sums[blockIdx.x] + block_scan_1024(s, wave_sum, nullptr);
    for (uint32_t i = 0; i < 4; ++i) {
        if (i0 + i < m) data[i0 + i] = run;
        run += val[i];
    }
}

// Words a histogram buffer needs for n_blocks key blocks: the counters + the chunk sums of their scan.
inline uint64_t radix_hist_words(uint32_t n_blocks)
{
    const uint64_t m = 256ull * n_blocks;
    return m + (m + kScanChunkWords - 1) / kScanChunkWords + 1;
}

inline void launch_radix_scan(uint32_t *hist, uint32_t n_blocks, hipStream_t stream)
{
    const uint64_t m = 256ull * n_blocks;
    if (m == 0) return;
    const uint32_t chunks = static_cast<uint32_t>((m + kScanChunkWords - 1) / kScanChunkWords);
    uint32_t *sums = hist + m;
    hipLaunchKernelGGL(k_radix_scan_sums, dim3(chunks), dim3(1024), 0, stream, hist, m, sums);
    hipLaunchKernelGGL(k_radix_scan_top, dim3(1), dim3(1024), 0, stream, sums, chunks);
    hipLaunchKernelGGL(k_radix_scan_apply, dim3(chunks), dim3(1024), 0, stream, hist, m, sums);
}

// What a pass writes for the key at index i: the key itself (sorting) ...
struct KeyItself {
    __device__ uint64_t operator()(uint64_t key, uint64_t) const { return key; }
};

// Stable scatter: inside a workgroup keys keep their order (round, wave, lane) among equal digits.
template <typename Payload>
__global__ __launch_bounds__(256) void k_radix_scatter(const uint64_t *__restrict__ in, uint64_t *__restrict__ out,
                                                       uint64_t n, uint32_t shift, const uint32_t *__restrict__ offsets,
                                                       uint32_t n_blocks, Payload payload, uint32_t mask)
{
    __shared__ uint32_t next[256];       // next free output slot of every digit for this workgroup
    __shared__ uint32_t wave_cnt[4][256]; // keys of every digit per wave in the current round
    const uint32_t wave = threadIdx.x >> 6;
    next[threadIdx.x] = offsets[static_cast<uint64_t>(threadIdx.x) * n_blocks + blockIdx.x];
    const uint64_t base = static_cast<uint64_t>(blockIdx.x) * (256 * kSortItems);
    for (uint32_t r = 0; r < kSortItems; ++r) {
        for (uint32_t w = 0; w < 4; ++w) wave_cnt[w][threadIdx.x] = 0;
        __syncthreads();
        const uint64_t i = base + r * 256 + threadIdx.x;
        const bool valid = i < n;
        const uint64_t key = valid ? in[i] : 0ull;
        const uint32_t d = static_cast<uint32_t>(key >> shift) & mask;
        uint64_t same = __ballot(valid); // lanes of this wave holding the same digit
#pragma unroll
        for (uint32_t b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            same &= bit ? bal : ~bal;
        }
        const uint32_t before = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(same >> 32),
                                                          __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(same), 0u));
        if (valid && before == 0) wave_cnt[wave][d] = static_cast<uint32_t>(__builtin_popcountll(same));
        __syncthreads();
        if (valid) {
            uint32_t at = next[d] + before;
            for (uint32_t w = 0; w < wave; ++w) at += wave_cnt[w][d];
            out[at] = payload(key, i);
        }
        __syncthreads();
        next[threadIdx.x] += wave_cnt[0][threadIdx.x] + wave_cnt[1][threadIdx.x] + wave_cnt[2][threadIdx.x] +
                             wave_cnt[3][threadIdx.x];
        __syncthreads();
    }
}


} // namespace
} // namespace issl
