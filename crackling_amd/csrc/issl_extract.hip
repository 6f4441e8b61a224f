// Off-target site extraction on the GPU: the step that feeds the index builder.
// Counterpart of /root/reference/src/crackling/utils/extractOfftargets.py:
//   :23-24   forward pattern [ACG][ACGT]{19}[ACGT][AG]G, reverse pattern C[CT][ACGT][ACGT]{19}[TGC], both as lookaheads
//            (every position is tried, matches overlap)
//   :97-110  a match contributes the first 20 characters of its 23 -- as they are (forward) or reverse-complemented
//            (reverse pattern; Helpers.py:7-10)
//   :112-191 all sites, one per line, sorted as text, duplicates kept
//
// Host: FASTA records -> one upper-cased byte string with '\n' between records (no match can span a separator).
// Device: k_match_count / k_match_emit find the matches and write each site as a 40-bit key whose numeric order is
// the text order (base 0 in the two most significant bits); an LSD radix sort (5 passes of 8 bits) orders the keys;
// k_keys_to_text expands them to "<20 chars>\n".
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "issl_host.hpp"
#include "issl_radix.hpp"

namespace issl {

namespace {

#define EX_HIP_TRY(expr)                                                                           \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            set_error(std::string("HIP error: ") + hipGetErrorString(e_) + " at " #expr);          \
            return ISSL_E_DEVICE;                                                                  \
        }                                                                                          \
    } while (0)

constexpr uint32_t kPosPerBlock = 4096; // text positions per 256-thread workgroup

// 0..3 for A C G T, 4 for anything else
__device__ __forceinline__ uint32_t base_code(uint8_t c)
{
    return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
}

// Matches starting at position i: bit 0 = forward pattern, bit 1 = reverse pattern; keys of the two sites.
__device__ __forceinline__ uint32_t match_at(const uint8_t *__restrict__ s, uint64_t i, uint64_t len, uint64_t &key_fwd,
                                             uint64_t &key_rev)
{
    if (i + 23 > len) return 0;
    uint32_t code[23];
    bool body = true; // characters 1..20 are [ACGT] in both patterns
#pragma unroll
    for (int k = 0; k < 23; ++k) code[k] = base_code(s[i + k]);
#pragma unroll
    for (int k = 1; k <= 20; ++k) body = body && code[k] < 4u;
    if (!body) return 0;
    const bool fwd = code[0] < 3u && (code[21] == 0u || code[21] == 2u) && code[22] == 2u;
    const bool rev = code[0] == 1u && (code[1] == 1u || code[1] == 3u) && code[21] < 4u &&
                     (code[22] == 3u || code[22] == 2u || code[22] == 1u);
    if (!fwd && !rev) return 0;
    uint64_t kf = 0, kr = 0;
#pragma unroll
    for (int p = 0; p < 20; ++p) {
        kf |= static_cast<uint64_t>(code[p]) << (2 * (19 - p));   // text order: base 0 most significant
        kr |= static_cast<uint64_t>(3u - code[p]) << (2 * p);      // reverse complement of the same 20 characters
    }
    key_fwd = kf;
    key_rev = kr;
    return (fwd ? 1u : 0u) | (rev ? 2u : 0u);
}

__global__ __launch_bounds__(256) void k_match_count(const uint8_t *__restrict__ s, uint64_t len,
                                                     unsigned long long *__restrict__ total)
{
    __shared__ uint32_t wave_cnt[4];
    const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kPosPerBlock;
    uint32_t cnt = 0;
    for (uint32_t k = threadIdx.x; k < kPosPerBlock; k += 256) {
        uint64_t a, b;
        const uint32_t m = match_at(s, base + k, len, a, b);
        cnt += (m & 1u) + (m >> 1);
    }
    for (int d = 32; d > 0; d >>= 1) cnt += __shfl_down(cnt, d, 64);
    if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        if (t) atomicAdd(total, static_cast<unsigned long long>(t));
    }
}

// Second pass: every workgroup reserves room for its matches with ONE atomic, then its threads write keys at
// wave-prefix offsets (the order of the keys is irrelevant, they are sorted afterwards).
__global__ __launch_bounds__(256) void k_match_emit(const uint8_t *__restrict__ s, uint64_t len,
                                                    unsigned long long *__restrict__ cursor, uint64_t *__restrict__ keys,
                                                    uint64_t cap)
{
    __shared__ uint32_t wave_cnt[4];
    __shared__ unsigned long long block_base;
    const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kPosPerBlock;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t cnt = 0;
    for (uint32_t k = threadIdx.x; k < kPosPerBlock; k += 256) {
        uint64_t a, b;
        const uint32_t m = match_at(s, base + k, len, a, b);
        cnt += (m & 1u) + (m >> 1);
    }
    uint32_t incl = cnt; // inclusive scan inside the wave
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(incl, d, 64);
        if (lane >= d) incl += y;
    }
    if (lane == 63) wave_cnt[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        block_base = t ? atomicAdd(cursor, static_cast<unsigned long long>(t)) : 0ull;
    }
    __syncthreads();
    uint64_t at = block_base + (incl - cnt);
    for (uint32_t w = 0; w < wave; ++w) at += wave_cnt[w];
    for (uint32_t k = threadIdx.x; k < kPosPerBlock; k += 256) {
        uint64_t a = 0, b = 0;
        const uint32_t m = match_at(s, base + k, len, a, b);
        if ((m & 1u) && at < cap) keys[at++] = a;
        if ((m & 2u) && at < cap) keys[at++] = b;
    }
}

__global__ __launch_bounds__(256) void k_keys_to_text(const uint64_t *__restrict__ keys, uint64_t n, char *__restrict__ text)
{
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = keys[i];
    char *dst = text + i * 21;
#pragma unroll
    for (int p = 0; p < 20; ++p) dst[p] = "ACGT"[(key >> (2 * (19 - p))) & 3u];
    dst[20] = '\n';
}

// Sort d_keys[0..n) ascending on the low `bits` bits; d_tmp has the same size.  Result in d_keys.
int radix_sort(uint64_t *d_keys, uint64_t *d_tmp, uint64_t n, uint32_t bits)
{
    if (n < 2) return ISSL_OK;
    const uint32_t n_blocks = static_cast<uint32_t>((n + 256ull * kSortItems - 1) / (256ull * kSortItems));
    uint32_t *d_hist = nullptr;
    EX_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_hist), 4ull * radix_hist_words(n_blocks)));
    uint64_t *src = d_keys, *dst = d_tmp;
    for (uint32_t shift = 0; shift < bits; shift += 8) {
        hipLaunchKernelGGL(k_radix_hist, dim3(n_blocks), dim3(256), 0, nullptr, src, n, shift, d_hist, n_blocks, 0xFFu);
        launch_radix_scan(d_hist, n_blocks, nullptr);
        hipLaunchKernelGGL(k_radix_scatter<KeyItself>, dim3(n_blocks), dim3(256), 0, nullptr, src, dst, n, shift, d_hist,
                           n_blocks, KeyItself{}, 0xFFu);
        std::swap(src, dst);
    }
    hipError_t e = hipDeviceSynchronize();
    (void)hipFree(d_hist);
    if (e != hipSuccess) {
        set_error(std::string("HIP error in radix sort: ") + hipGetErrorString(e));
        return ISSL_E_DEVICE;
    }
    if (src != d_keys) EX_HIP_TRY(hipMemcpy(d_keys, src, 8 * n, hipMemcpyDeviceToDevice));
    return ISSL_OK;
}

// FASTA bytes -> upper-cased sequence text with '\n' after every record (extractOfftargets.py:27-61,72-88).
// Lines [begin, end) of one piece of a file; `begin` is a line start.  A header line closes the record before it.
static void parse_fasta_lines(const char *fasta, size_t begin, size_t end, std::string &seq)
{
    size_t p = begin;
    while (p < end) {
        size_t e = p;
        while (e < end && fasta[e] != '\n') ++e;
        size_t a = p, b = e;
        while (a < b && std::isspace(static_cast<unsigned char>(fasta[a]))) ++a;
        while (b > a && std::isspace(static_cast<unsigned char>(fasta[b - 1]))) --b;
        if (b > a && fasta[a] == '>') {
            if (seq.empty() || seq.back() != '\n') seq.push_back('\n'); // (a piece's leading separator is settled when it is joined)
        } else {
            for (size_t k = a; k < b; ++k) seq.push_back(static_cast<char>(std::toupper(static_cast<unsigned char>(fasta[k]))));
        }
        p = e + 1;
    }
}

// The one host pass of the extraction, on up to 16 threads: the file is cut at line starts, every piece is parsed on
// its own, and the pieces are joined with the sequential rule for record separators (none at the very start, never two
// in a row), so the result is byte-for-byte what one thread produces.  (One thread manages ~0.3 GB/s: 10 s for a human
// genome, against ~0.3 s for everything that follows on the GPU.)
void append_records(const char *fasta, size_t len, std::string &seq)
{
    const size_t want = std::min<size_t>({16, std::max(1u, std::thread::hardware_concurrency()), len / (size_t(4) << 20) + 1});
    std::vector<size_t> cut(want + 1, len);
    cut[0] = 0;
    for (size_t t = 1; t < want; ++t) {
        size_t at = std::max(cut[t - 1], len / want * t);
        while (at < len && fasta[at] != '\n') ++at;  // the piece starts behind the next line end
        cut[t] = at < len ? at + 1 : len;
    }
    std::vector<std::string> piece(want);
    {
        std::vector<std::thread> pool;
        for (size_t t = 1; t < want; ++t)
            pool.emplace_back([&, t] { piece[t].reserve(cut[t + 1] - cut[t]); parse_fasta_lines(fasta, cut[t], cut[t + 1], piece[t]); });
        piece[0].reserve(cut[1] - cut[0]);
        parse_fasta_lines(fasta, cut[0], cut[1], piece[0]);
        for (auto &th : pool) th.join();
    }
    size_t total = seq.size() + 1;
    for (const auto &pc : piece) total += pc.size();
    seq.reserve(total);
    for (const auto &pc : piece) {
        size_t from = 0;
        if (!pc.empty() && pc[0] == '\n' && (seq.empty() || seq.back() == '\n')) from = 1;
        seq.append(pc, from, std::string::npos);
    }
    if (!seq.empty() && seq.back() != '\n') seq.push_back('\n');
}

// seq (host) -> sorted site text (host, malloc'd).
int extract_sorted_text(const std::string &seq, int device, char **out_text, size_t *out_len, uint64_t *n_sites)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        set_error("no HIP device available: the extraction has no CPU fallback");
        return ISSL_E_DEVICE;
    }
    if (device < 0 || device >= count) {
        set_error("device out of range");
        return ISSL_E_ARG;
    }
    EX_HIP_TRY(hipSetDevice(device));
    *out_text = nullptr;
    *out_len = 0;
    *n_sites = 0;
    const uint64_t len = seq.size();
    if (len < 23) {
        *out_text = static_cast<char *>(std::malloc(1));
        return ISSL_OK;
    }
    // device buffers are released on every path out of this function
    struct DevBuf {
        void *p = nullptr;
        ~DevBuf() { release(); }
        void release() { if (p) (void)hipFree(p); p = nullptr; }
    } seq_buf, ctr_buf, keys_buf, tmp_buf, text_buf;
    EX_HIP_TRY(hipMalloc(&seq_buf.p, len));
    EX_HIP_TRY(hipMalloc(&ctr_buf.p, 16));
    uint8_t *d_seq = static_cast<uint8_t *>(seq_buf.p);
    unsigned long long *d_ctr = static_cast<unsigned long long *>(ctr_buf.p);
    EX_HIP_TRY(hipMemcpy(d_seq, seq.data(), len, hipMemcpyHostToDevice));
    EX_HIP_TRY(hipMemset(d_ctr, 0, 16));
    const uint32_t blocks = static_cast<uint32_t>((len + kPosPerBlock - 1) / kPosPerBlock);
    hipLaunchKernelGGL(k_match_count, dim3(blocks), dim3(256), 0, nullptr, d_seq, len, d_ctr);
    unsigned long long total = 0;
    EX_HIP_TRY(hipMemcpy(&total, d_ctr, 8, hipMemcpyDeviceToHost));
    if (total > 0xFFFFFFFFull) { // the radix passes count and place with 32-bit offsets (issl_radix.hpp)
        set_error("more than 2^32 - 1 sites in one extraction (" + std::to_string(total) + "): split the input");
        return ISSL_E_UNSUPPORTED;
    }
    if (total == 0) {
        *out_text = static_cast<char *>(std::malloc(1));
        return ISSL_OK;
    }
    EX_HIP_TRY(hipMalloc(&keys_buf.p, 8 * total));
    EX_HIP_TRY(hipMalloc(&tmp_buf.p, 8 * total));
    uint64_t *d_keys = static_cast<uint64_t *>(keys_buf.p), *d_tmp = static_cast<uint64_t *>(tmp_buf.p);
    hipLaunchKernelGGL(k_match_emit, dim3(blocks), dim3(256), 0, nullptr, d_seq, len, d_ctr + 1, d_keys, total);
    EX_HIP_TRY(hipDeviceSynchronize());
    seq_buf.release();
    int rc = radix_sort(d_keys, d_tmp, total, 40);
    tmp_buf.release();
    if (rc) return rc;
    EX_HIP_TRY(hipMalloc(&text_buf.p, 21 * total));
    char *d_text = static_cast<char *>(text_buf.p);
    hipLaunchKernelGGL(k_keys_to_text, dim3(static_cast<uint32_t>((total + 255) / 256)), dim3(256), 0, nullptr, d_keys,
                       total, d_text);
    char *host = static_cast<char *>(std::malloc(21 * total));
    if (!host) {
        set_error("out of memory");
        return ISSL_E_NOMEM;
    }
    hipError_t e = hipMemcpy(host, d_text, 21 * total, hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        std::free(host);
        set_error(std::string("HIP error: ") + hipGetErrorString(e));
        return ISSL_E_DEVICE;
    }
    *out_text = host;
    *out_len = 21 * total;
    *n_sites = total;
    return ISSL_OK;
}

} // namespace

} // namespace issl

extern "C" {

int issl_extract_from_memory(const char *const *files, const size_t *lens, int n_files, int device, char **out_text,
                             size_t *out_len, uint64_t *n_sites)
{
    if (!files || !lens || n_files <= 0 || !out_text || !out_len || !n_sites) {
        issl::set_error("null argument");
        return ISSL_E_ARG;
    }
    std::string seq;
    for (int f = 0; f < n_files; ++f) issl::append_records(files[f], lens[f], seq);
    return issl::extract_sorted_text(seq, device, out_text, out_len, n_sites);
}

int issl_extract_offtargets(const char *const *inputs, int n_inputs, const char *output_path, int device,
                            uint64_t *n_sites)
{
    if (!inputs || n_inputs <= 0 || !output_path || !n_sites) {
        issl::set_error("null argument");
        return ISSL_E_ARG;
    }
    std::string seq;
    for (int f = 0; f < n_inputs; ++f) {
        FILE *fp = std::fopen(inputs[f], "rb");
        if (!fp) {
            issl::set_error(std::string("cannot open '") + inputs[f] + "'");
            return ISSL_E_IO;
        }
        std::fseek(fp, 0, SEEK_END);
        const long sz = std::ftell(fp);
        std::fseek(fp, 0, SEEK_SET);
        std::vector<char> buf(sz > 0 ? static_cast<size_t>(sz) : 0);
        if (sz > 0 && std::fread(buf.data(), buf.size(), 1, fp) < 1) {
            std::fclose(fp);
            issl::set_error(std::string("cannot read '") + inputs[f] + "'");
            return ISSL_E_IO;
        }
        std::fclose(fp);
        issl::append_records(buf.data(), buf.size(), seq);
    }
    char *text = nullptr;
    size_t len = 0;
    int rc = issl::extract_sorted_text(seq, device, &text, &len, n_sites);
    if (rc) return rc;
    FILE *out = std::fopen(output_path, "wb");
    if (!out) {
        std::free(text);
        issl::set_error(std::string("cannot write '") + output_path + "'");
        return ISSL_E_IO;
    }
    const bool ok = (len == 0 || std::fwrite(text, 1, len, out) == len);
    const bool closed = std::fclose(out) == 0;
    std::free(text);
    if (!ok || !closed) {
        issl::set_error(std::string("short write to '") + output_path + "'");
        return ISSL_E_IO;
    }
    return ISSL_OK;
}

} // extern "C"
