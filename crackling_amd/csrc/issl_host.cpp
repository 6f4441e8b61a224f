// Host-side ISSL index: parsing, validation, construction.  See issl_host.hpp.
#include "issl_host.hpp"

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

namespace issl {

static thread_local std::string g_error;
void set_error(const std::string &msg) { g_error = msg; }
const char *get_error() { return g_error.c_str(); }

HostIndex::~HostIndex()
{
    if (map_) munmap(map_, map_len_);
    if (fd_ >= 0) ::close(fd_);
}

bool HostIndex::file_range(const void *p, size_t bytes, int *fd, uint64_t *offset) const
{
    const uint8_t *b = static_cast<const uint8_t *>(p), *m = static_cast<const uint8_t *>(map_);
    if (fd_ < 0 || !map_ || b < m || b + bytes > m + map_len_) return false;
    *fd = fd_;
    *offset = static_cast<uint64_t>(b - m);
    return true;
}

uint64_t encode_guide(const char *p, size_t seq_len)
{
    // A/C/G/T -> 0..3 ; every other byte packs as 0 (isslScoreOfftargets.cpp:42,99-102)
    uint64_t sig = 0;
    for (size_t j = 0; j < seq_len; ++j) {
        const unsigned char c = static_cast<unsigned char>(p[j]);
        const uint64_t code = (c == 'C') ? 1u : (c == 'G') ? 2u : (c == 'T') ? 3u : 0u;
        sig |= code << (2 * j);
    }
    return sig;
}

void decode_guide(uint64_t sig, size_t seq_len, char *out)
{
    for (size_t j = 0; j < seq_len; ++j) out[j] = "ACGT"[(sig >> (2 * j)) & 3u];
    out[seq_len] = '\0';
}

int method_from_string(const char *s)
{
    if (!s) return ISSL_METHOD_UNKNOWN;
    if (!std::strcmp(s, "and")) return ISSL_METHOD_AND;
    if (!std::strcmp(s, "or")) return ISSL_METHOD_OR;
    if (!std::strcmp(s, "avg")) return ISSL_METHOD_AVG;
    if (!std::strcmp(s, "mit")) return ISSL_METHOD_MIT;
    if (!std::strcmp(s, "cfd")) return ISSL_METHOD_CFD;
    return ISSL_METHOD_UNKNOWN;
}

// Hsu et al. position weights as tabulated in isslCreateIndex.cpp:96.
static const double kHsuWeight[20] = {0.0,   0.0,   0.014, 0.0,   0.0,   0.395, 0.317,
                                      0.0,   0.389, 0.079, 0.445, 0.508, 0.613, 0.851,
                                      0.732, 0.828, 0.615, 0.804, 0.685, 0.583};

double local_mit_score(uint64_t mask, size_t seq_len)
{
    // Same operation order as isslCreateIndex.cpp:93-130 so that the doubles come out bit-equal:
    // product of (1-w) ascending, mean gap, then ((t1*t2)*t3)*100.
    int where[32];
    int k = 0;
    for (size_t j = 0; j < seq_len && j < 32; ++j)
        if ((mask >> (2 * j)) & 3u) where[k++] = static_cast<int>(j);
    if (k == 0) return 0.0;
    double t1 = 1.0;
    for (int i = 0; i < k; ++i) t1 = t1 * (1.0 - kHsuWeight[where[i] < 20 ? where[i] : 0]);
    double gap = 0.0;
    if (k == 1) {
        gap = 19.0;
    } else {
        for (int i = 0; i + 1 < k; ++i) gap += where[i + 1] - where[i];
        gap = gap / (k - 1);
    }
    const double t2 = 1.0 / ((19.0 - gap) / 19.0 * 4.0 + 1);
    const double t3 = 1.0 / (k * k);
    return t1 * t2 * t3 * 100;
}

static bool mul_overflows(uint64_t a, uint64_t b, uint64_t *out)
{
    return __builtin_mul_overflow(a, b, out);
}

int HostIndex::parse(const uint8_t *p, size_t len)
{
    if (len < 48) {
        set_error("Error reading index: header invalid");
        return ISSL_E_FORMAT;
    }
    uint64_t h[6];
    std::memcpy(h, p, 48);
    geo.n_sites = h[0];
    geo.seq_len = h[1];
    geo.n_lines = h[2];
    geo.slice_width = h[3];
    geo.n_slices = h[4];
    geo.n_scores = h[5];
    if (geo.seq_len == 0 || geo.seq_len > 32 || geo.slice_width == 0 || geo.slice_width > 16 ||
        geo.n_slices == 0 || geo.n_slices * geo.slice_width > 64) {
        set_error("Error reading index: header invalid (sequence length / slice geometry out of range)");
        return ISSL_E_FORMAT;
    }
    if (geo.n_sites == 0) {
        set_error("Error reading index: loading off-target sequences failed");
        return ISSL_E_FORMAT;
    }
    if (geo.n_sites > 0xFFFFFFFFull) {
        set_error("Error reading index: more than 2^32 distinct sites cannot be addressed by 32-bit ids");
        return ISSL_E_FORMAT;
    }
    uint64_t score_bytes, site_bytes, n_entries, entry_bytes;
    const uint64_t nb = geo.n_buckets();
    if (mul_overflows(geo.n_scores, 16, &score_bytes) || mul_overflows(geo.n_sites, 8, &site_bytes) ||
        mul_overflows(geo.n_sites, geo.n_slices, &n_entries) || mul_overflows(n_entries, 8, &entry_bytes)) {
        set_error("Error reading index: header invalid (section sizes overflow)");
        return ISSL_E_FORMAT;
    }
    uint64_t off = 48;
    if (len - off < score_bytes) {
        set_error("Error reading index: precalculated scores truncated");
        return ISSL_E_FORMAT;
    }
    own_masks_.resize(geo.n_scores);
    own_vals_.resize(geo.n_scores);
    for (uint64_t i = 0; i < geo.n_scores; ++i) {
        std::memcpy(&own_masks_[i], p + off + 16 * i, 8);
        std::memcpy(&own_vals_[i], p + off + 16 * i + 8, 8);
    }
    score_mask = own_masks_.data();
    score_val = own_vals_.data();
    off += score_bytes;
    if (len - off < site_bytes) {
        set_error("Error reading index: loading off-target sequences failed");
        return ISSL_E_FORMAT;
    }
    sites = reinterpret_cast<const uint64_t *>(p + off);
    off += site_bytes;
    if (len - off < nb * 8) {
        set_error("Error reading index: reading slice list sizes failed");
        return ISSL_E_FORMAT;
    }
    sizes = reinterpret_cast<const uint64_t *>(p + off);
    off += nb * 8;
    if (len - off < entry_bytes) {
        set_error("Error reading index: reading slice contents failed");
        return ISSL_E_FORMAT;
    }
    entries = reinterpret_cast<const uint64_t *>(p + off);
    uint64_t total = 0;
    for (uint64_t b = 0; b < nb; ++b) {
        if (sizes[b] > n_entries || (total += sizes[b]) > n_entries) {
            set_error("Error reading index: slice list sizes exceed the slice contents");
            return ISSL_E_FORMAT;
        }
    }
    return ISSL_OK;
}

int HostIndex::open_file(const char *path)
{
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) {
        set_error(std::string("cannot open index file '") + path + "': " + std::strerror(errno));
        return ISSL_E_IO;
    }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size <= 0) {
        ::close(fd);
        set_error("Error reading index: header invalid");
        return ISSL_E_FORMAT;
    }
    void *m = mmap(nullptr, static_cast<size_t>(st.st_size), PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) {
        ::close(fd);
        set_error(std::string("cannot map index file '") + path + "': " + std::strerror(errno));
        return ISSL_E_IO;
    }
    map_ = m;
    map_len_ = static_cast<size_t>(st.st_size);
    fd_ = fd; // kept: the upload reads the big sections with pread (file_range), which is several times faster than faulting the mapping in
    madvise(m, map_len_, MADV_SEQUENTIAL);
    return parse(static_cast<const uint8_t *>(m), map_len_);
}

int HostIndex::from_memory(const void *image, size_t len)
{
    image_.assign(static_cast<const uint8_t *>(image), static_cast<const uint8_t *>(image) + len);
    return parse(image_.data(), image_.size());
}

int HostIndex::build_from_text(const char *text, size_t n_lines, size_t seq_len, size_t slice_width)
{
    if (!text || seq_len == 0 || seq_len > 32) {
        set_error("Sequence length is greater than 32, which is the maximum supported currently");
        return ISSL_E_ARG;
    }
    // Collapse runs of identical consecutive lines (the list is assumed sorted,
    // isslCreateIndex.cpp:184-207; lines are compared as text, exactly like the reference's memcmp, so two lines
    // that differ only in a non-ACGT byte stay two sites).  The run never extends past the last line.
    const size_t stride = seq_len + 1;
    std::vector<uint64_t> line_sig(n_lines);
    std::vector<uint8_t> starts_run(n_lines);
    {
        const size_t n_threads = std::max<size_t>(1, std::min<size_t>(std::thread::hardware_concurrency(), 16));
        std::vector<std::thread> pool;
        for (size_t t = 0; t < n_threads; ++t) {
            pool.emplace_back([&, t]() {
                const size_t lo = n_lines * t / n_threads, hi = n_lines * (t + 1) / n_threads;
                for (size_t i = lo; i < hi; ++i) {
                    const char *cur = text + i * stride;
                    line_sig[i] = encode_guide(cur, seq_len);
                    starts_run[i] = (i == 0) || std::memcmp(cur - stride, cur, seq_len) != 0;
                }
            });
        }
        for (auto &th : pool) th.join();
    }
    std::vector<uint64_t> sigs;
    std::vector<uint32_t> occ;
    sigs.reserve(n_lines);
    occ.reserve(n_lines);
    for (size_t i = 0; i < n_lines; ++i) {
        if (starts_run[i]) {
            sigs.push_back(line_sig[i]);
            occ.push_back(1);
        } else {
            ++occ.back();
        }
    }
    if (n_lines == 0) {
        set_error("site list is empty");
        return ISSL_E_ARG;
    }
    return build_from_sites(sigs.data(), occ.data(), sigs.size(), n_lines, seq_len, slice_width);
}

int HostIndex::build_from_sites(const uint64_t *sigs, const uint32_t *occ, size_t n_sites, size_t n_lines,
                                size_t seq_len, size_t slice_width)
{
    if (seq_len == 0 || seq_len > 32) {
        set_error("sequence length must be 1..32");
        return ISSL_E_ARG;
    }
    if (slice_width < 2 || slice_width > 8 || (seq_len * 2) / slice_width == 0) {
        // The reference builder truncates slice values to 8 bits (isslCreateIndex.cpp:228) and
        // miscounts masks for width 1; refuse instead of writing an index that mis-scores.
        set_error("slice width must be 2..8 bits");
        return ISSL_E_UNSUPPORTED;
    }
    if (n_sites == 0 || n_sites > 0xFFFFFFFFull) {
        set_error("site count must be 1..2^32-1");
        return ISSL_E_ARG;
    }
    geo.n_sites = n_sites;
    geo.seq_len = seq_len;
    geo.n_lines = n_lines;
    geo.slice_width = slice_width;
    geo.n_slices = (seq_len * 2) / slice_width; // isslCreateIndex.cpp:213
    own_sites_.assign(sigs, sigs + n_sites);
    sites = own_sites_.data();

    const uint64_t per_slice = geo.buckets_per_slice();
    const uint64_t nb = geo.n_buckets();
    own_sizes_.assign(nb, 0);
    own_entries_.resize(n_sites * geo.n_slices);
    // One counting sort per slice; ids ascend inside every bucket (isslCreateIndex.cpp:218-234).
    // Slice s owns entries [s*n_sites, (s+1)*n_sites).
    auto do_slice = [&](uint64_t s) {
        const int shift = static_cast<int>(slice_width * s);
        const uint64_t low = per_slice - 1;
        uint64_t *cnt = own_sizes_.data() + s * per_slice;
        for (size_t id = 0; id < n_sites; ++id) ++cnt[(sigs[id] >> shift) & low];
        std::vector<uint64_t> cursor(per_slice);
        uint64_t run = s * n_sites;
        for (uint64_t k = 0; k < per_slice; ++k) {
            cursor[k] = run;
            run += cnt[k];
        }
        for (size_t id = 0; id < n_sites; ++id) {
            const uint64_t k = (sigs[id] >> shift) & low;
            own_entries_[cursor[k]++] = (static_cast<uint64_t>(occ[id]) << 32) | static_cast<uint64_t>(id);
        }
    };
    std::vector<std::thread> pool;
    for (uint64_t s = 0; s < geo.n_slices; ++s) pool.emplace_back(do_slice, s);
    for (auto &t : pool) t.join();
    sizes = own_sizes_.data();
    entries = own_entries_.data();
    return finish_build(slice_width);
}

static int check_geometry_args(size_t n_sites, size_t seq_len, size_t slice_width)
{
    if (seq_len == 0 || seq_len > 32) {
        set_error("sequence length must be 1..32");
        return ISSL_E_ARG;
    }
    if (slice_width < 2 || slice_width > 8 || (seq_len * 2) / slice_width == 0) {
        set_error("slice width must be 2..8 bits");
        return ISSL_E_UNSUPPORTED;
    }
    if (n_sites == 0 || n_sites > 0xFFFFFFFFull) {
        set_error("site count must be 1..2^32-1");
        return ISSL_E_ARG;
    }
    return ISSL_OK;
}

int HostIndex::init_without_arrays(const uint64_t *sigs, size_t n_sites, size_t n_lines, size_t seq_len,
                                   size_t slice_width)
{
    if (int rc = check_geometry_args(n_sites, seq_len, slice_width)) return rc;
    const uint64_t n_slices = (seq_len * 2) / slice_width, per_slice = 1ull << slice_width, nb = n_slices << slice_width;
    const unsigned n_threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<std::vector<uint64_t>> part(n_threads, std::vector<uint64_t>(nb, 0));
    auto count = [&](unsigned t) {
        const size_t lo = n_sites * t / n_threads, hi = n_sites * (t + 1) / n_threads;
        uint64_t *cnt = part[t].data();
        for (size_t id = lo; id < hi; ++id)
            for (uint64_t s = 0; s < n_slices; ++s) ++cnt[s * per_slice + ((sigs[id] >> (slice_width * s)) & (per_slice - 1))];
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < n_threads; ++t) pool.emplace_back(count, t);
    count(0);
    for (auto &t : pool) t.join();
    std::vector<uint64_t> total(nb, 0);
    for (const auto &p : part)
        for (uint64_t b = 0; b < nb; ++b) total[b] += p[b];
    return init_from_bucket_sizes(total.data(), n_sites, n_lines, seq_len, slice_width);
}

int HostIndex::init_from_bucket_sizes(const uint64_t *bucket_sizes, size_t n_sites, size_t n_lines, size_t seq_len,
                                      size_t slice_width)
{
    if (int rc = check_geometry_args(n_sites, seq_len, slice_width)) return rc;
    geo.n_sites = n_sites;
    geo.seq_len = seq_len;
    geo.n_lines = n_lines;
    geo.slice_width = slice_width;
    geo.n_slices = (seq_len * 2) / slice_width;
    const uint64_t nb = geo.n_buckets();
    own_sizes_.assign(bucket_sizes, bucket_sizes + nb);
    for (uint64_t s = 0; s < geo.n_slices; ++s) { // every slice lists every site once
        uint64_t sum = 0;
        for (uint64_t b = 0; b < geo.buckets_per_slice(); ++b) sum += own_sizes_[s * geo.buckets_per_slice() + b];
        if (sum != n_sites) {
            set_error("bucket sizes do not add up to the site count");
            return ISSL_E_ARG;
        }
    }
    sizes = own_sizes_.data();
    sites = nullptr;
    entries = nullptr;
    return finish_build(slice_width);
}

int HostIndex::finish_build(size_t slice_width)
{
    // Local MIT scores for every placement of 1..maxDist mismatches on 20 positions, ascending
    // by mask (isslCreateIndex.cpp:239-252, std::map order).  Enumerating the 20-bit position
    // sets in increasing order and spreading bit p to bit 2p keeps the masks ascending.
    const int max_dist = static_cast<int>(geo.seq_len * 2 / slice_width) - 1;
    own_masks_.clear();
    own_vals_.clear();
    for (uint32_t v = 1; v < (1u << 20); ++v) {
        if (__builtin_popcount(v) > max_dist) continue;
        uint64_t mask = 0;
        for (uint32_t t = v; t; t &= t - 1) mask |= 1ull << (2 * __builtin_ctz(t));
        own_masks_.push_back(mask);
        own_vals_.push_back(local_mit_score(mask, geo.seq_len));
    }
    geo.n_scores = own_masks_.size();
    score_mask = own_masks_.data();
    score_val = own_vals_.data();
    return ISSL_OK;
}

int HostIndex::write_leading_sections(FILE *fp) const
{
    const uint64_t h[6] = {geo.n_sites, geo.seq_len, geo.n_lines, geo.slice_width, geo.n_slices, geo.n_scores};
    bool ok = std::fwrite(h, 8, 6, fp) == 6;
    std::vector<uint64_t> pairs(2 * geo.n_scores);
    for (uint64_t i = 0; i < geo.n_scores; ++i) {
        pairs[2 * i] = score_mask[i];
        std::memcpy(&pairs[2 * i + 1], &score_val[i], 8);
    }
    ok = ok && std::fwrite(pairs.data(), 8, pairs.size(), fp) == pairs.size();
    return ok ? ISSL_OK : ISSL_E_IO;
}

int HostIndex::write_file(const char *path) const
{
    FILE *fp = std::fopen(path, "wb");
    if (!fp) {
        set_error(std::string("cannot write index file '") + path + "': " + std::strerror(errno));
        return ISSL_E_IO;
    }
    bool ok = write_leading_sections(fp) == ISSL_OK;
    ok = ok && std::fwrite(sites, 8, geo.n_sites, fp) == geo.n_sites;
    ok = ok && std::fwrite(sizes, 8, geo.n_buckets(), fp) == geo.n_buckets();
    const uint64_t ne = geo.n_sites * geo.n_slices;
    ok = ok && std::fwrite(entries, 8, ne, fp) == ne;
    ok = (std::fclose(fp) == 0) && ok;
    if (!ok) {
        set_error(std::string("short write to '") + path + "'");
        return ISSL_E_IO;
    }
    return ISSL_OK;
}

void HostIndex::unique_scores(std::vector<uint64_t> &masks, std::vector<double> &vals) const
{
    std::vector<uint64_t> order(geo.n_scores);
    for (uint64_t i = 0; i < geo.n_scores; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(),
                     [&](uint64_t a, uint64_t b) { return score_mask[a] < score_mask[b]; });
    masks.clear();
    vals.clear();
    for (uint64_t i : order) {
        if (!masks.empty() && masks.back() == score_mask[i]) continue; // first insert wins
        masks.push_back(score_mask[i]);
        vals.push_back(score_val[i]);
    }
}

} // namespace issl
