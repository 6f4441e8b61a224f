// Host-side ISSL index: .issl parsing/validation, index construction, geometry helpers.
// Product code (no oracle involved).  Reference paths are relative to /root/reference.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/issl_hip.h"

namespace issl {

// Thread-local error text behind issl_last_error().
void set_error(const std::string &msg);
const char *get_error();

// Geometry of an index (header fields + derived values).
struct Geometry {
    uint64_t n_sites = 0, seq_len = 0, n_lines = 0, slice_width = 0, n_slices = 0, n_scores = 0;
    uint64_t buckets_per_slice() const { return 1ull << slice_width; }
    uint64_t n_buckets() const { return n_slices << slice_width; }
};

// Host view of the sections of an .issl file (isslScoreOfftargets.cpp:152-243).  The arrays live
// either in a private file mapping or in vectors owned by this object (builder / from_memory).
class HostIndex {
  public:
    HostIndex() = default;
    ~HostIndex();
    HostIndex(const HostIndex &) = delete;
    HostIndex &operator=(const HostIndex &) = delete;

    Geometry geo;
    const uint64_t *score_mask = nullptr; // geo.n_scores pairs in file order: mask[i], val[i]
    const double *score_val = nullptr;    // (stored interleaved in the file; de-interleaved here)
    const uint64_t *sites = nullptr;      // geo.n_sites
    const uint64_t *sizes = nullptr;      // geo.n_buckets()
    const uint64_t *entries = nullptr;    // geo.n_sites * geo.n_slices

    // Parse + validate.  Return ISSL_OK or an error code (message via set_error).
    int open_file(const char *path);
    int from_memory(const void *image, size_t len);
    int build_from_text(const char *text, size_t n_lines, size_t seq_len, size_t slice_width);
    int build_from_sites(const uint64_t *sigs, const uint32_t *occ, size_t n_sites, size_t n_lines,
                         size_t seq_len, size_t slice_width);
    int write_file(const char *path) const;
    // Geometry, MIT table and bucket sizes only (threaded histogram of the slice values); `sites` and `entries` stay
    // null: the arrays of such an index exist only in the HBM image (issl_index_build_on_device).
    int init_without_arrays(const uint64_t *sigs, size_t n_sites, size_t n_lines, size_t seq_len, size_t slice_width);
    // The same from bucket lengths somebody else counted (a site table that lives in device memory).
    int init_from_bucket_sizes(const uint64_t *bucket_sizes, size_t n_sites, size_t n_lines, size_t seq_len, size_t slice_width);
    bool has_arrays() const { return sites != nullptr && entries != nullptr; }
    // [p, p + bytes) lies in the file this index is mapped from: its descriptor and the offset of p (for readers that want
    // the bytes without touching the mapping, e.g. pread into pinned memory).  False for indexes that were built or copied.
    bool file_range(const void *p, size_t bytes, int *fd, uint64_t *offset) const;
    // Header, score table (and, with `with_sites`, nothing more): the leading sections of write_file() for callers
    // that stream the big arrays themselves.
    int write_leading_sections(FILE *fp) const;

    // Sorted unique {mask, score} table, first occurrence wins (flat_hash_map::insert, :196).
    void unique_scores(std::vector<uint64_t> &masks, std::vector<double> &vals) const;

  private:
    int parse(const uint8_t *p, size_t len);
    int finish_build(size_t slice_width);
    void *map_ = nullptr;
    size_t map_len_ = 0;
    int fd_ = -1;
    std::vector<uint8_t> image_;         // from_memory copy
    std::vector<uint64_t> own_masks_;    // de-interleaved score table
    std::vector<double> own_vals_;
    std::vector<uint64_t> own_sites_, own_sizes_, own_entries_;
};

// A2: 2-bit packing, A=0 C=1 G=2 T=3, base j at bits 2j..2j+1, anything else 0
// (isslScoreOfftargets.cpp:63-71,99-102).
uint64_t encode_guide(const char *p, size_t seq_len);
void decode_guide(uint64_t sig, size_t seq_len, char *out); // writes seq_len chars + NUL

int method_from_string(const char *s);

// Local MIT score of a mismatch mask (isslCreateIndex.cpp:93-130), positions < seq_len only.
double local_mit_score(uint64_t mask, size_t seq_len);

} // namespace issl
