// Host-side parser / builder under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only: GPU sanitizers are
// not available on the pool).  Built and run by tests/test_host_sanitizers.py:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined tools/host_sanitize.cpp \
//       crackling_amd/csrc/issl_host.cpp crackling_amd/csrc/issl_text.cpp -lpthread -o <tmp>/host_sanitize && <tmp>/host_sanitize <index.issl> <sites.txt>
// The .issl reader is the one piece of the product that parses untrusted bytes (isslScoreOfftargets.cpp:152-243 trusts
// its file; this reader validates sizes, truncation and overflow): every truncation of the file, and a few thousand
// random single-field and single-byte corruptions, must come back as a clean error or a consistent index -- never as
// a sanitizer report.  Then the builder must reproduce the file from the sorted site list.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../crackling_amd/csrc/issl_host.hpp"

extern "C" void issl_free(void *p) { std::free(p); } // (lives in issl_capi.cpp, which needs the HIP runtime)

static std::vector<uint8_t> slurp(const char *path)
{
    FILE *f = std::fopen(path, "rb");
    if (!f) { std::fprintf(stderr, "cannot open %s\n", path); std::exit(2); }
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> d(static_cast<size_t>(n));
    if (n && std::fread(d.data(), d.size(), 1, f) != 1) std::exit(2);
    std::fclose(f);
    return d;
}

// Touch everything a consumer of a parsed index touches.
static uint64_t walk(const issl::HostIndex &h)
{
    uint64_t sum = 0;
    const uint64_t nb = h.geo.n_buckets();
    uint64_t total = 0;
    for (uint64_t b = 0; b < nb; ++b) total += h.sizes[b];
    for (uint64_t i = 0; i < h.geo.n_sites; ++i) sum += h.sites[i];
    for (uint64_t i = 0; i < total; ++i) sum += h.entries[i];
    std::vector<uint64_t> m;
    std::vector<double> v;
    h.unique_scores(m, v);
    for (size_t i = 0; i < m.size(); ++i) sum += m[i] + static_cast<uint64_t>(v[i]);
    return sum;
}

int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: host_sanitize <index.issl> <sites.txt>\n"); return 2; }
    const std::vector<uint8_t> good = slurp(argv[1]);
    const std::vector<uint8_t> text = slurp(argv[2]);
    uint64_t sink = 0;
    size_t accepted = 0, rejected = 0;
    {
        issl::HostIndex h;
        if (h.from_memory(good.data(), good.size())) { std::fprintf(stderr, "golden index rejected: %s\n", issl::get_error()); return 1; }
        sink += walk(h);
    }
    // every truncation (stride keeps the run short; all lengths inside the header and the last 64 bytes are covered)
    for (size_t len = 0; len < good.size(); len += (len < 4096 || len + 64 >= good.size()) ? 1 : 97) {
        issl::HostIndex h;
        if (h.from_memory(good.data(), len) == 0) { ++accepted; sink += walk(h); } else ++rejected;
    }
    // random corruptions: header fields replaced by edge values, bucket sizes, single bytes anywhere
    std::mt19937_64 rng(20261004);
    const uint64_t edges[] = {0, 1, 2, 7, 8, 20, 21, 32, 33, 64, 255, 256, 1ull << 31, 1ull << 32, (1ull << 32) + 1,
                              1ull << 40, 1ull << 61, ~0ull, ~0ull >> 1, good.size(), good.size() / 8};
    for (int it = 0; it < 4000; ++it) {
        std::vector<uint8_t> bad = good;
        const int kind = static_cast<int>(rng() % 3);
        if (kind == 0) { // a header field
            const uint64_t v = edges[rng() % (sizeof edges / sizeof edges[0])];
            std::memcpy(bad.data() + 8 * (rng() % 6), &v, 8);
        } else if (kind == 1) { // a 64-bit word anywhere (bucket sizes, entries, masks)
            const size_t at = (rng() % (bad.size() / 8)) * 8;
            const uint64_t v = (rng() & 1) ? edges[rng() % (sizeof edges / sizeof edges[0])] : rng();
            std::memcpy(bad.data() + at, &v, 8);
        } else {
            bad[rng() % bad.size()] ^= static_cast<uint8_t>(1u << (rng() % 8));
        }
        issl::HostIndex h;
        if (h.from_memory(bad.data(), bad.size()) == 0) { ++accepted; sink += walk(h); } else ++rejected;
    }
    // builder: the sorted site list gives the file back
    {
        issl::HostIndex h;
        if (h.build_from_text(reinterpret_cast<const char *>(text.data()), text.size() / 21, 20, 8)) {
            std::fprintf(stderr, "builder failed: %s\n", issl::get_error());
            return 1;
        }
        const std::string out = std::string(argv[1]) + ".sanitize.tmp";
        if (h.write_file(out.c_str())) { std::fprintf(stderr, "write failed: %s\n", issl::get_error()); return 1; }
        const std::vector<uint8_t> again = slurp(out.c_str());
        std::remove(out.c_str());
        if (again != good) { std::fprintf(stderr, "builder bytes differ from the golden index\n"); return 1; }
    }
    // guide text: every byte value, both lengths of line
    char buf[33];
    for (int c = 0; c < 256; ++c) {
        std::memset(buf, c, sizeof buf);
        sink += issl::encode_guide(buf, 20) + issl::encode_guide(buf, 32);
        char dec[40];
        issl::decode_guide(sink, 20, dec);
        issl::decode_guide(sink, 32, dec);
    }
    // the scorer's text (issl_text.cpp): the library's own "%f" against snprintf on doubles of every exponent and on the exact
    // ties k/128, through issl_format_scores on several threads (buffers taken from and given back to its pool), and the
    // threaded query-file reader on a file of odd length
    {
        const size_t n = 70000;
        std::vector<uint64_t> g(n);
        std::vector<double> a(n), b(n);
        for (size_t i = 0; i < n; ++i) {
            g[i] = rng() & ((1ull << 40) - 1);
            uint64_t bits = rng() >> (i % 3 ? 1 : 0); // (two thirds with the sign bit clear)
            std::memcpy(&a[i], &bits, 8);
            b[i] = (i % 5 == 0) ? static_cast<double>(2 * (rng() % 4000) + 1) / 128.0 : 10000.0 / (100.0 + static_cast<double>(rng() % 1000000) / 97.0);
        }
        for (int round = 0; round < 3; ++round) {
            issl_span *spans = nullptr;
            size_t n_spans = 0;
            if (issl_format_scores(g.data(), a.data(), b.data(), n, 20, ISSL_METHOD_AND, round == 0 ? 1 : 4, &spans, &n_spans)) { std::fprintf(stderr, "format failed\n"); return 1; }
            size_t i = 0;
            for (size_t sp = 0; sp < n_spans; ++sp) {
                const char *q = spans[sp].data, *end = q + spans[sp].len;
                while (q < end) {
                    char want[1024], seq[40];
                    issl::decode_guide(g[i], 20, seq);
                    const int k = std::snprintf(want, sizeof want, "%s\t%f\t%f\n", seq, a[i], b[i]);
                    if (static_cast<size_t>(end - q) < static_cast<size_t>(k) || std::memcmp(q, want, static_cast<size_t>(k)) != 0) {
                        std::fprintf(stderr, "line %zu differs from printf's: %s", i, want);
                        return 1;
                    }
                    q += k;
                    ++i;
                }
            }
            if (i != n) { std::fprintf(stderr, "formatted %zu of %zu lines\n", i, n); return 1; }
            sink += n_spans;
            issl_free_spans(spans, n_spans);
        }
        const std::string qpath = std::string(argv[1]) + ".query.tmp";
        FILE *qf = std::fopen(qpath.c_str(), "wb");
        const size_t n_lines = 200003;
        for (size_t i = 0; i < n_lines; ++i) {
            char seq[40];
            issl::decode_guide(g[i % n], 20, seq);
            seq[20] = '\n';
            std::fwrite(seq, 1, 21, qf);
        }
        std::fclose(qf);
        uint64_t *out = nullptr;
        size_t got = 0;
        if (issl_read_query_file(qpath.c_str(), 20, &out, &got) || got != n_lines) { std::fprintf(stderr, "query reader failed: %s\n", issl::get_error()); return 1; }
        for (size_t i = 0; i < n_lines; ++i)
            if (out[i] != g[i % n]) { std::fprintf(stderr, "query reader: guide %zu differs\n", i); return 1; }
        issl_free(out);
        std::remove(qpath.c_str());
        // an index opened from its file keeps the descriptor for the upload's pread path: ranges inside and outside the mapping
        issl::HostIndex h;
        if (h.open_file(argv[1])) { std::fprintf(stderr, "open_file: %s\n", issl::get_error()); return 1; }
        int fd = -1;
        uint64_t off = 0;
        if (!h.file_range(h.sites, 8 * h.geo.n_sites, &fd, &off) || fd < 0 || off < 48) { std::fprintf(stderr, "file_range: site table not found in the mapping\n"); return 1; }
        if (h.file_range(g.data(), 8, &fd, &off)) { std::fprintf(stderr, "file_range: a foreign pointer accepted\n"); return 1; }
        sink += off;
    }
    std::printf("ok: %zu corrupted or truncated images rejected, %zu accepted as consistent (checksum %llu)\n", rejected, accepted,
                static_cast<unsigned long long>(sink));
    return rejected > 100 ? 0 : 1;
}
