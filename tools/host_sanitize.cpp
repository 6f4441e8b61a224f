// Host-side parser / builder under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only: GPU sanitizers are
// not available on the pool).  Built and run by tests/test_host_sanitizers.py:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined tools/host_sanitize.cpp \
//       crackling_amd/csrc/issl_host.cpp -lpthread -o <tmp>/host_sanitize && <tmp>/host_sanitize <index.issl> <sites.txt>
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
    std::printf("ok: %zu corrupted or truncated images rejected, %zu accepted as consistent (checksum %llu)\n", rejected, accepted,
                static_cast<unsigned long long>(sink));
    return rejected > 100 ? 0 : 1;
}
