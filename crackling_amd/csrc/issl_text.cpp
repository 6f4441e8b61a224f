// Text at the process boundary of the scorer: the query file in (isslScoreOfftargets.cpp:275-305) and the TSV out
// (:514-527).  Host code only.  At a million guides per page both are as long as the scoring itself when done the
// obvious way (one thread, printf("%f") twice per line), so both run on several threads and the "%f" is a formatter of
// its own -- exact, digit for digit what glibc prints.
#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <mutex>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

#include "issl_host.hpp"

namespace issl {

// printf("%f") of a finite, non-negative double below 2^40, exactly: the decimal expansion of the binary value rounded
// to six places, ties to even -- what glibc's printf does in the default rounding mode (it works on the exact value,
// not on a shortest-round-trip approximation).  v = m * 2^-s with m < 2^53, so v * 10^6 = m * 10^6 / 2^s with a
// numerator below 2^73: one 128-bit product, a shift, and the remainder against half decides the rounding.
// Returns the number of characters written (no terminator), or 0 when the value is not one of these (negative, NaN,
// infinite, huge): the caller falls back to snprintf.  `out` needs 24 bytes.
int format_f6(double v, char *out)
{
    uint64_t bits;
    std::memcpy(&bits, &v, 8);
    if (bits >> 63) return 0; // negative (and -0.0, which prints "-0.000000")
    const uint32_t e = static_cast<uint32_t>(bits >> 52);
    const uint64_t frac = bits & ((1ull << 52) - 1ull);
    if (e >= 1023 + 40) return 0; // >= 2^40, infinite, NaN
    uint64_t q; // round(v * 10^6)
    if (e == 0 && frac == 0) {
        q = 0;
    } else {
        const uint64_t m = e ? (frac | (1ull << 52)) : frac;
        const uint32_t s = e ? 1075u - e : 1074u; // v = m * 2^-s, s >= 13
        const unsigned __int128 p = static_cast<unsigned __int128>(m) * 1000000u;
        if (s >= 75) {
            q = 0; // p < 2^73 <= 2^(s-2): below half of the last place
        } else {
            const unsigned __int128 one = static_cast<unsigned __int128>(1) << s;
            const unsigned __int128 rem = p & (one - 1), half = one >> 1;
            q = static_cast<uint64_t>(p >> s);
            if (rem > half || (rem == half && (q & 1u))) ++q;
        }
    }
    uint64_t ip = q / 1000000u;
    uint32_t fp = static_cast<uint32_t>(q % 1000000u);
    char tmp[16];
    int k = 0;
    do { tmp[k++] = static_cast<char>('0' + ip % 10); ip /= 10; } while (ip);
    int n = 0;
    while (k) out[n++] = tmp[--k];
    out[n++] = '.';
    for (int d = 5; d >= 0; --d) { out[n + d] = static_cast<char>('0' + fp % 10); fp /= 10; }
    return n + 6;
}

static inline char *append_f(char *p, double v)
{
    const int k = format_f6(v, p);
    if (k) return p + k;
    return p + std::snprintf(p, 400, "%f", v); // (a double prints in at most 1 + 309 + 1 + 6 characters)
}

// Output buffers are kept between calls (up to kPoolBytes of them): a page of a fresh allocation costs a fault when it is
// first written -- 10 000 of them for the 41 MB a million lines take, as long as formatting them -- and a resident scorer
// formats page after page of the same size.  A buffer carries its capacity in a 16-byte header in front of the text.
namespace {
constexpr size_t kPoolBytes = size_t(512) << 20, kBufHeader = 16;
struct BufPool {
    std::mutex mu;
    std::vector<char *> idle; // base pointers (header included)
    size_t bytes = 0;
    ~BufPool() { for (char *b : idle) std::free(b); }
} g_pool;
size_t buf_cap(const char *base) { size_t c; std::memcpy(&c, base, sizeof c); return c; }
char *buf_take(size_t cap) // returns the text pointer (base + header) of a buffer that holds at least `cap` bytes, or null
{
    {
        std::lock_guard<std::mutex> lock(g_pool.mu);
        size_t best = g_pool.idle.size();
        for (size_t i = 0; i < g_pool.idle.size(); ++i)
            if (buf_cap(g_pool.idle[i]) >= cap && (best == g_pool.idle.size() || buf_cap(g_pool.idle[i]) < buf_cap(g_pool.idle[best]))) best = i;
        if (best != g_pool.idle.size()) {
            char *base = g_pool.idle[best];
            g_pool.idle.erase(g_pool.idle.begin() + static_cast<long>(best));
            g_pool.bytes -= buf_cap(base);
            return base + kBufHeader;
        }
    }
    char *base = static_cast<char *>(std::malloc(cap + kBufHeader));
    if (!base) return nullptr;
    std::memcpy(base, &cap, sizeof cap);
    return base + kBufHeader;
}
char *buf_grow(char *text, size_t cap)
{
    char *base = static_cast<char *>(std::realloc(text - kBufHeader, cap + kBufHeader));
    if (!base) return nullptr;
    std::memcpy(base, &cap, sizeof cap);
    return base + kBufHeader;
}
void buf_give(char *text)
{
    if (!text) return;
    char *base = text - kBufHeader;
    const size_t cap = buf_cap(base);
    {
        std::lock_guard<std::mutex> lock(g_pool.mu);
        if (g_pool.bytes + cap <= kPoolBytes && g_pool.idle.size() < 64) {
            g_pool.idle.push_back(base);
            g_pool.bytes += cap;
            return;
        }
    }
    std::free(base);
}
} // namespace

static size_t worker_count(size_t n, size_t per_thread, int threads)
{
    size_t t = threads > 0 ? std::min<size_t>(static_cast<size_t>(threads), 64) : std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency()));
    return std::max<size_t>(1, std::min(t, n / per_thread));
}

} // namespace issl

using namespace issl;

extern "C" {

int issl_format_scores(const uint64_t *guides, const double *mit, const double *cfd, size_t n, size_t seq_len, int method,
                       int threads, issl_span **spans, size_t *n_spans)
{
    if (!spans || !n_spans || (n && (!guides || !mit || !cfd)) || seq_len == 0 || seq_len > 32) {
        set_error("bad argument to issl_format_scores");
        return ISSL_E_ARG;
    }
    const bool want_mit = method == ISSL_METHOD_MIT || method == ISSL_METHOD_AND || method == ISSL_METHOD_OR || method == ISSL_METHOD_AVG;
    const bool want_cfd = method == ISSL_METHOD_CFD || method == ISSL_METHOD_AND || method == ISSL_METHOD_OR || method == ISSL_METHOD_AVG;
    const size_t nt = worker_count(n, 16384, threads);
    issl_span *out = static_cast<issl_span *>(std::calloc(nt, sizeof(issl_span)));
    if (!out) { set_error("out of memory"); return ISSL_E_NOMEM; }
    std::vector<int> failed(nt, 0);
    auto work = [&](size_t t) {
        const size_t lo = n * t / nt, hi = n * (t + 1) / nt;
        // a line of in-range scores takes seq_len + 2 * 21 + 3 bytes at most; a score that falls back to snprintf (a
        // negative, huge or non-finite one) may take 400: the buffer grows when fewer than two such lines are left
        size_t cap = (hi - lo) * (seq_len + 48) + 1024;
        char *buf = buf_take(cap);
        if (!buf) { failed[t] = 1; return; }
        cap = buf_cap(buf - kBufHeader);
        char *p = buf;
        for (size_t i = lo; i < hi; ++i) {
            if (static_cast<size_t>(p - buf) + 1024 > cap) {
                const size_t used = static_cast<size_t>(p - buf);
                cap = cap * 2 + 4096;
                char *grown = buf_grow(buf, cap);
                if (!grown) { buf_give(buf); failed[t] = 1; return; }
                buf = grown;
                p = buf + used;
            }
            const uint64_t sig = guides[i];
            for (size_t j = 0; j < seq_len; ++j) p[j] = "ACGT"[(sig >> (2 * j)) & 3u]; // :82-89
            p += seq_len;
            *p++ = '\t';
            if (want_mit) p = append_f(p, mit[i]); else { *p++ = '-'; *p++ = '1'; } // :517-525
            *p++ = '\t';
            if (want_cfd) p = append_f(p, cfd[i]); else { *p++ = '-'; *p++ = '1'; }
            *p++ = '\n';
        }
        out[t].data = buf;
        out[t].len = static_cast<size_t>(p - buf);
    };
    {
        std::vector<std::thread> pool;
        for (size_t t = 1; t < nt; ++t) pool.emplace_back(work, t);
        work(0);
        for (auto &th : pool) th.join();
    }
    for (size_t t = 0; t < nt; ++t)
        if (failed[t]) {
            issl_free_spans(out, nt);
            set_error("out of memory");
            return ISSL_E_NOMEM;
        }
    *spans = out;
    *n_spans = nt;
    return ISSL_OK;
}

void issl_free_spans(issl_span *spans, size_t n_spans)
{
    if (!spans) return;
    for (size_t t = 0; t < n_spans; ++t) buf_give(spans[t].data);
    std::free(spans);
}

int issl_read_query_file(const char *path, size_t seq_len, uint64_t **out, size_t *n)
{
    if (!path || !out || !n || seq_len == 0 || seq_len > 32) { set_error("bad argument"); return ISSL_E_ARG; }
    const int fd = ::open(path, O_RDONLY | O_CLOEXEC);
    struct stat st;
    if (fd < 0 || ::fstat(fd, &st) != 0) {
        if (fd >= 0) ::close(fd);
        set_error(std::string("cannot open query file '") + path + "'");
        return ISSL_E_IO;
    }
    const size_t line = seq_len + 1;
    const size_t sz = static_cast<size_t>(st.st_size);
    if (st.st_size < 0 || sz % line != 0) { // isslScoreOfftargets.cpp:277-282
        ::close(fd);
        set_error("Error: query file is not a multiple of the expected line length (" + std::to_string(line) +
                  ")\nThe sequence length may be incorrect; alternatively, the line endings\nmay be something "
                  "other than LF, or there may be junk at the end of the file.");
        return ISSL_E_FORMAT;
    }
    if (sz == 0) { // :290-293
        ::close(fd);
        set_error("Failed to read in query file.");
        return ISSL_E_FORMAT;
    }
    const size_t count = sz / line;
    uint64_t *g = static_cast<uint64_t *>(std::malloc(8 * count));
    if (!g) { ::close(fd); set_error("out of memory"); return ISSL_E_NOMEM; }
    // every thread reads and packs its own stretch of lines (pread: no shared file position)
    const size_t nt = worker_count(count, 65536, 0);
    std::vector<int> failed(nt, 0);
    auto work = [&](size_t t) {
        const size_t lo = count * t / nt, hi = count * (t + 1) / nt;
        const size_t step = 32768; // lines per read
        std::vector<char> buf(step * line);
        for (size_t i = lo; i < hi; i += step) {
            const size_t k = std::min(step, hi - i), want = k * line;
            size_t got = 0;
            while (got < want) {
                const ssize_t r = ::pread(fd, buf.data() + got, want - got, static_cast<off_t>(i * line + got));
                if (r < 0 && errno == EINTR) continue;
                if (r <= 0) { failed[t] = 1; return; }
                got += static_cast<size_t>(r);
            }
            for (size_t j = 0; j < k; ++j) g[i + j] = encode_guide(buf.data() + j * line, seq_len);
        }
    };
    {
        std::vector<std::thread> pool;
        for (size_t t = 1; t < nt; ++t) pool.emplace_back(work, t);
        work(0);
        for (auto &th : pool) th.join();
    }
    ::close(fd);
    for (size_t t = 0; t < nt; ++t)
        if (failed[t]) {
            std::free(g);
            set_error("Failed to read in query file.");
            return ISSL_E_FORMAT;
        }
    *out = g;
    *n = count;
    return ISSL_OK;
}

} // extern "C"
