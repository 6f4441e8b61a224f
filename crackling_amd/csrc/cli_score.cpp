// bin/isslScoreOfftargets -- drop-in for the reference scorer process
// (src/ISSL/isslScoreOfftargets.cpp:91-530, invoked by src/crackling/Crackling.py:767-778):
//
//   isslScoreOfftargets [issltable] [query file] [max distance] [score-threshold] [score-method]
//
// stdout carries data only ("<20-mer>\t<MIT>\t<CFD>\n" per guide, input order, "-1" for a score
// that was not requested); diagnostics go to stderr; exit status 1 on any error.
// Optional environment (the five positionals stay untouched so that Crackling needs no change):
//   ISSL_DEVICE=<n>   HIP device to use (default 0)
//   ISSL_DEVICES=all | <a,b,...>   several GPUs of the node: index image broadcast over RCCL/xGMI, guides handed out
//                     in chunks (opt-in only: by default one device is used, whatever the size of the query file --
//                     a page process must not fail because a neighbour GPU is busy or RCCL cannot start)
//   ISSL_TIMING=1     one JSON line on stderr: where the wall time of this invocation went (library load, index open,
//                     query parse, device runtime start, upload + layout transform, scoring, formatting, writing)
//   ISSL_LIBRARY=<path>   libissl_hip.so to load (default: ../crackling_amd/ next to the executable, then the loader's path)
//   ISSL_SERVER=<unix socket path>   resident mode, see below
//   ISSL_VERDICTS=<file>   also write "<20-mer>\t<0|1>\n" per guide: the accept/reject decision Crackling derives
//                     from stdout (Crackling.py:780-835), so that a caller can skip parsing the floats
//
// Resident mode.  Crackling starts one scorer process per page of guides (config.ini:106-112) and the reference
// reloads the whole index every time.  `isslScoreOfftargets --serve <socket>` keeps every index it has been asked
// for uploaded in HBM; a normal invocation with ISSL_SERVER=<socket> set hands its five arguments to that server
// and the answer arrives on stdout (same bytes, same exit status): when stdout is a regular file -- Crackling redirects it
// into one, Crackling.py:767 -- the descriptor itself travels to the server (SCM_RIGHTS) and the server writes the text
// where it belongs; otherwise the text comes back over the socket and is copied.  If the server cannot be reached the
// process scores by itself, so the variable is always safe to set.  `isslScoreOfftargets --stop <socket>` ends the server.
//
// The executable does not link the library: libissl_hip.so (and with it the HIP runtime, ~15 ms of loader work and more
// once a device is present) is loaded with dlopen when this process has to score by itself, never when a resident
// server answers -- a page of 10 000 guides is scored in less time than the runtime takes to load.
#include <atomic>
#include <cerrno>
#include <chrono>
#include <climits>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <algorithm>
#include <thread>
#include <dlfcn.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <unistd.h>
#include <vector>

#include "../../include/issl_hip.h"

namespace {

double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// ---- the library, loaded on demand ----------------------------------------------------------------------------------
#define ISSL_CLI_API(X)                                                                                                   \
    X(issl_last_error) X(issl_abi_version) X(issl_index_open) X(issl_index_set_option) X(issl_index_header)                \
    X(issl_index_upload) X(issl_index_device_bytes) X(issl_index_close) X(issl_device_memory) X(issl_read_query_file)     \
    X(issl_free) X(issl_method_from_string) X(issl_score) X(issl_last_stats) X(issl_decode_guide) X(issl_verdicts)        \
    X(issl_format_scores) X(issl_free_spans) X(issl_node_create) X(issl_node_get_info) X(issl_node_score) X(issl_node_close)
struct Api {
#define X(f) decltype(&::f) f = nullptr;
    ISSL_CLI_API(X)
#undef X
    double load_ms = 0;
};
Api api;

// Loads libissl_hip.so once; false + message on stderr when it cannot be had (the caller exits 1).
bool load_api()
{
    if (api.issl_score) return true;
    const double t0 = now_ms();
    std::vector<std::string> tried;
    void *h = nullptr;
    auto attempt = [&](const std::string &path) {
        if (h || path.empty()) return;
        h = ::dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) tried.push_back(path + ": " + ::dlerror());
    };
    if (const char *e = std::getenv("ISSL_LIBRARY")) attempt(e);
    char exe[PATH_MAX];
    const ssize_t k = ::readlink("/proc/self/exe", exe, sizeof exe - 1);
    if (k > 0) {
        exe[k] = 0;
        std::string dir(exe);
        dir.erase(dir.find_last_of('/') == std::string::npos ? 0 : dir.find_last_of('/'));
        attempt(dir + "/../crackling_amd/libissl_hip.so");
        attempt(dir + "/libissl_hip.so");
        attempt(dir + "/../lib/libissl_hip.so");
    }
    attempt("libissl_hip.so");
    if (!h) {
        std::fprintf(stderr, "isslScoreOfftargets: cannot load libissl_hip.so (set ISSL_LIBRARY):\n");
        for (const auto &t : tried) std::fprintf(stderr, "  %s\n", t.c_str());
        return false;
    }
#define X(f)                                                                                                              \
    api.f = reinterpret_cast<decltype(api.f)>(::dlsym(h, #f));                                                            \
    if (!api.f) { std::fprintf(stderr, "isslScoreOfftargets: libissl_hip.so lacks %s (another version of the library?)\n", #f); return false; }
    ISSL_CLI_API(X)
#undef X
    if (api.issl_abi_version() != ISSL_ABI_VERSION) {
        std::fprintf(stderr, "isslScoreOfftargets: libissl_hip.so has ABI %d, this executable was built for %d\n", api.issl_abi_version(), ISSL_ABI_VERSION);
        return false;
    }
    api.load_ms = now_ms() - t0;
    return true;
}

std::string last_error(const char *fallback)
{
    const char *e = api.issl_last_error ? api.issl_last_error() : nullptr;
    return (e && e[0]) ? e : fallback;
}

struct DeviceChoice {
    bool all = false;
    std::vector<int> list;
    int single = 0;
    bool single_given = false;
};

DeviceChoice device_choice_from_env()
{
    DeviceChoice c;
    if (const char *d = std::getenv("ISSL_DEVICE")) {
        c.single = std::atoi(d);
        c.single_given = true;
    }
    if (const char *list = std::getenv("ISSL_DEVICES")) {
        if (!std::strcmp(list, "all")) {
            c.all = true;
        } else {
            for (const char *p = list; *p;) {
                char *end = nullptr;
                const long d = std::strtol(p, &end, 10);
                if (end == p) break;
                c.list.push_back(static_cast<int>(d));
                p = (*end == ',') ? end + 1 : end;
            }
        }
    }
    return c;
}

// An index kept ready for scoring: host view + HBM image(s).
struct Resident {
    issl_index *idx = nullptr;
    issl_node *node = nullptr; // when several devices are used
    issl_header hdr{};
    double load_ms = 0, upload_ms = 0, broadcast_ms = 0;
    int n_devices = 1, used_rccl = 0;
    // identity of the file it was loaded from (a rebuilt index must not be served stale from HBM)
    off_t size = 0;
    struct timespec mtime{};
    ino_t ino = 0;
    dev_t dev = 0;
    // resident server: what the image takes of every device it sits on, and when it was last asked for
    size_t image_bytes = 0;
    unsigned long long last_used = 0;
};

void release(Resident &r)
{
    if (r.node) api.issl_node_close(r.node);
    if (r.idx) api.issl_index_close(r.idx);
    r = Resident{};
}

// A process that scores ONE page and exits pays for the sorted image layout (0.36 ns per site at upload) and earns
// ~0.5 fs per site and guide from the pruned scan it enables: worth it from ~0.7 M guides up, whatever the index
// size.  The resident server keeps its indexes and always builds it.
constexpr long long kSortedLayoutPaysFromGuides = 750000;

// Open (host side only: file mapping, header, validation).
bool open_index(const char *issl_path, Resident &r, std::string &err)
{
    const double t0 = now_ms();
    if (api.issl_index_open(issl_path, &r.idx)) { err = last_error("cannot open index"); return false; }
    api.issl_index_header(r.idx, &r.hdr);
    r.load_ms = now_ms() - t0;
    return true;
}

// Upload of an opened index.  Several devices only when ISSL_DEVICES asks for them.  one_shot_guides: guides of the only
// page this process will score, or -1 (resident server).
bool upload_index(const DeviceChoice &dc, Resident &r, std::string &err, long long one_shot_guides = -1)
{
    if (one_shot_guides >= 0 && one_shot_guides < kSortedLayoutPaysFromGuides && !std::getenv("ISSL_SORTED_LAYOUT"))
        (void)api.issl_index_set_option(r.idx, "sorted_layout", "0");
    // (the page is known: its scoring workspace is set up beside the upload)
    if (one_shot_guides > 0 && !std::getenv("ISSL_EXPECT_GUIDES"))
        (void)api.issl_index_set_option(r.idx, "expect_guides", std::to_string(one_shot_guides).c_str());
    const bool all = dc.all;
    const double t1 = now_ms();
    if (all || dc.list.size() > 1) {
        if (api.issl_node_create(r.idx, all ? nullptr : dc.list.data(), static_cast<int>(dc.list.size()), &r.node)) {
            err = last_error("cannot set up the devices");
            release(r);
            return false;
        }
        issl_node_info inf;
        api.issl_node_get_info(r.node, &inf);
        r.upload_ms = inf.ms_upload;
        r.broadcast_ms = inf.ms_broadcast;
        r.n_devices = inf.n_devices;
        r.used_rccl = inf.used_rccl;
    } else {
        const int dev = dc.list.size() == 1 ? dc.list[0] : dc.single;
        if (api.issl_index_upload(r.idx, dev)) {
            err = last_error("cannot upload index");
            release(r);
            return false;
        }
        r.upload_ms = now_ms() - t1;
    }
    (void)api.issl_index_device_bytes(r.idx, &r.image_bytes); // (of the layout the upload settled on)
    return true;
}

// Open + upload (the resident server).  before_upload: called with the opened index (host side only) before any device
// memory is taken: the server makes room there.
bool make_resident(const char *issl_path, const DeviceChoice &dc, Resident &r, std::string &err,
                   const std::function<void(issl_index *)> &before_upload = nullptr)
{
    if (!open_index(issl_path, r, err)) return false;
    if (before_upload) before_upload(r.idx);
    return upload_index(dc, r, err, -1);
}

struct Request {
    std::string issl, query, method_arg, verdict_path;
    int max_dist = 4;
    double threshold = 75.0;
};

// Where the wall time of one invocation went (ISSL_TIMING).
struct Stages {
    double query_ms = 0, score_ms = 0, format_ms = 0, write_ms = 0;
};

// All of [p, p + n) to a descriptor of any kind.
bool write_fd(int fd, const char *p, size_t n)
{
    while (n) {
        const ssize_t k = ::write(fd, p, n);
        if (k < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        p += k;
        n -= static_cast<size_t>(k);
    }
    return true;
}

// The scored page as text: the pieces issl_format_scores made, in order.
struct Text {
    issl_span *spans = nullptr;
    size_t n_spans = 0;
    size_t bytes() const { size_t b = 0; for (size_t i = 0; i < n_spans; ++i) b += spans[i].len; return b; }
    ~Text() { if (spans) api.issl_free_spans(spans, n_spans); }
    Text() = default;
    Text(const Text &) = delete;
    Text &operator=(const Text &) = delete;
};

// Score one request against a resident index: the TSV text of isslScoreOfftargets.cpp:514-527 in `out`.
// guides_in / n_in: the query file's guides when the caller has read it already (they are released here), else null.
bool score_request(Resident &r, const Request &q, Text &out, std::string &err, Stages &st, std::string &stats_json,
                   uint64_t *guides_in = nullptr, size_t n_in = 0)
{
    const int method = api.issl_method_from_string(q.method_arg.c_str()); // :121-143
    uint64_t *guides = guides_in;
    size_t n = n_in;
    double t0 = now_ms();
    if (!guides && api.issl_read_query_file(q.query.c_str(), r.hdr.seq_len, &guides, &n)) { // :275-305
        err = last_error("cannot read query file");
        return false;
    }
    st.query_ms += now_ms() - t0;
    std::vector<double> mit(n), cfd(n);
    t0 = now_ms();
    const int rc = r.node ? api.issl_node_score(r.node, guides, n, q.max_dist, q.threshold, method, mit.data(), cfd.data())
                          : api.issl_score(r.idx, guides, n, q.max_dist, q.threshold, method, mit.data(), cfd.data());
    st.score_ms = now_ms() - t0;
    if (rc) {
        err = last_error("scoring failed");
        api.issl_free(guides);
        return false;
    }
    // :514-527, in input order: formatted by the library on several threads with its own exact "%f"
    t0 = now_ms();
    if (api.issl_format_scores(guides, mit.data(), cfd.data(), n, r.hdr.seq_len, method, 0, &out.spans, &out.n_spans)) {
        err = last_error("formatting failed");
        api.issl_free(guides);
        return false;
    }
    st.format_ms = now_ms() - t0;
    char seq[40];
    if (!q.verdict_path.empty()) { // Crackling.py:780-835, fused
        std::vector<uint8_t> verdict(n);
        std::string text;
        text.reserve(n * 24);
        if (api.issl_verdicts(mit.data(), cfd.data(), n, q.threshold, q.method_arg.c_str(), verdict.data())) {
            err = last_error("thresholding failed");
            api.issl_free(guides);
            return false;
        }
        for (size_t i = 0; i < n; ++i) {
            if (verdict[i] == ISSL_VERDICT_NONE) continue; // the caller leaves such guides untouched
            api.issl_decode_guide(guides[i], r.hdr.seq_len, seq);
            text.append(seq);
            text.append(verdict[i] == ISSL_VERDICT_ACCEPTED ? "\t1\n" : "\t0\n");
        }
        FILE *vf = std::fopen(q.verdict_path.c_str(), "w");
        if (!vf || std::fwrite(text.data(), 1, text.size(), vf) != text.size() || std::fclose(vf) != 0) {
            err = "cannot write verdict file '" + q.verdict_path + "': " + std::strerror(errno);
            api.issl_free(guides);
            return false;
        }
    }
    api.issl_free(guides);
    char buf[512];
    if (r.node) {
        std::snprintf(buf, sizeof buf, "\"guides\": %zu, \"devices\": %d, \"rccl\": %d, \"broadcast_ms\": %.3f", n, r.n_devices,
                      r.used_rccl, r.broadcast_ms);
    } else {
        issl_stats is;
        api.issl_last_stats(r.idx, &is);
        std::snprintf(buf, sizeof buf, "\"guides\": %zu, \"devices\": 1, \"scan_ms\": %.3f, \"replay_ms\": %.3f, \"candidates\": %llu, "
                      "\"hits\": %llu, \"pruned\": %llu", n, is.ms_scan, is.ms_replay, (unsigned long long)is.candidates,
                      (unsigned long long)is.hits, (unsigned long long)is.pruned);
    }
    stats_json = buf;
    return true;
}

// The ISSL_TIMING line.  open / upload are the index's own (zero when a resident server had it already).
std::string timing_json(const Resident &r, const Stages &st, const std::string &stats_json, bool resident_hit, double extra_start_ms,
                        double runtime_ms, double total_ms)
{
    char buf[1024];
    std::snprintf(buf, sizeof buf,
                  "{%s, \"resident\": %s, \"start_ms\": %.3f, \"open_ms\": %.3f, \"load_ms\": %.3f, \"runtime_ms\": %.3f, \"upload_ms\": %.3f, "
                  "\"query_ms\": %.3f, \"score_ms\": %.3f, \"format_ms\": %.3f, \"write_ms\": %.3f, \"total_ms\": %.3f}",
                  stats_json.c_str(), resident_hit ? "true" : "false", extra_start_ms, r.load_ms, r.load_ms, runtime_ms, r.upload_ms,
                  st.query_ms, st.score_ms, st.format_ms, st.write_ms, total_ms);
    return buf;
}

// ---- resident mode: tiny line protocol over a unix stream socket ---------------------------------------------
//   request : "SCORE\t<issl>\t<query>\t<maxDist>\t<threshold>\t<method>[\t<verdict file>]\n"   |  "QUIT\n" | "STATUS\n"
//             a SCORE request may carry ONE descriptor (SCM_RIGHTS): the client's stdout when that is a regular file
//   response: "OK <nbytes> <timing json>\n" + nbytes of TSV               |  "ERR <message>\n"
//             "OKFD <nbytes> <timing json>\n": the nbytes went to the descriptor the request carried, none follow

// Sockets only: a peer that went away gives EPIPE instead of a signal (the server must outlive its clients).
bool write_all(int fd, const char *p, size_t n)
{
    while (n) {
        const ssize_t k = ::send(fd, p, n, MSG_NOSIGNAL);
        if (k < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        p += k;
        n -= static_cast<size_t>(k);
    }
    return true;
}

// The request line of a client and the descriptor it may have sent along (-1: none).  The line is short and the client
// sends nothing behind it, so whole chunks are received at once; ancillary data arrives with the first byte.
bool read_request(int fd, std::string &line, int &passed_fd)
{
    line.clear();
    passed_fd = -1;
    char data[4096];
    while (true) {
        alignas(cmsghdr) char ctl[CMSG_SPACE(sizeof(int))];
        iovec iov{data, sizeof data};
        msghdr msg{};
        msg.msg_iov = &iov;
        msg.msg_iovlen = 1;
        msg.msg_control = ctl;
        msg.msg_controllen = sizeof ctl;
        const ssize_t k = ::recvmsg(fd, &msg, MSG_CMSG_CLOEXEC);
        if (k < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        for (cmsghdr *c = CMSG_FIRSTHDR(&msg); c; c = CMSG_NXTHDR(&msg, c))
            if (c->cmsg_level == SOL_SOCKET && c->cmsg_type == SCM_RIGHTS && c->cmsg_len >= CMSG_LEN(sizeof(int))) {
                int got;
                std::memcpy(&got, CMSG_DATA(c), sizeof got);
                if (passed_fd >= 0) ::close(passed_fd);
                passed_fd = got;
            }
        if (k == 0) return !line.empty();
        for (ssize_t i = 0; i < k; ++i) {
            if (data[i] == '\n') return true;
            line.push_back(data[i]);
        }
        if (line.size() > 65536) return false;
    }
}

bool read_line(int fd, std::string &line)
{
    line.clear();
    char c;
    while (true) {
        const ssize_t k = ::read(fd, &c, 1);
        if (k == 0) return !line.empty();
        if (k < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        if (c == '\n') return true;
        line.push_back(c);
        if (line.size() > 65536) return false;
    }
}

int unix_socket(const char *path, sockaddr_un &addr)
{
    if (std::strlen(path) >= sizeof(addr.sun_path)) return -1;
    const int fd = ::socket(AF_UNIX, SOCK_STREAM, 0);
    if (fd < 0) return -1;
    std::memset(&addr, 0, sizeof addr);
    addr.sun_family = AF_UNIX;
    std::strncpy(addr.sun_path, path, sizeof(addr.sun_path) - 1);
    return fd;
}

int serve(const char *sock_path)
{
    if (!load_api()) return 1;
    sockaddr_un addr;
    const int lfd = unix_socket(sock_path, addr);
    if (lfd < 0) { std::fprintf(stderr, "cannot create socket %s\n", sock_path); return 1; }
    std::signal(SIGPIPE, SIG_IGN); // a client killed mid-answer must not take the server (and its resident indexes) down
    struct stat old;
    if (::lstat(sock_path, &old) == 0) { // a stale socket of an earlier server may be replaced, nothing else
        if (!S_ISSOCK(old.st_mode)) {
            std::fprintf(stderr, "%s exists and is not a socket: not touching it\n", sock_path);
            return 1;
        }
        ::unlink(sock_path);
    }
    const mode_t mask = ::umask(0177); // the socket is created 0600: only this user may send requests
    const bool bound = ::bind(lfd, reinterpret_cast<sockaddr *>(&addr), sizeof addr) == 0;
    ::umask(mask);
    if (!bound || ::listen(lfd, 16) != 0) {
        std::fprintf(stderr, "cannot listen on %s: %s\n", sock_path, std::strerror(errno));
        return 1;
    }
    std::fprintf(stderr, "isslScoreOfftargets: serving on %s\n", sock_path);
    const DeviceChoice dc = device_choice_from_env();
    long request_timeout_s = 10;
    if (const char *t = std::getenv("ISSL_SERVER_TIMEOUT_S")) request_timeout_s = std::max(1L, std::atol(t));
    // Resident indexes, least recently used first out: before an index is uploaded, others are released until the image of
    // its preferred (fastest) layout, the temporaries of its construction and a working reserve fit the free HBM -- an
    // upload that found the HBM full of idle indexes would otherwise settle for a slower layout or fail.  The index a
    // request is for is never evicted for itself; one request is served at a time, so none is in use meanwhile.
    // ISSL_SERVER_HBM_BUDGET (bytes) caps what the server lets its indexes take together (tests, shared GPUs).
    std::map<std::string, Resident> cache;
    unsigned long long tick = 0, evictions = 0;
    size_t budget = 0;
    if (const char *b = std::getenv("ISSL_SERVER_HBM_BUDGET")) budget = static_cast<size_t>(std::strtoull(b, nullptr, 10));
    std::vector<int> mem_devices = dc.list; // every device an index of this server is uploaded to
    if (mem_devices.empty()) mem_devices.push_back(dc.single);
    auto make_room = [&](issl_index *incoming) {
        size_t preferred = 0;
        issl_header h{};
        if (api.issl_index_device_bytes(incoming, &preferred) || api.issl_index_header(incoming, &h)) return;
        // Room asked for: the image of the fastest layout + the temporaries of its construction + the scoring workspace (hit
        // slots: up to 8 GiB).  An upload settles for the smallest layout (52 B/site, 8 B/site of temporaries) when the
        // device cannot hold the fastest even when it is empty -- evicting beyond what THAT one needs buys nothing.
        const size_t reserve = size_t(10) << 30;
        const size_t want_fast = preferred + 8 * h.n_sites + reserve, want_small = (52 + 8) * h.n_sites + reserve;
        while (!cache.empty()) {
            size_t free_b = ~size_t(0), total_b = ~size_t(0), held = 0;
            for (int d : mem_devices) { // the tightest device decides
                size_t f = 0, t = 0;
                if (api.issl_device_memory(d, &f, &t)) return;
                free_b = std::min(free_b, f);
                total_b = std::min(total_b, t);
            }
            for (const auto &kv : cache) held += kv.second.image_bytes;
            if (budget) free_b = std::min(free_b, budget > held ? budget - held : size_t(0));
            const size_t want = want_fast <= total_b ? want_fast : std::min(want_fast, want_small);
            if (free_b >= want) return;
            auto lru = cache.begin();
            for (auto it = cache.begin(); it != cache.end(); ++it)
                if (it->second.last_used < lru->second.last_used) lru = it;
            std::fprintf(stderr, "isslScoreOfftargets: releasing %s (%.1f GB, least recently used) to make room\n", lru->first.c_str(),
                         lru->second.image_bytes / 1e9);
            release(lru->second);
            cache.erase(lru);
            ++evictions;
        }
    };
    auto json_string = [](const std::string &in) { // a path may hold quotes, backslashes, control characters
        std::string out = "\"";
        for (unsigned char ch : in) {
            if (ch == '"' || ch == '\\') { out += '\\'; out += static_cast<char>(ch); }
            else if (ch < 0x20) { char buf[8]; std::snprintf(buf, sizeof buf, "\\u%04x", ch); out += buf; }
            else out += static_cast<char>(ch);
        }
        return out + "\"";
    };
    bool quit = false;
    while (!quit) {
        const int fd = ::accept(lfd, nullptr, nullptr);
        if (fd < 0) {
            if (errno == EINTR) continue;
            break;
        }
        // one client at a time: a client that connects and then says nothing may hold the server for 10 s, no longer
        // (ISSL_SERVER_TIMEOUT_S, read once at start-up, overrides); one that stops reading its answer, for six times that
        const struct timeval rcv_to = {request_timeout_s, 0}, snd_to = {6 * request_timeout_s, 0};
        ::setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &rcv_to, sizeof rcv_to);
        ::setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &snd_to, sizeof snd_to);
        std::string line;
        int out_fd = -1; // the client's stdout, when it sent it along
        const double t_req = now_ms();
        if (read_request(fd, line, out_fd)) {
            if (line == "QUIT") {
                write_all(fd, "OK 0 {}\n", 8);
                quit = true;
            } else if (line == "STATUS") { // what is resident, in order of last use
                std::vector<const std::pair<const std::string, Resident> *> order;
                for (const auto &kv : cache) order.push_back(&kv);
                std::sort(order.begin(), order.end(), [](auto a, auto b) { return a->second.last_used < b->second.last_used; });
                std::string body = "{\"evictions\": " + std::to_string(evictions) + ", \"resident\": [";
                for (size_t i = 0; i < order.size(); ++i)
                    body += std::string(i ? ", " : "") + "{\"issl\": " + json_string(order[i]->first) + ", \"image_bytes\": " +
                            std::to_string(order[i]->second.image_bytes) + "}";
                body += "]}\n";
                const std::string head = "OK " + std::to_string(body.size()) + " {}\n";
                write_all(fd, head.data(), head.size()) && write_all(fd, body.data(), body.size());
            } else {
                std::vector<std::string> f;
                size_t a = 0;
                while (true) {
                    const size_t b = line.find('\t', a);
                    f.push_back(line.substr(a, b == std::string::npos ? std::string::npos : b - a));
                    if (b == std::string::npos) break;
                    a = b + 1;
                }
                std::string err, stats;
                Text out;
                Stages stg;
                bool ok = false, hit = false;
                const Resident *scored = nullptr;
                if ((f.size() == 6 || f.size() == 7) && f[0] == "SCORE") {
                    Request q;
                    if (f.size() == 7) q.verdict_path = f[6];
                    q.issl = f[1]; q.query = f[2]; q.max_dist = std::atoi(f[3].c_str());
                    q.threshold = std::atof(f[4].c_str()); q.method_arg = f[5];
                    struct stat st;
                    if (::stat(q.issl.c_str(), &st) != 0) {
                        err = "cannot open index file '" + q.issl + "': " + std::strerror(errno);
                    } else {
                        auto it = cache.find(q.issl);
                        hit = it != cache.end() && it->second.size == st.st_size && it->second.ino == st.st_ino &&
                                   it->second.dev == st.st_dev && it->second.mtime.tv_sec == st.st_mtim.tv_sec &&
                                   it->second.mtime.tv_nsec == st.st_mtim.tv_nsec;
                        if (it != cache.end() && !hit) { release(it->second); cache.erase(it); it = cache.end(); }
                        if (it == cache.end()) {
                            Resident r;
                            if (make_resident(q.issl.c_str(), dc, r, err, make_room)) {
                                r.size = st.st_size;
                                r.mtime = st.st_mtim;
                                r.ino = st.st_ino;
                                r.dev = st.st_dev;
                                it = cache.emplace(q.issl, r).first;
                            }
                        }
                        if (it != cache.end()) {
                            it->second.last_used = ++tick;
                            ok = score_request(it->second, q, out, err, stg, stats);
                            scored = &it->second;
                        }
                    }
                } else {
                    err = "malformed request";
                }
                if (ok) {
                    // The text goes where the client's stdout points when the client handed that over and it is a regular
                    // file (a write to it cannot block on a reader; pipes and terminals get their bytes through the socket,
                    // whose timeouts protect the server); the head line follows, so the client learns of a failed write.
                    struct stat ost;
                    const bool direct = out_fd >= 0 && ::fstat(out_fd, &ost) == 0 && S_ISREG(ost.st_mode);
                    const double t_w = now_ms();
                    bool wrote = true;
                    if (direct)
                        for (size_t i = 0; i < out.n_spans && wrote; ++i) wrote = write_fd(out_fd, out.spans[i].data, out.spans[i].len);
                    if (direct) stg.write_ms = now_ms() - t_w;
                    if (!wrote) {
                        const std::string head = std::string("ERR cannot write to the client's stdout: ") + std::strerror(errno) + "\n";
                        write_all(fd, head.data(), head.size());
                    } else {
                        Resident shown = *scored;
                        if (hit) { shown.load_ms = 0; shown.upload_ms = 0; } // (this request did not pay for them)
                        const std::string tj = timing_json(shown, stg, stats, hit, 0.0, 0.0, now_ms() - t_req);
                        const std::string head = std::string(direct ? "OKFD " : "OK ") + std::to_string(out.bytes()) + " " + tj + "\n";
                        bool sent = write_all(fd, head.data(), head.size());
                        for (size_t i = 0; !direct && i < out.n_spans && sent; ++i) sent = write_all(fd, out.spans[i].data, out.spans[i].len);
                    }
                } else {
                    for (char &c : err) if (c == '\n') c = ' ';
                    const std::string head = "ERR " + err + "\n";
                    write_all(fd, head.data(), head.size());
                }
            }
        }
        if (out_fd >= 0) ::close(out_fd);
        ::close(fd);
    }
    for (auto &kv : cache) release(kv.second);
    ::close(lfd);
    ::unlink(sock_path);
    return 0;
}

// Returns -1 when the server cannot be reached (caller scores locally), else the exit status.
int try_server(const char *sock_path, char **argv, bool timing)
{
    sockaddr_un addr;
    const int fd = unix_socket(sock_path, addr);
    if (fd < 0) return -1;
    if (::connect(fd, reinterpret_cast<sockaddr *>(&addr), sizeof addr) != 0) {
        ::close(fd);
        return -1;
    }
    char issl_abs[PATH_MAX], query_abs[PATH_MAX];
    if (!::realpath(argv[1], issl_abs)) std::snprintf(issl_abs, sizeof issl_abs, "%s", argv[1]);
    if (!::realpath(argv[2], query_abs)) std::snprintf(query_abs, sizeof query_abs, "%s", argv[2]);
    std::string req = std::string("SCORE\t") + issl_abs + "\t" + query_abs + "\t" + argv[3] + "\t" + argv[4] + "\t" +
                      argv[5];
    if (const char *vp = std::getenv("ISSL_VERDICTS")) { // the server writes the file: make the path absolute
        std::string abs = vp;
        char cwd[PATH_MAX];
        if (vp[0] != '/' && ::getcwd(cwd, sizeof cwd)) abs = std::string(cwd) + "/" + vp;
        req += "\t" + abs;
    }
    req += "\n";
    // stdout a regular file (Crackling: `> output`, Crackling.py:767): the descriptor goes with the request and the server
    // writes the text into it -- 41 MB per million guides that then cross no socket and are copied by nobody
    struct stat ost;
    const bool pass_stdout = ::fstat(STDOUT_FILENO, &ost) == 0 && S_ISREG(ost.st_mode) && !std::getenv("ISSL_SERVER_NO_FD");
    bool sent;
    if (pass_stdout) {
        alignas(cmsghdr) char ctl[CMSG_SPACE(sizeof(int))] = {};
        iovec iov{const_cast<char *>(req.data()), req.size()};
        msghdr msg{};
        msg.msg_iov = &iov;
        msg.msg_iovlen = 1;
        msg.msg_control = ctl;
        msg.msg_controllen = sizeof ctl;
        cmsghdr *c = CMSG_FIRSTHDR(&msg);
        c->cmsg_level = SOL_SOCKET;
        c->cmsg_type = SCM_RIGHTS;
        c->cmsg_len = CMSG_LEN(sizeof(int));
        const int out_fd = STDOUT_FILENO;
        std::memcpy(CMSG_DATA(c), &out_fd, sizeof out_fd);
        ssize_t k;
        do k = ::sendmsg(fd, &msg, MSG_NOSIGNAL); while (k < 0 && errno == EINTR);
        sent = k >= 0 && write_all(fd, req.data() + k, req.size() - static_cast<size_t>(k));
    } else {
        sent = write_all(fd, req.data(), req.size());
    }
    std::string head;
    if (!sent || !read_line(fd, head)) {
        ::close(fd);
        return -1;
    }
    if (head.compare(0, 5, "OKFD ") == 0) { // the text is in our stdout already
        if (timing) {
            const char *sp = std::strchr(head.c_str() + 5, ' ');
            if (sp) std::fprintf(stderr, "%s\n", sp + 1);
        }
        ::close(fd);
        return 0;
    }
    if (head.compare(0, 3, "OK ") != 0) {
        std::fprintf(stderr, "%s\n", head.size() > 4 ? head.c_str() + 4 : "scoring failed");
        ::close(fd);
        return 1;
    }
    char *end = nullptr;
    size_t left = std::strtoull(head.c_str() + 3, &end, 10);
    if (timing && end && *end) std::fprintf(stderr, "%s\n", end + 1);
    std::vector<char> buf(1 << 20);
    while (left) {
        const ssize_t k = ::read(fd, buf.data(), left < buf.size() ? left : buf.size());
        if (k <= 0) {
            if (k < 0 && errno == EINTR) continue;
            std::fprintf(stderr, "resident scorer closed the connection early\n");
            ::close(fd);
            return 1;
        }
        if (!write_fd(STDOUT_FILENO, buf.data(), static_cast<size_t>(k))) {
            std::fprintf(stderr, "short write on stdout\n");
            ::close(fd);
            return 1;
        }
        left -= static_cast<size_t>(k);
    }
    ::close(fd);
    return 0;
}

} // namespace

int main(int argc, char **argv)
{
    if (argc == 3 && !std::strcmp(argv[1], "--serve")) return serve(argv[2]);
    if (argc == 3 && !std::strcmp(argv[1], "--stop")) {
        sockaddr_un addr;
        const int fd = unix_socket(argv[2], addr);
        if (fd < 0 || ::connect(fd, reinterpret_cast<sockaddr *>(&addr), sizeof addr) != 0) return 1;
        std::string line;
        const bool ok = write_all(fd, "QUIT\n", 5) && read_line(fd, line);
        ::close(fd);
        return ok ? 0 : 1;
    }
    if (argc == 3 && !std::strcmp(argv[1], "--status")) { // what the resident server holds (JSON on stdout)
        sockaddr_un addr;
        const int fd = unix_socket(argv[2], addr);
        if (fd < 0 || ::connect(fd, reinterpret_cast<sockaddr *>(&addr), sizeof addr) != 0) return 1;
        std::string line;
        bool ok = write_all(fd, "STATUS\n", 7) && read_line(fd, line) && line.rfind("OK ", 0) == 0;
        if (ok) {
            std::string body(static_cast<size_t>(std::atol(line.c_str() + 3)), '\0');
            size_t got = 0;
            while (got < body.size()) {
                const ssize_t k = ::read(fd, &body[got], body.size() - got);
                if (k <= 0) { ok = false; break; }
                got += static_cast<size_t>(k);
            }
            if (ok) std::fwrite(body.data(), 1, body.size(), stdout);
        }
        ::close(fd);
        return ok ? 0 : 1;
    }
    if (argc < 6) { // the reference checks argc < 4 and then reads argv[4], argv[5] regardless (:93,112,121)
        std::fprintf(stderr, "Usage: %s [issltable] [query file] [max distance] [score-threshold] [score-method]\n",
                     argv[0]);
        return 1;
    }
    const double t_main = now_ms();
    const bool timing = std::getenv("ISSL_TIMING") != nullptr;
    if (const char *sock = std::getenv("ISSL_SERVER")) {
        const int rc = try_server(sock, argv, timing);
        if (rc >= 0) return rc;
    }
    if (!load_api()) return 1;
    Request q;
    q.issl = argv[1];
    q.query = argv[2];
    q.max_dist = std::atoi(argv[3]);     // :109
    q.threshold = std::atof(argv[4]);    // :112
    q.method_arg = argv[5];
    if (const char *vp = std::getenv("ISSL_VERDICTS")) q.verdict_path = vp;
    const DeviceChoice dc = device_choice_from_env();
    // The device runtime takes a few hundred milliseconds to come up; it does so on a thread of its own while this one
    // maps the index and packs the query (one device only: a node sets its devices up itself).
    std::atomic<double> runtime_ms{0.0};
    std::thread warm;
    if (!dc.all && dc.list.size() <= 1 && !std::getenv("ISSL_NO_WARMUP"))
        warm = std::thread([&] {
            const double t0 = now_ms();
            size_t f = 0, t = 0;
            (void)api.issl_device_memory(dc.list.size() == 1 ? dc.list[0] : dc.single, &f, &t); // (an error shows again at the upload)
            runtime_ms = now_ms() - t0;
        });
    auto fail = [&](const std::string &msg) {
        std::fprintf(stderr, "%s\n", msg.c_str());
        if (warm.joinable()) warm.join();
        return 1;
    };
    Resident r;
    std::string err, stats;
    Stages stg;
    // same order of checks as the reference: index file first, then the query file (:152-294), before any device work
    if (!open_index(argv[1], r, err)) return fail(err);
    uint64_t *guides = nullptr;
    size_t n_guides = 0;
    double t0 = now_ms();
    if (api.issl_read_query_file(argv[2], r.hdr.seq_len, &guides, &n_guides)) {
        const std::string msg = last_error("cannot read query file");
        release(r);
        return fail(msg);
    }
    stg.query_ms = now_ms() - t0;
    if (warm.joinable()) warm.join();
    if (!upload_index(dc, r, err, static_cast<long long>(n_guides))) {
        api.issl_free(guides);
        return fail(err);
    }
    Text out;
    if (!score_request(r, q, out, err, stg, stats, guides, n_guides)) {
        release(r);
        return fail(err);
    }
    t0 = now_ms();
    for (size_t i = 0; i < out.n_spans; ++i)
        if (!write_fd(STDOUT_FILENO, out.spans[i].data, out.spans[i].len)) {
            std::fprintf(stderr, "short write on stdout\n");
            release(r);
            return 1;
        }
    stg.write_ms = now_ms() - t0;
    if (timing) std::fprintf(stderr, "%s\n", timing_json(r, stg, stats, false, api.load_ms, runtime_ms, now_ms() - t_main).c_str());
    // Everything is written; what is left is returning 45 GB of HBM and tearing the runtime down, which the driver does
    // for a process that ends anyway: leave at once (ISSL_TIDY_EXIT=1: release everything first, for leak checkers).
    if (std::getenv("ISSL_TIDY_EXIT")) {
        release(r);
        return 0;
    }
    std::fflush(stderr);
    ::_exit(0);
}
