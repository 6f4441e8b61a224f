// bin/isslScoreOfftargets -- drop-in for the reference scorer process
// (src/ISSL/isslScoreOfftargets.cpp:91-530, invoked by src/crackling/Crackling.py:767-778):
//
//   isslScoreOfftargets [issltable] [query file] [max distance] [score-threshold] [score-method]
//
// stdout carries data only ("<20-mer>\t<MIT>\t<CFD>\n" per guide, input order, "-1" for a score
// that was not requested); diagnostics go to stderr; exit status 1 on any error.
// Optional environment (the five positionals stay untouched so that Crackling needs no change):
//   ISSL_DEVICE=<n>   HIP device to use (default 0)
//   ISSL_DEVICES=all | <a,b,...>   several GPUs of the node: index image broadcast over RCCL/xGMI, guides sharded
//                     (default: one device; all visible devices when the query file holds >= 2^20 guides)
//   ISSL_TIMING=1     one JSON line with load/upload/score timings on stderr
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/issl_hip.h"

static double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

static int fail(const char *what)
{
    std::fprintf(stderr, "%s\n", issl_last_error()[0] ? issl_last_error() : what);
    return 1;
}

int main(int argc, char **argv)
{
    if (argc < 6) { // the reference checks argc < 4 and then reads argv[4], argv[5] regardless (:93,112,121)
        std::fprintf(stderr, "Usage: %s [issltable] [query file] [max distance] [score-threshold] [score-method]\n",
                     argv[0]);
        return 1;
    }
    const int max_dist = std::atoi(argv[3]);      // :109
    const double threshold = std::atof(argv[4]);  // :112
    const int method = issl_method_from_string(argv[5]);
    const bool want_mit = method == ISSL_METHOD_MIT || method == ISSL_METHOD_AND || method == ISSL_METHOD_OR ||
                          method == ISSL_METHOD_AVG;
    const bool want_cfd = method == ISSL_METHOD_CFD || method == ISSL_METHOD_AND || method == ISSL_METHOD_OR ||
                          method == ISSL_METHOD_AVG;
    const char *dev_env = std::getenv("ISSL_DEVICE");
    const int device = dev_env ? std::atoi(dev_env) : 0;
    const bool timing = std::getenv("ISSL_TIMING") != nullptr;

    const double t0 = now_ms();
    issl_index *idx = nullptr;
    if (issl_index_open(argv[1], &idx)) return fail("cannot open index");
    issl_header hdr;
    issl_index_header(idx, &hdr);
    uint64_t *guides = nullptr;
    size_t n = 0;
    if (issl_read_query_file(argv[2], hdr.seq_len, &guides, &n)) return fail("cannot read query file");
    const double t1 = now_ms();
    // device selection
    std::vector<int> devices;
    bool all_devices = false;
    if (const char *list = std::getenv("ISSL_DEVICES")) {
        if (!std::strcmp(list, "all")) {
            all_devices = true;
        } else {
            for (const char *p = list; *p;) {
                char *end = nullptr;
                const long d = std::strtol(p, &end, 10);
                if (end == p) break;
                devices.push_back(static_cast<int>(d));
                p = (*end == ',') ? end + 1 : end;
            }
        }
    } else if (!dev_env && n >= (size_t(1) << 20)) {
        all_devices = true;
    }
    std::vector<double> mit(n), cfd(n);
    issl_node *node = nullptr;
    double t2;
    if (all_devices || devices.size() > 1) {
        if (issl_node_create(idx, all_devices ? nullptr : devices.data(), static_cast<int>(devices.size()), &node))
            return fail("cannot set up the devices");
        t2 = now_ms();
        if (issl_node_score(node, guides, n, max_dist, threshold, method, mit.data(), cfd.data()))
            return fail("scoring failed");
    } else {
        const int dev = devices.size() == 1 ? devices[0] : device;
        if (issl_index_upload(idx, dev)) return fail("cannot upload index");
        t2 = now_ms();
        if (issl_score(idx, guides, n, max_dist, threshold, method, mit.data(), cfd.data())) return fail("scoring failed");
    }
    const double t3 = now_ms();

    // :514-527
    std::vector<char> out;
    out.reserve(n * 48 + 16);
    char seq[40], line[128];
    for (size_t i = 0; i < n; ++i) {
        issl_decode_guide(guides[i], hdr.seq_len, seq);
        int k = std::snprintf(line, sizeof line, "%s\t", seq);
        if (want_mit) k += std::snprintf(line + k, sizeof line - k, "%f\t", mit[i]);
        else k += std::snprintf(line + k, sizeof line - k, "-1\t");
        if (want_cfd) k += std::snprintf(line + k, sizeof line - k, "%f\n", cfd[i]);
        else k += std::snprintf(line + k, sizeof line - k, "-1\n");
        out.insert(out.end(), line, line + k);
    }
    if (!out.empty() && std::fwrite(out.data(), 1, out.size(), stdout) != out.size()) {
        std::fprintf(stderr, "short write on stdout\n");
        return 1;
    }
    std::fflush(stdout);
    if (timing && node) {
        issl_node_info inf;
        issl_node_get_info(node, &inf);
        std::fprintf(stderr,
                     "{\"guides\": %zu, \"devices\": %d, \"rccl\": %d, \"load_ms\": %.3f, \"upload_ms\": %.3f, "
                     "\"broadcast_ms\": %.3f, \"score_ms\": %.3f}\n",
                     n, inf.n_devices, inf.used_rccl, t1 - t0, inf.ms_upload, inf.ms_broadcast, inf.ms_last_score);
    } else if (timing) {
        issl_stats st;
        issl_last_stats(idx, &st);
        std::fprintf(stderr,
                     "{\"guides\": %zu, \"load_ms\": %.3f, \"upload_ms\": %.3f, \"score_ms\": %.3f, \"scan_ms\": %.3f, "
                     "\"replay_ms\": %.3f, \"candidates\": %llu, \"hits\": %llu}\n",
                     n, t1 - t0, t2 - t1, t3 - t2, st.ms_scan, st.ms_replay, (unsigned long long)st.candidates,
                     (unsigned long long)st.hits);
    }
    issl_free(guides);
    if (node) issl_node_close(node);
    issl_index_close(idx);
    return 0;
}
