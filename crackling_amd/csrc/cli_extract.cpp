// bin/extractOfftargets -- counterpart of the reference console script `extractOfftargets`
// (src/crackling/utils/extractOfftargets.py:248-296, setup.py:28):
//
//   extractOfftargets <output> <input FASTA ...| input directory> [--maxOpenFiles N] [--threads N]
//
// Writes every N20 site next to a PAM (both strands, the reference's two patterns), one per line, sorted, duplicates
// kept -- the input of isslCreateIndex.  The two options of the reference are accepted and ignored (there are no
// intermediate files and no process pool).  ISSL_DEVICE selects the GPU.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <string>
#include <sys/stat.h>
#include <vector>

#include "../../include/issl_hip.h"

int main(int argc, char **argv)
{
    std::vector<std::string> pos;
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--maxOpenFiles") || !std::strcmp(argv[i], "--threads")) {
            ++i; // value ignored
            continue;
        }
        if (!std::strncmp(argv[i], "--maxOpenFiles=", 15) || !std::strncmp(argv[i], "--threads=", 10)) continue;
        pos.push_back(argv[i]);
    }
    if (pos.size() < 2) {
        std::fprintf(stderr, "usage: %s output inputs [inputs ...] [--maxOpenFiles N] [--threads N]\n", argv[0]);
        return 2; // argparse's exit status for a usage error
    }
    std::vector<std::string> inputs(pos.begin() + 1, pos.end());
    struct stat st;
    if (inputs.size() == 1 && ::stat(inputs[0].c_str(), &st) == 0 && S_ISDIR(st.st_mode)) { // :204-210
        const std::string dir = inputs[0];
        inputs.clear();
        if (DIR *d = ::opendir(dir.c_str())) {
            while (dirent *e = ::readdir(d)) {
                if (e->d_name[0] == '.') continue;
                inputs.push_back(dir + "/" + e->d_name);
            }
            ::closedir(d);
        }
        std::sort(inputs.begin(), inputs.end());
    }
    std::vector<const char *> ptrs;
    for (auto &s : inputs) ptrs.push_back(s.c_str());
    const char *dev = std::getenv("ISSL_DEVICE");
    uint64_t n = 0;
    if (issl_extract_offtargets(ptrs.data(), static_cast<int>(ptrs.size()), pos[0].c_str(), dev ? std::atoi(dev) : 0, &n)) {
        std::fprintf(stderr, "%s\n", issl_last_error());
        return 1;
    }
    std::fprintf(stderr, "Processing completed. Found %llu targets.\n", (unsigned long long)n);
    return 0;
}
