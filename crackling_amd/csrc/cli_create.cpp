// bin/isslCreateIndex -- counterpart of the reference index builder
// (src/ISSL/isslCreateIndex.cpp:132-296):
//
//   isslCreateIndex [offtargetSites.txt] [sequence length] [slice width (bits)] [sissltable]
//
// Reads a SORTED list of sites (one per line), collapses consecutive duplicates into
// (signature, occurrences) and writes the .issl bytes the reference writes.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "issl_host.hpp"

int main(int argc, char **argv)
{
    if (argc < 5) {
        std::fprintf(stderr, "Usage: %s [offtargetSites.txt] [sequence length] [slice width (bits)] [sissltable]\n",
                     argv[0]);
        return 1;
    }
    const size_t seq_len = static_cast<size_t>(std::atoi(argv[2]));
    const size_t slice_width = static_cast<size_t>(std::atoi(argv[3]));
    if (seq_len == 0 || seq_len > 32) {
        std::fprintf(stderr, "Sequence length is greater than 32, which is the maximum supported currently\n");
        return 1;
    }
    FILE *fp = std::fopen(argv[1], "rb");
    if (!fp) {
        std::fprintf(stderr, "cannot open '%s'\n", argv[1]);
        return 1;
    }
    std::fseek(fp, 0, SEEK_END);
    const long sz = std::ftell(fp);
    std::fseek(fp, 0, SEEK_SET);
    const size_t line = seq_len + 1;
    if (sz <= 0 || static_cast<size_t>(sz) % line != 0) {
        std::fprintf(stderr, "fileSize: %ld\n", sz);
        std::fprintf(stderr, "Error: file does is not a multiple of the expected line length (%zu)\n", line);
        std::fprintf(stderr, "The sequence length may be incorrect; alternatively, the line endings\n");
        std::fprintf(stderr, "may be something other than LF, or there may be junk at the end of the file.\n");
        return 1;
    }
    std::vector<char> text(static_cast<size_t>(sz));
    if (std::fread(text.data(), text.size(), 1, fp) < 1) {
        std::fprintf(stderr, "Failed to read in file.\n");
        return 1;
    }
    std::fclose(fp);
    const size_t n_lines = text.size() / line;
    std::fprintf(stderr, "Number of sequences: %zu\n", n_lines);
    // host-only code path: this executable does not touch the GPU and does not load the HIP runtime
    issl::HostIndex idx;
    if (idx.build_from_text(text.data(), n_lines, seq_len, slice_width) || idx.write_file(argv[4])) {
        std::fprintf(stderr, "%s\n", issl::get_error());
        return 1;
    }
    std::printf("Done. %llu distinct sites, %llu slices of %llu bits.\n", (unsigned long long)idx.geo.n_sites,
                (unsigned long long)idx.geo.n_slices, (unsigned long long)idx.geo.slice_width);
    return 0;
}
