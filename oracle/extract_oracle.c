/*
 * extract_oracle.c -- CPU ORACLE for the off-target extraction step.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE (only tests/ and measurement scripts may use it).
 * Restates /root/reference/src/crackling/utils/extractOfftargets.py:
 *   :23-24   the two lookahead patterns  [ACG][ACGT]{19}[ACGT][AG]G   and   C[CT][ACGT][ACGT]{19}[TGC]
 *   :27-61   a multi-FASTA file is cut into records at lines starting with '>'; sequence lines are stripped,
 *            upper-cased and concatenated
 *   :97-110  every (overlapping) match contributes the first 20 characters of the 23-character match -- as they
 *            are on the forward pattern, reverse-complemented (Helpers.py:7-10) on the reverse pattern
 *   :112-191 all sites, one per line, sorted (duplicates kept)
 * Pinned by tests/golden/extract/ (made by oracle/make_golden_extract.py from that Python code).
 */
#include <ctype.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int is_acgt(char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

static int cmp20(const void *a, const void *b) { return memcmp(a, b, 20); }

static void scan_record(const char *s, size_t n, char **out, size_t *cnt, size_t *cap)
{
    for (size_t i = 0; i + 23 <= n; i++) {
        int body = 1;
        for (int k = 1; k <= 20 && body; k++) body = is_acgt(s[i + k]); /* chars 1..20 are [ACGT] in both patterns */
        if (!body) continue;
        /* forward: [ACG] [ACGT]{19} [ACGT] [AG] G */
        int fwd = (s[i] == 'A' || s[i] == 'C' || s[i] == 'G') && (s[i + 21] == 'A' || s[i + 21] == 'G') && s[i + 22] == 'G';
        /* reverse: C [CT] [ACGT] [ACGT]{19} [TGC] */
        int rev = s[i] == 'C' && (s[i + 1] == 'C' || s[i + 1] == 'T') && is_acgt(s[i + 21]) &&
                  (s[i + 22] == 'T' || s[i + 22] == 'G' || s[i + 22] == 'C');
        for (int pass = 0; pass < 2; pass++) {
            if (!(pass == 0 ? fwd : rev)) continue;
            if (*cnt == *cap) {
                *cap = *cap ? *cap * 2 : 1024;
                *out = (char *)realloc(*out, *cap * 20);
            }
            char *dst = *out + *cnt * 20;
            if (pass == 0) {
                memcpy(dst, s + i, 20);
            } else {
                for (int k = 0; k < 20; k++) {
                    char c = s[i + 19 - k];
                    dst[k] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
                }
            }
            (*cnt)++;
        }
    }
}

/* fasta: bytes of one FASTA / multi-FASTA / plain text file.  Appends the record's sites to *sites (20 bytes each). */
static void scan_file(const char *fasta, size_t len, char **sites, size_t *cnt, size_t *cap)
{
    char *seq = (char *)malloc(len + 1);
    size_t n = 0, p = 0;
    while (p < len) {
        size_t e = p;
        while (e < len && fasta[e] != '\n') e++;
        size_t a = p, b = e;
        while (a < b && isspace((unsigned char)fasta[a])) a++;
        while (b > a && isspace((unsigned char)fasta[b - 1])) b--;
        if (b > a && fasta[a] == '>') { /* new record */
            scan_record(seq, n, sites, cnt, cap);
            n = 0;
        } else {
            for (size_t k = a; k < b; k++) seq[n++] = (char)toupper((unsigned char)fasta[k]);
        }
        p = e + 1;
    }
    scan_record(seq, n, sites, cnt, cap);
    free(seq);
}

/* Several files -> sorted text (20 chars + '\n' per site), malloc'd. */
char *oracle_extract(const char *const *files, const size_t *lens, int n_files, size_t *out_len)
{
    char *sites = NULL;
    size_t cnt = 0, cap = 0;
    for (int f = 0; f < n_files; f++) scan_file(files[f], lens[f], &sites, &cnt, &cap);
    qsort(sites, cnt, 20, cmp20);
    char *text = (char *)malloc(cnt * 21 + 1);
    for (size_t i = 0; i < cnt; i++) {
        memcpy(text + i * 21, sites + i * 20, 20);
        text[i * 21 + 20] = '\n';
    }
    free(sites);
    *out_len = cnt * 21;
    return text;
}

void oracle_extract_free(void *p) { free(p); }
