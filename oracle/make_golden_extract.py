#!/usr/bin/env python3
"""Golden vectors for the off-target extraction step from the REFERENCE Python
(/root/reference/src/crackling/utils/extractOfftargets.py), run in the build container only:

    python oracle/make_golden_extract.py

Writes tests/golden/extract/<name>.fa (inputs, data made here) and <name>.sites.txt (the file the reference's
startMultiprocessing() produces: every N20 site next to an NGG/NAG PAM on both strands, sorted, duplicates kept).
Only data is stored."""
import os, pathlib, sys, multiprocessing, tempfile
import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, "/root/reference/src")
GOLD = ROOT / "tests" / "golden" / "extract"


def fasta(records, width):
    out = []
    for name, seq in records:
        out.append(">" + name + "\n")
        for i in range(0, len(seq), width):
            out.append(seq[i:i + width] + "\n")
    return "".join(out)


def random_seq(rng, n, p_n=0.0, lower=0.0):
    s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].copy()
    if p_n:
        s[rng.random(n) < p_n] = ord("N")
    if lower:
        m = rng.random(n) < lower
        s[m] = np.char.lower(s[m].view("S1")).view(np.uint8)
    return s.tobytes().decode()


def main():
    import crackling.utils.extractOfftargets as eo
    GOLD.mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(42)
    sets = {
        # one multi-FASTA file (exploded per record by the reference), wrapped lines, lower case, N runs
        "multi": fasta([("chr1 test", random_seq(rng, 6000, p_n=0.002, lower=0.3)),
                        ("chr2", random_seq(rng, 2500)),
                        ("tiny", "ACGTACGTACGTACGTACGTAGG"),            # exactly one forward site
                        ("short", "ACGT"),
                        ("polyG", "G" * 60),
                        ("rev", "CCAACGTACGTACGTACGTACGTT")], 60),
        # no line wrapping, repeats (duplicates in the output); two records because the reference's paginatedSort
        # fails (unbound `mergedFile`) when there is a single intermediate file
        "repeat": fasta([("r", ("ACGTTGCAAGGCTAGCTAGGATCCGGTTAACCGGA" * 40) + random_seq(rng, 500)),
                         ("r2", "TTCCGGAACC" * 30)], 100000),
    }
    pool = multiprocessing.Pool(2)
    for name, text in sets.items():
        fa = GOLD / f"{name}.fa"
        fa.write_text(text)
        out = GOLD / f"{name}.sites.txt"
        eo.startMultiprocessing([str(fa)], str(out), pool, 2, 100)
        os.chmod(out, 0o644)
        n = sum(1 for _ in open(out))
        print(name, "sites:", n)
    pool.close()


if __name__ == "__main__":
    main()
