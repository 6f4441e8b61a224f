#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the COMPILED REFERENCE binaries.

Run in the build container only (needs /root/reference and `make -C oracle ref`):

    python oracle/make_golden.py

For every fixture set it writes
  sites.txt      sorted 20-mer list (input of isslCreateIndex)
  index.issl     bytes written by the reference isslCreateIndex (slice width 8)
  guides.txt     query file (20 chars + LF per guide)
  expected.json  {"<method>|<thr>|<maxDist>": stdout of the reference isslScoreOfftargets}
  hits_<thr>.tsv hit lists "guide slice pos id dist occ" from the reference scorer compiled with one
                 extra fprintf (oracle/Makefile target _ref/isslScoreOfftargets_hits), 1 thread
Only data is written; no reference source text is stored.  The script finally checks that the C
restatement (oracle/_build) reproduces every vector byte-for-byte.
"""
import json, os, subprocess, sys, hashlib, pathlib
import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
REF = ROOT / "oracle" / "_ref"
ORA = ROOT / "oracle" / "_build"
GOLD = ROOT / "tests" / "golden"
BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
METHODS = ["and", "or", "avg", "mit", "cfd"]


def to_text(arr):
    """arr: (n,20) uint8 in 0..3 -> list of 20-char strings"""
    return [BASES[r].tobytes().decode() for r in arr]


def mutate(rng, row, k):
    row = row.copy()
    pos = rng.choice(20, size=k, replace=False)
    for p in pos:
        row[p] = (row[p] + rng.integers(1, 4)) % 4
    return row


def make_uniform(rng, n_sites, n_guides):
    sites = rng.integers(0, 4, size=(n_sites, 20), dtype=np.uint8)
    dup = sites[rng.integers(0, n_sites, size=n_sites // 40)]  # ~2.5 % duplicate lines
    allsites = np.concatenate([sites, dup])
    guides = []
    for g in range(n_guides):
        if g % 5 == 4:
            guides.append(rng.integers(0, 4, size=20, dtype=np.uint8))
        else:
            guides.append(mutate(rng, sites[rng.integers(0, n_sites)], int(rng.integers(0, 5))))
    return sorted(to_text(allsites)), to_text(np.array(guides))


def make_clustered(rng, n_centres, n_guides):
    """Neighbourhoods of <=3 mismatches with 1..6 occurrences so that early exit fires at thr=75."""
    centres = rng.integers(0, 4, size=(n_centres, 20), dtype=np.uint8)
    rows = []
    for c in centres:
        for _ in range(int(rng.integers(6, 60))):
            v = mutate(rng, c, int(rng.integers(0, 4)))
            rows += [v] * int(rng.integers(1, 7))
    rows += list(rng.integers(0, 4, size=(3000, 20), dtype=np.uint8))
    guides = [mutate(rng, centres[rng.integers(0, n_centres)], int(rng.integers(0, 3))) for _ in range(n_guides)]
    return sorted(to_text(np.array(rows))), to_text(np.array(guides))


def make_edge(rng):
    sites = set(to_text(rng.integers(0, 4, size=(600, 20), dtype=np.uint8)))
    # no site may start with "TTTT": slice 0, key 0xFF stays empty
    sites = {s for s in sites if not s.startswith("TTTT")}
    # unit vector of the worked example in isslScoreOfftargets.cpp:350-375 (8-mer embedded, 4 mismatches)
    sites.add("ATATCGAT" + "ACGTACGTACGT")
    sites.add("A" * 20)
    sites.add("T" * 16 + "ACGT")
    lst = sorted(sites)
    lst += ["CCCCCCCCCCGGGGGGGGGG"] * 5  # occurrences > 1, run at the very end of the (sorted) file
    lst = sorted(lst)
    guides = [
        "AATTGCAT" + "ACGTACGTACGT",  # 4 mismatches vs the embedded example
        "A" * 20,                      # exact match, dist 0
        "TTTT" + "ACGTACGTACGTACGT",   # empty bucket in slice 0
        "ACGTNCGTACGTACGTACGT",        # non-ACGT byte encodes as 'A'
        "acgtacgtacgtacgtacgt",        # lower case -> all 'A'
        "CCCCCCCCCCGGGGGGGGGG",        # site with 5 occurrences
        "CCCCCCCCCCGGGGGGGGGA",        # 1 mismatch vs it
        "GATTACAGATTACAGATTAC",        # (almost surely) absent from the index
        "T" * 16 + "ACGT",
        "T" * 20,
    ]
    return lst, guides


def run(cmd, **kw):
    return subprocess.run(cmd, check=True, capture_output=True, **kw)


def build_set(name, sites, guides, thresholds, dists):
    d = GOLD / name
    d.mkdir(parents=True, exist_ok=True)
    (d / "sites.txt").write_text("".join(s + "\n" for s in sites))
    (d / "guides.txt").write_text("".join(g + "\n" for g in guides))
    run([str(REF / "isslCreateIndex"), str(d / "sites.txt"), "20", "8", str(d / "index.issl")])
    env1 = dict(os.environ, OMP_NUM_THREADS="1")
    expected = {}
    for m in METHODS + ["xyz"]:
        for t in thresholds:
            for k in dists:
                out = run([str(REF / "isslScoreOfftargets"), str(d / "index.issl"), str(d / "guides.txt"),
                           str(k), str(t), m]).stdout.decode()
                expected[f"{m}|{t}|{k}"] = out
    (d / "expected.json").write_text(json.dumps(expected, indent=0, sort_keys=True))
    for t in thresholds:
        r = run([str(REF / "isslScoreOfftargets_hits"), str(d / "index.issl"), str(d / "guides.txt"), "4", str(t), "and"], env=env1)
        rows = [l.split("\t", 1)[1] for l in r.stderr.decode().splitlines() if l.startswith("HIT\t")]
        (d / f"hits_and_{t}.tsv").write_text("".join(x + "\n" for x in rows))
    sha = hashlib.sha256((d / "index.issl").read_bytes()).hexdigest()
    (d / "index.sha256").write_text(sha + "\n")
    return expected


def check_oracle(name, thresholds, dists):
    """The C restatement must reproduce the reference byte-for-byte."""
    d = GOLD / name
    expected = json.loads((d / "expected.json").read_text())
    tmp = pathlib.Path("/tmp") / f"oracle_{name}.issl"
    run([str(ORA / "oracle_create"), str(d / "sites.txt"), "20", "8", str(tmp)])
    assert tmp.read_bytes() == (d / "index.issl").read_bytes(), f"{name}: builder bytes differ"
    bad = 0
    for key, want in expected.items():
        m, t, k = key.split("|")
        got = run([str(ORA / "oracle_score"), str(d / "index.issl"), str(d / "guides.txt"), k, t, m]).stdout.decode()
        if got != want:
            bad += 1
            print(f"MISMATCH {name} {key}")
    for t in thresholds:
        hp = f"/tmp/oracle_hits_{name}_{t}.tsv"
        run([str(ORA / "oracle_score"), str(d / "index.issl"), str(d / "guides.txt"), "4", str(t), "and"],
            env=dict(os.environ, ORACLE_DUMP_HITS=hp))
        if open(hp).read() != (d / f"hits_and_{t}.tsv").read_text():
            bad += 1
            print(f"HIT MISMATCH {name} thr={t}")
    print(f"{name}: {len(expected)} outputs, {bad} mismatches")
    return bad


def main():
    subprocess.run(["make", "-C", str(ROOT / "oracle"), "all", "ref"], check=True, capture_output=True)
    rng = np.random.default_rng(20261003)
    sets = {
        "uniform": (make_uniform(rng, 8000, 200), [0, 75], [0, 1, 2, 3, 4]),
        "clustered": (make_clustered(rng, 40, 60), [0, 50, 75, 90], [0, 2, 4]),
        "edge": (make_edge(rng), [0, 75], [0, 1, 4, 6]),
    }
    bad = 0
    for name, ((sites, guides), thr, dists) in sets.items():
        build_set(name, sites, guides, thr, dists)
        bad += check_oracle(name, thr, dists)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
