#!/usr/bin/env python3
"""Golden vectors for a local-MIT table that is NOT what isslCreateIndex writes (tests/golden/oddtable/).

Run in the build container only (needs /root/reference and `make -C oracle ref`):

    python oracle/make_golden_oddtable.py

The reference loads the {mask, score} pairs of the .issl into a hash map with insert() -- the FIRST pair of a mask
wins (isslScoreOfftargets.cpp:188-197) -- and reads it with operator[], so a mask that is absent contributes 0.0
(:394).  Every reference-built table holds each mask once, on even bits only; the GPU path then uses a dense
2^20-entry table.  This set exercises the other branch (sorted unique table + binary search): the clustered golden
index with its score table rewritten in place --
  * every 7th pair takes the MASK of the pair 3 places before it (a duplicate with a different score: the first
    wins; the overwritten mask is now missing -> 0.0),
  * every 11th pair gets bit 1 set in its mask (an odd bit: the table is no longer "dense-shaped"; that mask can
    never equal a mismatch pattern, so its original pattern is missing too).
Outputs: index.issl, guides.txt (the clustered guides), expected.json {"<method>|<thr>|<maxDist>": reference stdout},
hits_and_<thr>.tsv from the reference scorer with the one extra fprintf (1 thread: the reference's operator[] inserts
into the shared map on a miss, which is only safe single-threaded).  Data only; finally the C restatement is checked
against every vector."""
import json, os, pathlib, struct, subprocess, sys, hashlib

ROOT = pathlib.Path(__file__).resolve().parent.parent
REF = ROOT / "oracle" / "_ref"
ORA = ROOT / "oracle" / "_build"
GOLD = ROOT / "tests" / "golden"


def run(cmd, **kw):
    return subprocess.run(cmd, check=True, capture_output=True, **kw)


def main():
    subprocess.run(["make", "-C", str(ROOT / "oracle"), "all", "ref"], check=True, capture_output=True)
    src = GOLD / "clustered"
    d = GOLD / "oddtable"
    d.mkdir(parents=True, exist_ok=True)
    data = bytearray((src / "index.issl").read_bytes())
    n_scores = struct.unpack_from("<6Q", data, 0)[5]
    pairs = [list(struct.unpack_from("<Qd", data, 48 + 16 * i)) for i in range(n_scores)]
    changed = 0
    for i in range(n_scores):
        if i % 7 == 5 and i >= 3:
            pairs[i][0] = pairs[i - 3][0]
            changed += 1
        elif i % 11 == 4:
            pairs[i][0] |= 2
            changed += 1
    for i, (m, s) in enumerate(pairs):
        struct.pack_into("<Qd", data, 48 + 16 * i, m, s)
    (d / "index.issl").write_bytes(bytes(data))
    (d / "guides.txt").write_text((src / "guides.txt").read_text())
    env1 = dict(os.environ, OMP_NUM_THREADS="1")
    expected = {}
    thresholds, dists = [0, 75], [2, 4]
    for m in ("and", "or", "avg", "mit", "cfd"):
        for t in thresholds:
            for k in dists:
                expected[f"{m}|{t}|{k}"] = run([str(REF / "isslScoreOfftargets"), str(d / "index.issl"), str(d / "guides.txt"),
                                                str(k), str(t), m], env=env1).stdout.decode()
    (d / "expected.json").write_text(json.dumps(expected, indent=0, sort_keys=True))
    for t in thresholds:
        r = run([str(REF / "isslScoreOfftargets_hits"), str(d / "index.issl"), str(d / "guides.txt"), "4", str(t), "and"], env=env1)
        rows = [l.split("\t", 1)[1] for l in r.stderr.decode().splitlines() if l.startswith("HIT\t")]
        (d / f"hits_and_{t}.tsv").write_text("".join(x + "\n" for x in rows))
    (d / "index.sha256").write_text(hashlib.sha256((d / "index.issl").read_bytes()).hexdigest() + "\n")
    # the table must matter: MIT columns differ from the clustered set's
    base = json.loads((src / "expected.json").read_text())
    differs = sum(expected[k] != base[k] for k in expected if k in base)
    bad = 0
    for key, want in expected.items():
        m, t, k = key.split("|")
        got = run([str(ORA / "oracle_score"), str(d / "index.issl"), str(d / "guides.txt"), k, t, m], env=env1).stdout.decode()
        if got != want:
            bad += 1
            print("MISMATCH", key)
    print(f"oddtable: {changed} of {n_scores} table pairs rewritten, {len(expected)} outputs ({differs} differ from the "
          f"clustered set's), oracle mismatches: {bad}")
    sys.exit(1 if bad or not differs else 0)


if __name__ == "__main__":
    main()
