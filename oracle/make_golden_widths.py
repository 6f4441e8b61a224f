#!/usr/bin/env python3
"""Golden vectors for slice widths 4 and 2 (tests/golden/width4, tests/golden/width2): the reference scorer is generic
in sliceWidth / sliceCount (isslScoreOfftargets.cpp:261-270,330-341) and isslCreateIndex writes such indexes
(isslCreateIndex.cpp:212-234; its local-MIT table then covers up to 9 / 19 mismatches: 431 909 / 1 048 574 masks).

Run in the build container only (needs /root/reference and `make -C oracle ref`):

    python oracle/make_golden_widths.py

The indexes are 7 and 17 MB (the score table), too large to commit: stored are the site list, the guides, the reference's
stdout for every method x threshold x max distance, its hit lists, and the SHA-256 of the reference-built index; the
tests rebuild the index with this repository's host builder and compare the digest first.  Finally the C restatement
(oracle/_build) is checked against every vector."""
import hashlib, json, os, pathlib, subprocess, sys
import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
REF = ROOT / "oracle" / "_ref"
ORA = ROOT / "oracle" / "_build"
GOLD = ROOT / "tests" / "golden"
sys.path.insert(0, str(ROOT / "oracle"))
from make_golden import make_clustered, METHODS  # noqa: E402  (the same neighbourhood generator as the clustered set)


def run(cmd, **kw):
    return subprocess.run(cmd, check=True, capture_output=True, **kw)


def main():
    subprocess.run(["make", "-C", str(ROOT / "oracle"), "all", "ref"], check=True, capture_output=True)
    bad = 0
    for width, seed in ((4, 41), (2, 21)):
        d = GOLD / f"width{width}"
        d.mkdir(parents=True, exist_ok=True)
        rng = np.random.default_rng(seed)
        sites, guides = make_clustered(rng, 25, 40)
        (d / "sites.txt").write_text("".join(s + "\n" for s in sites))
        (d / "guides.txt").write_text("".join(g + "\n" for g in guides))
        issl = pathlib.Path("/tmp") / f"golden_width{width}.issl"
        run([str(REF / "isslCreateIndex"), str(d / "sites.txt"), "20", str(width), str(issl)])
        (d / "index.sha256").write_text(hashlib.sha256(issl.read_bytes()).hexdigest() + "\n")
        env1 = dict(os.environ, OMP_NUM_THREADS="1")
        expected = {}
        thresholds, dists = [0, 75], [2, 4, 6]
        for m in METHODS:
            for t in thresholds:
                for k in dists:
                    expected[f"{m}|{t}|{k}"] = run([str(REF / "isslScoreOfftargets"), str(issl), str(d / "guides.txt"), str(k), str(t), m], env=env1).stdout.decode()
        (d / "expected.json").write_text(json.dumps(expected, indent=0, sort_keys=True))
        for t in thresholds:
            r = run([str(REF / "isslScoreOfftargets_hits"), str(issl), str(d / "guides.txt"), "4", str(t), "and"], env=env1)
            rows = [l.split("\t", 1)[1] for l in r.stderr.decode().splitlines() if l.startswith("HIT\t")]
            (d / f"hits_and_{t}.tsv").write_text("".join(x + "\n" for x in rows))
        # the C restatement: builder bytes, stdout, hit lists
        tmp = pathlib.Path("/tmp") / f"oracle_width{width}.issl"
        run([str(ORA / "oracle_create"), str(d / "sites.txt"), "20", str(width), str(tmp)])
        if tmp.read_bytes() != issl.read_bytes():
            bad += 1
            print("BUILDER MISMATCH", width)
        for key, want in expected.items():
            m, t, k = key.split("|")
            if run([str(ORA / "oracle_score"), str(issl), str(d / "guides.txt"), k, t, m], env=env1).stdout.decode() != want:
                bad += 1
                print("MISMATCH", width, key)
        for t in thresholds:
            hp = f"/tmp/oracle_hits_width{width}_{t}.tsv"
            run([str(ORA / "oracle_score"), str(issl), str(d / "guides.txt"), "4", str(t), "and"], env=dict(env1, ORACLE_DUMP_HITS=hp))
            if open(hp).read() != (d / f"hits_and_{t}.tsv").read_text():
                bad += 1
                print("HIT MISMATCH", width, t)
        n_hits = sum(1 for _ in open(d / "hits_and_0.tsv"))
        print(f"width{width}: {len(sites)} lines, index {issl.stat().st_size / 1e6:.1f} MB, {len(expected)} outputs, {n_hits} hits at threshold 0, oracle mismatches so far: {bad}")
        issl.unlink(); tmp.unlink()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
