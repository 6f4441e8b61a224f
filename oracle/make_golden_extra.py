#!/usr/bin/env python3
"""Golden vectors for branches the builder-made indexes never reach (tests/golden/{bigocc,mixedocc,signedtable}/).

Run in the build container only (needs /root/reference and `make -C oracle ref`):

    python oracle/make_golden_extra.py

All outputs are data produced by the COMPILED REFERENCE (oracle/_ref); nothing of its source text is stored.

bigocc       the clustered golden index with the occurrence half of its list entries rewritten IN PLACE (the scorer
             reads it at isslScoreOfftargets.cpp:348 as a uint32 and multiplies the hit's terms by it, :394 / :460):
             the sites the clustered guides actually hit get counts 254, 255, 256, 70 000, 2^24 - 2, 2^24 - 1, 2^24,
             2^24 + 5 and 2^32 - 1 -- the saturation points of the 8-bit (`occ8`, host-cold layout) and 24-bit
             (`srec`, sorted layout) copies the GPU image keeps -- the same count in all five lists of a site.
mixedocc     the same sites, but every one of a site's five list entries carries a DIFFERENT count (base + slice): the
             reference takes the count of the entry it meets first (the first exactly matching slice), so an image
             that keeps one count per site would be wrong; no builder writes such an index.
signedtable  a dense neighbourhood (every site within two substitutions of one centre, plus noise) so that guides
             have 600 ... 1800 hits (the one-workgroup-per-guide replay) next to guides with a handful (the
             one-wave replay), and a local-MIT table whose values were rewritten: every 5th negative, two NaN
             (quiet, positive), several +inf.  The reference adds whatever the table holds (:394): totals can fall,
             stick at NaN (every exit test false from there on) or at +inf.  (-inf is left out on purpose: inf - inf
             makes the x86 "default NaN", which has the sign bit set and prints as "-nan"; what a GPU writes there is
             not the reference's business.)

For every set: index.issl, guides.txt, expected.json {"<method>|<thr>|<maxDist>": reference stdout},
hits_and_<thr>.tsv from the reference scorer with the one extra fprintf (1 thread), index.sha256; signedtable also
sites.txt (input of the reference isslCreateIndex).  Finally the C restatement (oracle/_build) is checked against
every vector."""
import hashlib, json, os, pathlib, struct, subprocess, sys
import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
REF = ROOT / "oracle" / "_ref"
ORA = ROOT / "oracle" / "_build"
GOLD = ROOT / "tests" / "golden"
METHODS = ["and", "or", "avg", "mit", "cfd"]
BIG_COUNTS = [254, 255, 256, 70000, (1 << 24) - 2, (1 << 24) - 1, 1 << 24, (1 << 24) + 5, (1 << 32) - 1]


def run(cmd, **kw):
    return subprocess.run(cmd, check=True, capture_output=True, **kw)


def sections(data):
    """Offsets of the .issl sections (isslCreateIndex.cpp:256-289)."""
    n_sites, seq_len, n_lines, width, n_slices, n_scores = struct.unpack_from("<6Q", data, 0)
    off_scores = 48
    off_sites = off_scores + 16 * n_scores
    off_sizes = off_sites + 8 * n_sites
    off_entries = off_sizes + 8 * n_slices * (1 << width)
    return dict(n_sites=n_sites, n_slices=n_slices, n_scores=n_scores, width=width, off_scores=off_scores,
                off_sites=off_sites, off_sizes=off_sizes, off_entries=off_entries)


def rewrite_occ(data, site_to_count, per_slice_step):
    """Entries are occ << 32 | id; slice s owns entries [s * n_sites, (s + 1) * n_sites)."""
    sec = sections(data)
    n = sec["n_sites"]
    changed = 0
    for s in range(sec["n_slices"]):
        for k in range(n):
            off = sec["off_entries"] + 8 * (s * n + k)
            (e,) = struct.unpack_from("<Q", data, off)
            sid = e & 0xFFFFFFFF
            if sid in site_to_count:
                occ = (site_to_count[sid] + per_slice_step * s) & 0xFFFFFFFF
                struct.pack_into("<Q", data, off, (occ << 32) | sid)
                changed += 1
    return changed


def reference_outputs(d, thresholds, dists, hit_thresholds):
    env1 = dict(os.environ, OMP_NUM_THREADS="1")
    expected = {}
    for m in METHODS:
        for t in thresholds:
            for k in dists:
                expected[f"{m}|{t}|{k}"] = run([str(REF / "isslScoreOfftargets"), str(d / "index.issl"), str(d / "guides.txt"),
                                                str(k), str(t), m], env=env1).stdout.decode()
    (d / "expected.json").write_text(json.dumps(expected, indent=0, sort_keys=True))
    for t in hit_thresholds:
        r = run([str(REF / "isslScoreOfftargets_hits"), str(d / "index.issl"), str(d / "guides.txt"), "4", str(t), "and"], env=env1)
        rows = [l.split("\t", 1)[1] for l in r.stderr.decode().splitlines() if l.startswith("HIT\t")]
        (d / f"hits_and_{t}.tsv").write_text("".join(x + "\n" for x in rows))
    (d / "index.sha256").write_text(hashlib.sha256((d / "index.issl").read_bytes()).hexdigest() + "\n")
    return expected


def check_oracle(d, expected, hit_thresholds):
    env1 = dict(os.environ, OMP_NUM_THREADS="1")
    bad = 0
    for key, want in expected.items():
        m, t, k = key.split("|")
        got = run([str(ORA / "oracle_score"), str(d / "index.issl"), str(d / "guides.txt"), k, t, m], env=env1).stdout.decode()
        if got != want:
            bad += 1
            print("MISMATCH", d.name, key)
    for t in hit_thresholds:
        hp = f"/tmp/oracle_hits_{d.name}_{t}.tsv"
        run([str(ORA / "oracle_score"), str(d / "index.issl"), str(d / "guides.txt"), "4", str(t), "and"],
            env=dict(env1, ORACLE_DUMP_HITS=hp))
        if open(hp).read() != (d / f"hits_and_{t}.tsv").read_text():
            bad += 1
            print("HIT MISMATCH", d.name, t)
    return bad


def occ_set(name, per_slice_step):
    src = GOLD / "clustered"
    d = GOLD / name
    d.mkdir(parents=True, exist_ok=True)
    data = bytearray((src / "index.issl").read_bytes())
    hit_ids = sorted({int(l.split("\t")[3]) for l in (src / "hits_and_0.tsv").read_text().splitlines()})
    # every third site the clustered guides hit takes one of the large counts, in turn
    site_to_count = {sid: BIG_COUNTS[(i // 3) % len(BIG_COUNTS)] for i, sid in enumerate(hit_ids) if i % 3 == 0}
    changed = rewrite_occ(data, site_to_count, per_slice_step)
    (d / "index.issl").write_bytes(bytes(data))
    (d / "guides.txt").write_text((src / "guides.txt").read_text())
    thr, dists, hthr = [0, 50, 75], [2, 4], [0, 50, 75]
    expected = reference_outputs(d, thr, dists, hthr)
    base = json.loads((src / "expected.json").read_text())
    differs = sum(expected[k] != base[k] for k in expected if k in base)
    bad = check_oracle(d, expected, hthr)
    print(f"{name}: {changed} list entries of {len(site_to_count)} sites rewritten, {len(expected)} outputs "
          f"({differs} differ from the clustered set's), oracle mismatches: {bad}")
    return bad + (0 if differs else 1)


def signed_table_set():
    name = "signedtable"
    d = GOLD / name
    d.mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(20261004)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    centre = rng.integers(0, 4, size=20, dtype=np.uint8)
    rows = {centre.tobytes()}
    for p in range(20):                      # every site within two substitutions of the centre: 1 + 60 + 1710
        for a in range(1, 4):
            v = centre.copy(); v[p] = (v[p] + a) % 4
            rows.add(v.tobytes())
            for q in range(p + 1, 20):
                for b in range(1, 4):
                    w = v.copy(); w[q] = (w[q] + b) % 4
                    rows.add(w.tobytes())
    lines = []
    for r in rows:
        lines += [bases[np.frombuffer(r, dtype=np.uint8)].tobytes().decode()] * int(rng.integers(1, 4))
    lines += [bases[r].tobytes().decode() for r in rng.integers(0, 4, size=(1500, 20), dtype=np.uint8)]
    lines.sort()
    (d / "sites.txt").write_text("".join(s + "\n" for s in lines))

    def mutated(k):
        v = centre.copy()
        for p in rng.choice(20, size=k, replace=False):
            v[p] = (v[p] + rng.integers(1, 4)) % 4
        return bases[v].tobytes().decode()
    guides = [mutated(0)] + [mutated(1) for _ in range(4)] + [mutated(2) for _ in range(6)] + \
             [mutated(3) for _ in range(8)] + [mutated(4) for _ in range(6)] + [mutated(5) for _ in range(3)] + \
             [bases[r].tobytes().decode() for r in rng.integers(0, 4, size=(4, 20), dtype=np.uint8)]
    (d / "guides.txt").write_text("".join(g + "\n" for g in guides))
    run([str(REF / "isslCreateIndex"), str(d / "sites.txt"), "20", "8", str(d / "index.issl")])
    data = bytearray((d / "index.issl").read_bytes())
    sec = sections(data)
    nan_bits, inf_bits = 0x7FF8000000000000, 0x7FF0000000000000
    changed = {"negative": 0, "nan": 0, "inf": 0}
    for i in range(sec["n_scores"]):
        off = sec["off_scores"] + 16 * i + 8
        (v,) = struct.unpack_from("<d", data, off)
        if i in (1777, 4001):
            struct.pack_into("<Q", data, off, nan_bits); changed["nan"] += 1
        elif i % 997 == 500:
            struct.pack_into("<Q", data, off, inf_bits); changed["inf"] += 1
        elif i % 5 == 2:
            struct.pack_into("<d", data, off, -v); changed["negative"] += 1
    (d / "index.issl").write_bytes(bytes(data))
    thr, dists, hthr = [0, 50, 75], [2, 4], [0, 50, 75]
    expected = reference_outputs(d, thr, dists, hthr)
    nan_lines = sum(v.count("nan") for v in expected.values())
    inf_lines = sum(v.count("\t0.000000") for v in expected.values())
    n_hits = [sum(1 for l in (d / "hits_and_0.tsv").read_text().splitlines() if l.split("\t")[0] == str(g)) for g in range(len(guides))]
    bad = check_oracle(d, expected, hthr)
    print(f"{name}: {len(lines)} lines, table rewritten {changed}, hits per guide min/median/max "
          f"{min(n_hits)}/{sorted(n_hits)[len(n_hits) // 2]}/{max(n_hits)}, guides with > 512 hits: {sum(h > 512 for h in n_hits)}, "
          f"'nan' fields {nan_lines}, zero scores (inf totals) {inf_lines}, {len(expected)} outputs, oracle mismatches: {bad}")
    return bad + (0 if nan_lines and inf_lines and max(n_hits) > 512 else 1)


def main():
    subprocess.run(["make", "-C", str(ROOT / "oracle"), "all", "ref"], check=True, capture_output=True)
    bad = occ_set("bigocc", 0) + occ_set("mixedocc", 1) + signed_table_set()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
