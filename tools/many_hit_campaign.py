#!/usr/bin/env python3
"""Randomised campaign over batches of many-hit guides (GPU box): what tests/test_scale.py::
test_many_hit_guides_several_per_replay_workgroup does once with fixed sizes, here with random class sizes, neighbourhood
shapes, layouts, methods, thresholds and distances.  Every trial scores the batch whole, again, in pieces small enough
that no replay workgroup meets two guides, without hit slots, with whole buckets and as asynchronous batches on two
lanes: all must agree bit for bit; a sample is checked against the CPU oracle on the index of its brute-force
neighbourhoods.

    python tools/many_hit_campaign.py --trials 20 --seed 1"""
import argparse, pathlib, sys, tempfile, time
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
import crackling_amd as ca
import oracle_util as ou
from synth import text_order_key

ap = argparse.ArgumentParser()
ap.add_argument("--trials", type=int, default=10)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
LAYOUTS = [None, {"compact": 1}, {"compact": 1, "host_cold": 1}, {"keep_lists": 0}, {"sorted_layout": 0}, {"sorted_layout": 0, "inline_sigs": 0},
           {"sorted_layout": 0, "host_cold": 1}]
METHODS = ["and", "or", "avg", "mit", "cfd"]


def variants(c, count, k_from, first_pos):
    out = np.repeat(c, count)
    k = rng.integers(k_from, 5, size=len(out))
    for j in range(4):
        pos = rng.integers(first_pos, 20, size=len(out)).astype(np.uint64)
        sub = rng.integers(1, 4, size=len(out)).astype(np.uint64)
        out = np.where(j < k, out ^ (sub << (np.uint64(2) * pos)), out)
    return out


def neighbours(sigs, guide):
    x = sigs ^ np.uint64(guide)
    x |= x >> np.uint64(1)
    x &= np.uint64(0x5555555555555555)
    return np.flatnonzero(np.bitwise_count(x) <= 4)


torch.zeros(1, device="cuda:0")
for trial in range(a.trials):
    t0 = time.time()
    # classes: (guides, draws per centre, smallest substitution count, first position that may change)
    classes = [(int(rng.integers(100, 3000)), int(rng.integers(100, 1200)), 1, 0),
               (int(rng.integers(100, 2600)), int(rng.integers(1500, 5000)), int(rng.integers(1, 3)), int(rng.choice([0, 0, 4]))),
               (int(rng.integers(0, 700)), int(rng.integers(12000, 30000)), 3, int(rng.choice([0, 4, 8])))]
    n_centres = sum(c[0] for c in classes)
    centres = rng.integers(0, 1 << 40, size=n_centres, dtype=np.uint64)
    parts, at = [rng.integers(0, 1 << 40, size=int(rng.integers(1000, 300000)), dtype=np.uint64)], 0
    for n, draws, k_from, first_pos in classes:
        if n:
            parts.append(variants(centres[at:at + n], draws, k_from, first_pos))
        at += n
    sig = np.unique(np.concatenate(parts))
    sig = sig[np.argsort(text_order_key(sig), kind="stable")]
    occ = rng.integers(1, int(rng.choice([2, 4, 300])), size=len(sig)).astype(np.uint32)
    guides = centres[rng.permutation(n_centres)]
    if rng.integers(0, 2):  # guides next to the centres as well: other hit sets, other exits
        guides = np.concatenate([guides, guides[: n_centres // 4] ^ np.uint64(int(rng.integers(1, 4)) << int(2 * rng.integers(0, 20)))])
    layout = LAYOUTS[int(rng.integers(0, len(LAYOUTS)))]
    method = METHODS[int(rng.integers(0, 5))]
    thr = float(rng.choice([0.0, 50.0, 75.0, 90.0, 99.0]))
    dist = int(rng.choice([4, 4, 4, 3, 2]))
    what = dict(trial=trial, classes=classes, sites=len(sig), guides=len(guides), layout=layout, method=method, thr=thr, dist=dist)
    ix = ca.IsslIndex.build_on_device(sig, occ, device=0, options=layout)
    try:
        mit, cfd = ix.score(guides, dist, thr, method)
        hits = ix.stats()["hits"]

        def same(m, c, tag):
            bad = np.flatnonzero((m.view(np.uint64) != mit.view(np.uint64)) | (c.view(np.uint64) != cfd.view(np.uint64)))
            if len(bad):
                raise SystemExit(f"MISMATCH {tag}: {len(bad)} guides, first {bad[:8].tolist()}; {what}")

        same(*ix.score(guides, dist, thr, method), "again")
        pm = np.empty(len(guides)); pc = np.empty(len(guides))
        for lo in range(0, len(guides), 64):
            pm[lo:lo + 64], pc[lo:lo + 64] = ix.score(guides[lo:lo + 64], dist, thr, method)
        same(pm, pc, "pieces of 64")
        ix.set_option("hit_slots", 0)
        same(*ix.score(guides, dist, thr, method), "no hit slots")
        ix.set_option("hit_slots", 2)
        same(*ix.score(guides, dist, thr, method), "wide hit slots")
        ix.set_option("hit_slots", 1)
        if ix.get_option("is_sorted") == 1:
            ix.set_option("prune", 0)
            same(*ix.score(guides, dist, thr, method), "whole buckets")
            ix.set_option("prune", 1)
            same(*ix.score(guides, dist, thr, method), "pruned scan forced")
            ix.set_option("prune", -1)
        ix.set_option("lanes", 2)
        d_g = torch.from_numpy(guides.view(np.int64)).cuda()
        out_m = torch.zeros(4, len(guides), dtype=torch.float64, device="cuda:0"); out_c = torch.zeros_like(out_m)
        while True:
            for i in range(4):
                ix.score_device_async(d_g, out_m[i], out_c[i], dist, thr, method, stream=None)
            if ix.finish(None):
                break
        for i in range(4):
            same(out_m[i].cpu().numpy(), out_c[i].cpu().numpy(), f"two lanes, step {i}")
        ix.set_option("lanes", 1)
        # the oracle on the neighbourhoods of a sample
        pick = np.unique(rng.integers(0, len(guides), size=24))
        near = [neighbours(sig, g) for g in guides[pick]]
        keep = np.unique(np.concatenate(near + [np.arange(0, len(sig), 2000)]))
        with tempfile.TemporaryDirectory() as tmp:
            mini = ca.IsslIndex.build_from_sites(sig[keep], occ[keep])
            path = pathlib.Path(tmp) / "mini.issl"
            mini.write(path); mini.close()
            oracle = ou.OracleIndex(path)
            om, oc = oracle.score(guides[pick], dist, thr, method)
            oracle.close()
        if not (np.array_equal(mit[pick].view(np.uint64), om.view(np.uint64)) and np.array_equal(cfd[pick].view(np.uint64), oc.view(np.uint64))):
            raise SystemExit(f"MISMATCH against the oracle; {what}")
    finally:
        ix.close()
    print(f"trial {trial}: ok  {len(guides)} guides, {len(sig)} sites, {hits} hits, {method} thr {thr} dist {dist}, layout {layout}, {time.time() - t0:.1f} s", flush=True)
print(f"{a.trials} trials, no mismatch")
