#!/usr/bin/env python3
"""Repeatability of a large batch on the skewed synthetic index (development aid): the same batch scored several times,
whole and in pieces; guides whose scores differ between runs are listed with their hit counts."""
import argparse, pathlib, sys, time
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import crackling_amd as ca
from synth import random_sites_fast, markov_sites_fast, random_guides_fast
ap = argparse.ArgumentParser()
ap.add_argument("--sites", type=int, default=300_000_000)
ap.add_argument("--guides", type=int, default=100_000)
ap.add_argument("--dist", default="markov")
ap.add_argument("--thr", type=float, default=75.0)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--piece", type=int, default=5000)
a = ap.parse_args()
sigs, occ = (markov_sites_fast if a.dist == "markov" else random_sites_fast)(a.sites, seed=20261003)
ix = ca.IsslIndex.build_on_device(sigs, occ, device=0)
guides = random_guides_fast(sigs, a.guides, seed=777)
runs = []
for r in range(a.reps):
    t = time.perf_counter()
    m, c = ix.score(guides, 4, a.thr, "and")
    dt = time.perf_counter() - t
    st = ix.stats()
    runs.append((m.copy(), c.copy()))
    print(f"rep {r}: {dt * 1e3:.1f} ms, hits {st['hits']}, kernels {st['ms_total']:.2f} ms "
          f"(bin {st['ms_bin']:.2f} scan {st['ms_scan']:.2f} verify {st['ms_verify']:.2f} group {st['ms_group']:.2f} replay {st['ms_replay']:.2f})", flush=True)
pm = np.empty(a.guides); pc = np.empty(a.guides)
for lo in range(0, a.guides, a.piece):
    pm[lo:lo + a.piece], pc[lo:lo + a.piece] = ix.score(guides[lo:lo + a.piece], 4, a.thr, "and")
print("pieces done", flush=True)
for r, (m, c) in enumerate(runs):
    bad = np.flatnonzero((m.view(np.uint64) != pm.view(np.uint64)) | (c.view(np.uint64) != pc.view(np.uint64)))
    print(f"rep {r}: {len(bad)} guides differ from the piecewise scores", bad[:20].tolist(), flush=True)
    if len(bad):
        sub = guides[bad[:64]]
        h0 = ix.dump_hits(sub, 4, 0.0, "and")
        cnt = np.bincount(h0[:, 0], minlength=len(sub))
        per_slice = np.zeros((len(sub), 5), np.int64); np.add.at(per_slice, (h0[:, 0], np.minimum(h0[:, 1], 4)), 1)
        for k in range(min(len(sub), 12)):
            print(f"   guide {bad[k]}: hits {cnt[k]} by slice {per_slice[k].tolist()}  whole ({m[bad[k]]!r}, {c[bad[k]]!r}) pieces ({pm[bad[k]]!r}, {pc[bad[k]]!r})")
