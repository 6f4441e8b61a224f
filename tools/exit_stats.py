#!/usr/bin/env python3
"""How much of the tail's work lies behind the early exit?  (development aid)
Scores a synthetic batch, then reports hits found vs hits actually scored before the exit (isslScoreOfftargets.cpp:467-496)
and the distribution of hits per guide."""
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
a = ap.parse_args()
sigs, occ = (markov_sites_fast if a.dist == "markov" else random_sites_fast)(a.sites, seed=20261003)
ix = ca.IsslIndex.build_on_device(sigs, occ, device=0)
guides = random_guides_fast(sigs, a.guides, seed=777)
ix.score(guides, 4, a.thr, "and"); st = ix.stats()
print("hits found", st["hits"], "per guide", st["hits"] / a.guides, flush=True)
kept_total = 0; per_guide_kept = []; found = []
exit_slice = []; found_by_slice = []; kept_by_slice = []
step = 4096
for lo in range(0, a.guides, step):
    g = guides[lo:lo + step]
    h = ix.dump_hits(g, 4, a.thr, "and")
    kept = np.bincount(h[:, 0], minlength=len(g))
    per_guide_kept.append(kept)
    ix.score(g, 4, 0.0, "and")  # threshold 0: no exit -> all hits counted
    h0 = ix.dump_hits(g, 4, 0.0, "and")
    found.append(np.bincount(h0[:, 0], minlength=len(g)))
    # per slice: hits found / scored, and the slice the last scored hit of every guide lies in
    fs = np.zeros((len(g), 5), np.int64); np.add.at(fs, (h0[:, 0], np.minimum(h0[:, 1], 4)), 1); found_by_slice.append(fs)
    ks = np.zeros((len(g), 5), np.int64); np.add.at(ks, (h[:, 0], np.minimum(h[:, 1], 4)), 1); kept_by_slice.append(ks)
    last = np.full(len(g), -1, np.int64); np.maximum.at(last, h[:, 0], h[:, 1].astype(np.int64)); exit_slice.append(last)
    if lo // step % 5 == 0:
        print(lo, "kept so far", int(np.concatenate(per_guide_kept).sum()), "found", int(np.concatenate(found).sum()), flush=True)
kept = np.concatenate(per_guide_kept); found = np.concatenate(found)
exited = kept < found
print(f"guides {a.guides}: hits found {found.sum()}, scored before the exit {kept.sum()} ({kept.sum() / found.sum():.3%}); "
      f"guides that exit early {exited.sum()} ({exited.mean():.1%}), their hits found {found[exited].sum()} "
      f"({found[exited].sum() / found.sum():.1%} of all), scored {kept[exited].sum()}")
for q in (50, 90, 99, 99.9):
    print(f"  hits per guide p{q}: {np.percentile(found, q):.0f}")
big = found > 512
print(f"guides with > 512 hits: {big.sum()} holding {found[big].sum() / found.sum():.1%} of the hits; of them exit early: {(exited & big).sum()}")
# slice of the exit: first matching slice of the last scored hit
fs = np.concatenate(found_by_slice); ks = np.concatenate(kept_by_slice); last = np.concatenate(exit_slice)
print("hits found by slice", fs.sum(0).tolist(), "scored by slice", ks.sum(0).tolist())
for s_ in range(5):
    ex = exited & (last == s_)
    # a split after slice s_: phase 1 does slices 0..s_ for everybody, phase 2 the rest for the guides still running
    p1 = fs[:, :s_ + 1].sum(); still = ~(exited & (last <= s_)); p2 = fs[still][:, s_ + 1:].sum()
    print(f"  exit in slice {s_}: {ex.sum()} guides; split after slice {s_}: phase 1 finds {p1} hits, phase 2 {p2} "
          f"({(p1 + p2) / fs.sum():.1%} of all), guides in phase 2: {still.sum()} ({still.mean():.1%})")
