#!/usr/bin/env python3
"""Throughput of back-to-back asynchronous batches under different lane / scan-grid settings (development aid).

    python tools/pipeline_ab.py --sites 300000000 --guides 100000 --variants "lanes=1/lanes=2/lanes=2,scan_blocks=4096"

Every variant scores the same batch `--steps` times back to back (alternating output buffers), reports ms per step and
the scan launch's own span, and checks the scores of the last two steps bit-for-bit against the first variant's."""
import argparse, pathlib, sys, time
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
import crackling_amd as ca
from synth import random_sites_fast, markov_sites_fast, random_guides_fast

ap = argparse.ArgumentParser()
ap.add_argument("--sites", type=int, default=300_000_000)
ap.add_argument("--guides", type=int, default=100_000)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--thr", type=float, default=75.0)
ap.add_argument("--dist", default="uniform", choices=["uniform", "markov"])
ap.add_argument("--rounds", type=int, default=2, help="the variant list is measured this many times over, in turn")
ap.add_argument("--variants", default="lanes=1/lanes=2")
a = ap.parse_args()

torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()
sigs, occ = (markov_sites_fast if a.dist == "markov" else random_sites_fast)(a.sites, seed=20261003)
ix = ca.IsslIndex.build_on_device(sigs, occ, device=0)
guides = random_guides_fast(sigs, a.guides, seed=777)
d_g = torch.from_numpy(guides.view(np.int64)).cuda()
out_m = torch.empty(2, a.guides, dtype=torch.float64, device="cuda:0"); out_c = torch.empty_like(out_m)
print(f"index {len(sigs)} sites, image {ix.device_bytes() / 1e9:.1f} GB", flush=True)


def run(steps):
    for i in range(steps):
        ix.score_device_async(d_g, out_m[i & 1], out_c[i & 1], 4, a.thr, "and", stream=None)
    return ix.finish(None)


ref = None
for rnd in range(a.rounds):
    for variant in a.variants.split("/"):
        ix.set_option("lanes", 1).set_option("scan_blocks", 1024)
        for kv in filter(None, variant.split(",")):
            ix.set_option(*kv.split("="))
        while not run(2):
            pass
        while not run(4):
            pass
        torch.cuda.synchronize()
        t = time.perf_counter()
        ok = run(a.steps)
        dt = time.perf_counter() - t
        st = ix.stats()
        got = (out_m.cpu().numpy().copy(), out_c.cpu().numpy().copy())
        if ref is None:
            ref = got
        same = all(np.array_equal(x.view(np.uint64), y.view(np.uint64)) for x, y in zip(got, ref))
        print(f"round {rnd} {variant:40s} {dt * 1e3 / a.steps:7.3f} ms/step  {a.guides * a.steps / dt / 1e6:6.2f} M guides/s  "
              f"scan span {st['ms_scan']:.3f} ms (events {st['ms_scan_events']:.3f})  finished={ok} identical={same}", flush=True)
ix.close()
