#!/usr/bin/env python3
"""Ad-hoc timing of the scoring pipeline on synthetic data (development aid, not the bench)."""
import argparse, sys, time, pathlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import crackling_amd as ca
from synth import random_sites, random_sites_fast, markov_sites, markov_sites_fast, random_guides

ap = argparse.ArgumentParser()
ap.add_argument("--sites", type=int, default=5_000_000)
ap.add_argument("--guides", type=int, default=10_000)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--thr", type=float, default=0.0)
ap.add_argument("--variants", default="", help="option settings to compare, separated by / (each a comma list of key=value, "
                "see issl_index_set_option), e.g. scan_blocks=1024/scan_blocks=2048,item_guides=256")
ap.add_argument("--fast-synth", action="store_true", help="random_sites_fast + index built on the device (large indexes)")
ap.add_argument("--dist", default="uniform", choices=["uniform", "markov"])
ap.add_argument("--subsets", default="", help="comma list of batch sizes: score the first k guides for every k (default: all)")
ap.add_argument("--max-dist", type=int, default=4)
ap.add_argument("--slice-width", type=int, default=8, help="4: ten 4-bit slices (host-built index)")
ap.add_argument("--json", default=None, help="write the best repetition (by scan time) of the last variant here")
ap.add_argument("--write-issl", default=None)
ap.add_argument("--write-guides", default=None)
a = ap.parse_args()
gen = (markov_sites_fast if a.fast_synth else markov_sites) if a.dist == "markov" else (random_sites_fast if a.fast_synth else random_sites)
t = time.time(); sigs, occ = gen(a.sites, seed=1); guides = random_guides(sigs, a.guides, seed=2)
print(f"synth {time.time()-t:.1f}s  distinct={len(sigs)}", flush=True)
t = time.time()
on_device = (a.fast_synth or a.dist == "markov") and a.slice_width == 8
ix = ca.IsslIndex.build_on_device(sigs, occ, device=0) if on_device else ca.IsslIndex.build_from_sites(sigs, occ, slice_width=a.slice_width)
t_build = time.time() - t; print(f"build {t_build:.1f}s", flush=True)
if a.write_issl:
    t = time.time(); ix.write(a.write_issl); print(f"write issl {time.time()-t:.1f}s", flush=True)
if a.write_guides:
    open(a.write_guides, "w").write("".join(s + "\n" for s in ca.decode_guides(guides)))
t = time.time()
if not on_device:
    ix.upload(0)
t_upload = time.time() - t
print(f"upload {t_upload:.1f}s  image={ix.device_bytes()/1e9:.2f} GB", flush=True)
import os, json
best = None
for variant in a.variants.split("/"):
  for kv in filter(None, variant.split(",")):
    ix.set_option(*kv.split("="))
  for k in [int(x) for x in a.subsets.split(",") if x] or [len(guides)]:
   print(f"-- {variant or 'defaults'} | {k} guides", flush=True)
   for r in range(a.reps):
    t = time.time(); mit, cfd = ix.score(guides[:k], a.max_dist, a.thr, "and"); dt = time.time() - t
    st = ix.stats()
    algo = 8.0 * st["candidates"]
    print(f"rep{r}: pruned={st['pruned']} wall {dt*1e3:.2f} ms | bin {st['ms_bin']:.3f} scan {st['ms_scan']:.3f} verify {st['ms_verify']:.3f} group {st['ms_group']:.3f} "
          f"replay {st['ms_replay']:.3f} ms | cand {st['candidates']:.3e} hits {st['hits']} tiles {st['scan_tiles']} | "
          f"scan: {st['candidates']/st['ms_scan']/1e9:.2f} Tcmp/s, algorithmic {algo/st['ms_scan']/1e9:.1f} TB/s | "
          f"{k/st['ms_total']*1e3:.0f} guides/s (kernels)", flush=True)
    if r > 0 and (best is None or st["ms_scan"] < best["scan_ms"]):
        best = {"what": f"tools/quick_perf.py --sites {a.sites} --guides {a.guides} --thr {a.thr} on one MI355X (best of {a.reps - 1} warm repetitions)",
                "distinct_sites": int(len(sigs)), "image_GB": ix.device_bytes() / 1e9, "host_build_s": t_build, "upload_s": t_upload,
                "wall_ms": dt * 1e3, "scan_ms": st["ms_scan"], "bin_ms": st["ms_bin"], "verify_ms": st["ms_verify"],
                "group_ms": st["ms_group"], "replay_ms": st["ms_replay"], "pipeline_ms": st["ms_total"],
                "comparisons": st["candidates"], "hits": st["hits"],
                "scan_Tcmp_per_s": st["candidates"] / st["ms_scan"] / 1e9, "algorithmic_TBps": algo / st["ms_scan"] / 1e9,
                "guides_per_s_kernels": a.guides / st["ms_total"] * 1e3}
if a.json and best:
    json.dump(best, open(a.json, "w"), indent=1)
