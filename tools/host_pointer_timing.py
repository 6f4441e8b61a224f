import sys, time, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import crackling_amd as ca
from synth import random_sites_fast, random_guides_fast
sigs, occ = random_sites_fast(300_000_000, seed=20261003, threads=16)
ix = ca.IsslIndex.build_on_device(sigs, occ, device=0)
g = random_guides_fast(sigs, 100_000, seed=777)
for r in range(8):
    t=time.perf_counter(); mit, cfd = ix.score(g, 4, 75.0, "and"); dt=time.perf_counter()-t
    print(f"rep{r}: issl_score wall {dt*1e3:.2f} ms, kernels {ix.stats()['ms_total']:.2f} ms", flush=True)
print(float(mit.sum()), float(cfd.sum()))
