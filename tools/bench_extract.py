#!/usr/bin/env python3
"""Measurement of the off-target extraction step on one MI355X: synthetic genome of --mbp million bases in --records
records -> sorted site list.  Prints one JSON line: GPU end-to-end (host FASTA parse + H2D + match + radix sort + text
+ D2H) and the CPU oracle (C port of the reference's regex scan + sort, single thread) on a bounded sample."""
import argparse, json, sys, time, pathlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import crackling_amd as ca
from test_extract import oracle_extract

ap = argparse.ArgumentParser()
ap.add_argument("--mbp", type=float, default=200.0)
ap.add_argument("--records", type=int, default=24)
ap.add_argument("--sample-mbp", type=float, default=8.0)
a = ap.parse_args()
rng = np.random.default_rng(5)
n = int(a.mbp * 1e6)
per = n // a.records
parts = []
for r in range(a.records):
    s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=per)]
    parts.append(b">chr%d\n" % r + s.tobytes() + b"\n")
blob = b"".join(parts)
ca.extract_offtargets([blob[: 1 << 20]])  # warm-up (context, code objects)
t = time.perf_counter(); out = ca.extract_offtargets([blob]); gpu_s = time.perf_counter() - t
sites = len(out) // 21
m = int(a.sample_mbp * 1e6)
sample = blob[:m]
t = time.perf_counter(); ref = oracle_extract([sample]); cpu_s = time.perf_counter() - t
ok = ca.extract_offtargets([sample]) == ref
print(json.dumps({"metric": "off-target sites extracted per second (FASTA in host memory -> sorted text in host memory)",
                  "genome_mbp": a.mbp, "records": a.records, "sites": sites, "gpu_s": gpu_s, "gpu_sites_per_s": sites / gpu_s,
                  "gpu_mbp_per_s": a.mbp / gpu_s,
                  "cpu_port": {"sample_mbp": a.sample_mbp, "seconds": cpu_s, "sites_per_s": (len(ref) // 21) / cpu_s,
                               "mbp_per_s": a.sample_mbp / cpu_s, "threads": 1, "kind": "port"},
                  "parity_on_sample": bool(ok)}))
