#!/bin/bash
# Where does the one-shot process's upload time go, and does the way the previous process left matter?  (GPU box)
#   bash tools/cli_upload_probe.sh   -> gpurun_out/r05_cli_upload_probe.log
cd "$(dirname "$0")/.."
python3 - <<'PY'
import os, sys, time, subprocess, json, pathlib
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "tools")
import numpy as np
import crackling_amd as ca
from synth import random_sites_fast, random_guides_fast
import cli_end_to_end as e2e
tmp = pathlib.Path("/dev/shm")
sigs, occ = random_sites_fast(300_000_000, seed=21, threads=16)
ix = ca.IsslIndex.build_on_device(sigs, occ, device=0); issl = tmp / "probe.issl"; ix.write(issl); ix.close()
for n in (1_000_000, 10_000):
    e2e.write_query(tmp / f"probe_{n}.q", random_guides_fast(sigs, n, seed=5))
del sigs, occ
exe = "bin/isslScoreOfftargets"
def run(n, extra, label):
    env = dict(os.environ, ISSL_TIMING="1", ISSL_UPLOAD_TIMING="1", **extra)
    t = time.perf_counter()
    with open(tmp / "probe.out", "wb") as fh:
        r = subprocess.run([exe, str(issl), str(tmp / f"probe_{n}.q"), "4", "75", "and"], stdout=fh, stderr=subprocess.PIPE, env=env)
    wall = time.perf_counter() - t
    lines = r.stderr.decode().strip().splitlines()
    notes = " | ".join(l.replace("[issl upload] ", "") for l in lines if l.startswith("[issl upload]"))
    tj = json.loads(lines[-1])
    print(f"{label:34s} n={n:8d} wall {wall*1e3:7.0f} ms  runtime {tj['runtime_ms']:6.0f} upload {tj['upload_ms']:7.0f} score {tj['score_ms']:5.0f} total {tj['total_ms']:6.0f} :: {notes}", flush=True)
for n in (1_000_000, 10_000):
    for rep in range(3): run(n, {}, "quick exit, back to back")
    for rep in range(3): run(n, {"ISSL_TIDY_EXIT": "1"}, "tidy exit, back to back")
    for rep in range(2): run(n, {}, "quick exit, back to back (again)")
    for rep in range(2):
        time.sleep(2.0); run(n, {}, "quick exit, 2 s pause before")
    for rep in range(2): run(n, {"ISSL_NO_WARMUP": "1"}, "quick exit, no warm-up thread")
for p in (issl, tmp / "probe.out", tmp / "probe_1000000.q", tmp / "probe_10000.q"):
    p.unlink(missing_ok=True)
PY
