#!/usr/bin/env python3
"""Wall time of the drop-in executable on a synthetic index, the way Crackling runs it (one process per page):

    python tools/cli_end_to_end.py --sites 300000000 --guides 1000000 [--server] [--json out.json]

Writes the .issl and the query file to --tmp, runs `bin/isslScoreOfftargets <issl> <query> 4 75 and > out` with
ISSL_TIMING=1 (twice: the second run has the file in the page cache) and, with --server, twice more through a
resident scorer."""
import argparse, json, os, pathlib, subprocess, sys, time
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import crackling_amd as ca
from synth import random_sites_fast, random_guides

ap = argparse.ArgumentParser()
ap.add_argument("--sites", type=int, default=50_000_000)
ap.add_argument("--guides", type=int, default=1_000_000)
ap.add_argument("--tmp", default="/tmp")
ap.add_argument("--server", action="store_true")
ap.add_argument("--json", default=None)
a = ap.parse_args()
tmp = pathlib.Path(a.tmp)
issl, query, out = tmp / "e2e.issl", tmp / "e2e.query", tmp / "e2e.out"
t = time.time(); sigs, occ = random_sites_fast(a.sites, seed=21, threads=min(32, os.cpu_count() or 8))
ix = ca.IsslIndex.build_from_sites(sigs, occ); ix.write(issl)
print(f"index: {len(sigs)} distinct sites, {issl.stat().st_size / 1e9:.2f} GB, made in {time.time() - t:.1f}s", flush=True)
rng = np.random.default_rng(22)
guides = sigs[rng.integers(0, len(sigs), size=a.guides)] ^ (np.uint64(3) << (np.uint64(2) * rng.integers(0, 20, size=a.guides).astype(np.uint64)))
del sigs, occ, ix
seqs = ca.decode_guides(guides[:1000])
# query text: 2-bit decode, vectorised
letters = np.frombuffer(b"ACGT", dtype=np.uint8)
txt = np.empty((a.guides, 21), dtype=np.uint8)
for j in range(20):
    txt[:, j] = letters[((guides >> np.uint64(2 * j)) & np.uint64(3)).astype(np.int64)]
txt[:, 20] = ord("\n")
query.write_bytes(txt.tobytes())
assert query.read_text().splitlines()[:1000] == seqs
exe = str(ROOT / "bin" / "isslScoreOfftargets")
res = {"what": f"bin/isslScoreOfftargets on a {a.sites}-line index ({issl.stat().st_size / 1e9:.2f} GB .issl), "
               f"{a.guides} guides, 4 75 and; one process per call as Crackling does", "runs": []}


def run(label, env):
    t = time.time()
    with open(out, "wb") as fh:
        r = subprocess.run([exe, str(issl), str(query), "4", "75", "and"], stdout=fh, stderr=subprocess.PIPE, env=env)
    wall = time.time() - t
    assert r.returncode == 0, r.stderr.decode()
    err_lines = r.stderr.decode().strip().splitlines()
    timing = json.loads(err_lines[-1])
    notes = [l for l in err_lines[:-1] if l.startswith("[issl")]
    if notes:
        print("   " + " | ".join(notes), flush=True)
    n_lines = sum(1 for _ in open(out, "rb"))
    assert n_lines == a.guides
    res["runs"].append({"label": label, "wall_s": wall, "guides_per_s_wall": a.guides / wall, "timing": timing})
    print(label, f"wall {wall:.2f}s", json.dumps(timing), flush=True)


env = dict(os.environ, ISSL_TIMING="1")
run("process, first run", env)
run("process, file cached", env)
if a.server:
    sock = str(tmp / "e2e.sock")
    srv = subprocess.Popen([exe, "--serve", sock], stderr=subprocess.DEVNULL)
    try:
        for _ in range(200):
            if os.path.exists(sock):
                break
            time.sleep(0.05)
        env2 = dict(env, ISSL_SERVER=sock)
        run("resident server, first request (loads the index)", env2)
        run("resident server, index resident", env2)
    finally:
        subprocess.run([exe, "--stop", sock], capture_output=True)
        srv.wait(timeout=60)
for p in (issl, query, out):
    p.unlink(missing_ok=True)
if a.json:
    json.dump(res, open(a.json, "w"), indent=1)
