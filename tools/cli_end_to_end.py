#!/usr/bin/env python3
"""Wall time of the drop-in executable the way Crackling runs it (src/crackling/Crackling.py:767-778: one process per
page, `<binary> <issl> <query> 4 75 and > output`), one-shot and through the resident server:

    python tools/cli_end_to_end.py --sites 300000000 [--json out.json]

measure() is what bench.py calls for `extras.cli_end_to_end` (on the .issl its CPU-baseline leg has written)."""
import json
import os
import pathlib
import subprocess
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import numpy as np  # noqa: E402

EXE = str(ROOT / "bin" / "isslScoreOfftargets")


def write_query(path, guides):
    """The query file Crackling writes (Crackling.py:747-752): seq[0:20] + LF per guide."""
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    txt = np.empty((len(guides), 21), dtype=np.uint8)
    for j in range(20):
        txt[:, j] = letters[((guides >> np.uint64(2 * j)) & np.uint64(3)).astype(np.int64)]
    txt[:, 20] = ord("\n")
    pathlib.Path(path).write_bytes(txt.tobytes())


def _run(issl, query, out, env, args=("4", "75", "and"), exe=EXE):
    t = time.perf_counter()
    with open(out, "wb") as fh:
        r = subprocess.run([exe, str(issl), str(query), *args], stdout=fh, stderr=subprocess.PIPE, env=env, timeout=180)  # (a hung child must not hang the bench line)
    wall = time.perf_counter() - t
    if r.returncode != 0:
        raise RuntimeError(f"{exe} exited {r.returncode}: {r.stderr.decode(errors='replace')[-400:]}")
    timing = None
    for line in reversed(r.stderr.decode(errors="replace").strip().splitlines()):
        if line.startswith("{"):
            timing = json.loads(line)
            break
    return wall, timing


def measure(issl, pages, tmp, expected=None, server=True, exe=EXE, log=None, one_shot_runs=3, pause_s=1.5):
    """pages: [(label, uint64 guide signatures)].  expected(label) -> the stdout bytes the page must produce, or None.
    Returns {label: {"one_shot": [...], "resident": [...], "stdout_identical": bool}}; every run = wall seconds of the child
    process as the caller sees it (fork, exec, exit included) + the ISSL_TIMING stage breakdown the process printed.
    One-shot runs: the first of a page follows the previous process at once, the others after `pause_s` seconds --
    Crackling parses a page's output and filters the next page's guides between two scorer processes, and a process that
    starts while the driver is still taking back the 45 GB of its predecessor waits for it inside its own allocations
    (profiles/r05_cli_upload_probe.log: 1 - 2.4 s, erratic); every record says which kind it is."""
    tmp = pathlib.Path(tmp)
    say = log or (lambda *a: None)
    res = {}
    env = dict(os.environ, ISSL_TIMING="1")
    env.pop("ISSL_SERVER", None)
    queries = {}
    for label, guides in pages:
        q = tmp / f"cli_{os.getpid()}_{label}.query"
        write_query(q, guides)
        queries[label] = q
        res[label] = {"guides": int(len(guides)), "one_shot": [], "resident": [], "stdout_identical": None}
    out = tmp / f"cli_{os.getpid()}.out"

    def check(label):
        if expected is None:
            return
        want = expected(label)
        if want is None:
            return
        same = out.read_bytes() == want
        prev = res[label]["stdout_identical"]
        res[label]["stdout_identical"] = same if prev is None else (prev and same)

    try:
        for label, guides in pages:
            for i in range(one_shot_runs):
                pause = pause_s if i else 0.0
                time.sleep(pause)
                wall, timing = _run(issl, queries[label], out, env, exe=exe)
                check(label)
                res[label]["one_shot"].append({"wall_s": wall, "guides_per_s_wall": len(guides) / wall, "pause_before_s": pause, "timing": timing})
                say(f"[cli] {label} one-shot run {i} ({pause:g} s after the previous process): {wall*1e3:.0f} ms wall  {json.dumps(timing)}")
        if server:
            sock = str(tmp / f"cli_{os.getpid()}.sock")
            srv = subprocess.Popen([exe, "--serve", sock], stderr=subprocess.DEVNULL, env=env)
            try:
                for _ in range(400):
                    if os.path.exists(sock):
                        break
                    time.sleep(0.025)
                env2 = dict(env, ISSL_SERVER=sock)
                first = True
                for label, guides in pages:
                    for i in range(3 if not first else 4):
                        wall, timing = _run(issl, queries[label], out, env2, exe=exe)
                        check(label)
                        rec = {"wall_s": wall, "guides_per_s_wall": len(guides) / wall, "timing": timing}
                        if first:  # this request made the server open and upload the index
                            res["server_first_request"] = dict(rec, page=label)
                            first = False
                        else:
                            res[label]["resident"].append(rec)
                        say(f"[cli] {label} through the server, request {i}: {wall*1e3:.1f} ms wall  {json.dumps(timing)}")
            finally:
                try:
                    subprocess.run([exe, "--stop", sock], capture_output=True, timeout=60)
                except Exception:  # noqa: BLE001  (the server is waited for, and killed, below either way)
                    pass
                try:
                    srv.wait(timeout=60)
                except subprocess.TimeoutExpired:
                    srv.kill()
        for label, _ in pages:
            r = res[label]
            if r["one_shot"]:
                r["one_shot_best_wall_s"] = min(x["wall_s"] for x in r["one_shot"])
            if r["resident"]:
                r["resident_best_wall_s"] = min(x["wall_s"] for x in r["resident"])
                r["resident_guides_per_s"] = r["guides"] / r["resident_best_wall_s"]
    finally:
        for p in list(queries.values()) + [out]:
            pathlib.Path(p).unlink(missing_ok=True)
    return res


def main():
    import argparse
    import crackling_amd as ca
    from synth import random_sites_fast, random_guides_fast
    ap = argparse.ArgumentParser()
    ap.add_argument("--sites", type=int, default=50_000_000)
    ap.add_argument("--pages", default="1000000,10000")
    ap.add_argument("--tmp", default="/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
    ap.add_argument("--no-server", action="store_true")
    ap.add_argument("--exe", default=EXE, help="another build of the executable (A/B)")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    tmp = pathlib.Path(a.tmp)
    issl = tmp / f"e2e_{os.getpid()}.issl"
    t = time.time()
    sigs, occ = random_sites_fast(a.sites, seed=21, threads=min(16, os.cpu_count() or 8))
    ix = ca.IsslIndex.build_on_device(sigs, occ, device=0)
    ix.write(issl)
    print(f"index: {len(sigs)} distinct sites, {issl.stat().st_size / 1e9:.2f} GB, made in {time.time() - t:.1f}s", flush=True)
    pages = [(f"{n}_guides", random_guides_fast(sigs, n, seed=22 + i)) for i, n in enumerate(int(x) for x in a.pages.split(","))]
    scores = {}
    for label, g in pages:
        mit, cfd = ix.score(g, 4, 75.0, "and")
        scores[label] = ca.format_scores_native(g, mit, cfd, "and")
    ix.close()
    del sigs, occ
    allres = {}
    try:
        for exe in a.exe.split(","):   # (several builds of the executable: same-box A/B)
            print(f"== {exe}", flush=True)
            res = measure(issl, pages, tmp, expected=scores.get, server=not a.no_server, exe=exe, log=lambda *x: print(*x, flush=True))
            res["what"] = f"{exe} on a {a.sites}-line index, `<issl> <query> 4 75 and > out`, one process per page as Crackling runs it"
            print(json.dumps({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk not in ('one_shot', 'resident')}) for k, v in res.items()}, indent=1), flush=True)
            allres[exe] = res
    finally:
        issl.unlink(missing_ok=True)
    if a.json:
        json.dump(allres if len(allres) > 1 else next(iter(allres.values())), open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
