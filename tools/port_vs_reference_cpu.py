#!/usr/bin/env python3
"""Is the CPU port (oracle/issl_oracle.c, bench.py's `cpu_baseline.kind = "port"`) a fair stand-in for the compiled
reference?  Times oracle/_ref/isslScoreOfftargets (the reference's own sources, reference flags) next to
oracle/_build/oracle_score on the same index and query file, same thread count, in the BUILD container (the reference
does not travel to the GPU box), and checks that their stdouts are identical.

    python tools/port_vs_reference_cpu.py profiles/r03_port_vs_reference_cpu.json

BASELINE configs[0] (1 k guides x 1 M-line index) and configs[1] (50 M-line index; a 2 000-guide sample of the 10 k).
Each binary loads the index itself (the reference has no other mode), so every point is measured twice: with the
query file and with a ONE-guide query; the difference is the scoring time."""
import json, os, pathlib, subprocess, sys, tempfile, time
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
from synth import random_sites, random_sites_fast, random_guides, sigs_to_text

REF = ROOT / "oracle" / "_ref" / "isslScoreOfftargets"
REF_BUILD = ROOT / "oracle" / "_ref" / "isslCreateIndex"
PORT = ROOT / "oracle" / "_build" / "oracle_score"


def timed(cmd, env):
    t = time.perf_counter()
    out = subprocess.run(cmd, check=True, capture_output=True, env=env).stdout
    return time.perf_counter() - t, out


def decode(sigs):
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = np.empty((len(sigs), 21), dtype=np.uint8)
    for j in range(20):
        out[:, j] = lut[((sigs >> np.uint64(2 * j)) & np.uint64(3)).astype(np.int64)]
    out[:, 20] = ord("\n")
    return out.tobytes()


def point(tmp, label, n_lines, n_guides, threads):
    sigs, occ = (random_sites_fast(n_lines, seed=20261003, threads=8) if n_lines >= 20_000_000 else random_sites(n_lines, seed=20261003))
    guides = random_guides(sigs, n_guides, seed=777)
    sites_txt, issl, q, q1 = tmp / "sites.txt", tmp / "index.issl", tmp / "q.txt", tmp / "q1.txt"
    with open(sites_txt, "wb") as f:  # one line per occurrence, sorted (what isslCreateIndex expects)
        step = 1 << 22
        for lo in range(0, len(sigs), step):
            f.write(decode(np.repeat(sigs[lo:lo + step], occ[lo:lo + step])))
    t = time.perf_counter()
    subprocess.run([str(REF_BUILD), str(sites_txt), "20", "8", str(issl)], check=True, capture_output=True)
    t_build = time.perf_counter() - t
    sites_txt.unlink()
    q.write_bytes(decode(guides)); q1.write_bytes(decode(guides[:1]))
    env = dict(os.environ, OMP_NUM_THREADS=str(threads))
    rec = {"lines": n_lines, "distinct_sites": int(len(sigs)), "guides": n_guides, "threads": threads,
           "reference_index_build_s": t_build, "runs": {}}
    for thr in ("0", "75"):
        row = {}
        for name, exe in (("reference", REF), ("port", PORT)):
            best, best1, out = 1e9, 1e9, None
            for _ in range(2):
                dt, out = timed([str(exe), str(issl), str(q), "4", thr, "and"], env)
                dt1, _ = timed([str(exe), str(issl), str(q1), "4", thr, "and"], env)
                best, best1 = min(best, dt), min(best1, dt1)
            row[name] = {"wall_s": best, "load_only_s": best1, "scoring_s": best - best1,
                         "guides_per_s": n_guides / max(best - best1, 1e-9), "stdout": out}
        same = row["reference"].pop("stdout") == row["port"].pop("stdout")
        row["stdout_identical"] = same
        row["port_over_reference_scoring_time"] = row["port"]["scoring_s"] / row["reference"]["scoring_s"]
        rec["runs"][f"thr{thr}"] = row
        print(label, "thr", thr, json.dumps(row), flush=True)
    issl.unlink()
    return rec


def main():
    out = pathlib.Path(sys.argv[1]) if len(sys.argv) > 1 else None
    subprocess.run(["make", "-C", str(ROOT / "oracle"), "all", "ref"], check=True, capture_output=True)
    threads = len(os.sched_getaffinity(0))
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
        tmp = pathlib.Path(d)
        res = {"what": "tools/port_vs_reference_cpu.py in the build container: compiled reference (oracle/_ref, reference flags "
                       "-O3 -std=c++11 -fopenmp -mpopcnt) vs the C port (oracle/_build/oracle_score), same index, same query, "
                       f"OMP_NUM_THREADS={threads}; scoring_s = wall - wall of a one-guide query (index load)",
               "cpu": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?"),
               "configs0_1k_guides_1M_lines": point(tmp, "configs[0]", 1_000_000, 1000, threads),
               "configs1_2k_of_10k_guides_50M_lines": point(tmp, "configs[1]", 50_000_000, 2000, threads)}
    if out:
        out.write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
