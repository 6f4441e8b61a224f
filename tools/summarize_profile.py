#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_scan.sh into the summary committed under profiles/.

    python tools/summarize_profile.py gpurun_out/prof2 profiles/r02_bench_n1 [sites guides distribution]

writes <dst>_kernel_stats.csv (copy of the --kernel-trace --stats table), <dst>_pmc.json (per-kernel means of
every collected counter, the scan kernel's HBM traffic per launch with the gfx950 correction of MI355X_MICROARCH.md
-- FETCH_SIZE counts 128-B requests of a wide coalesced stream as 64 B: x2; counter unit KiB -- and the clock the chip
held during the scan: GRBM_GUI_ACTIVE / 8 XCDs / kernel duration).  With the three workload arguments the traffic is
also entered into profiles/scan_traffic.json, where bench.py looks its `roofline.traffic` up."""
import csv, json, sys, collections, pathlib, shutil

src, dst = pathlib.Path(sys.argv[1]), sys.argv[2]
shutil.copy(src / "kt" / "kt_kernel_stats.csv", dst + "_kernel_stats.csv")
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub, name in [("pmc_sq", "sq"), ("pmc_sq2", "sq2"), ("pmc_fetch", "fetch"), ("pmc_write", "write")]:
    f = src / sub / f"{name}_counter_collection.csv"
    if not f.exists():
        continue
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in pmc.items():
    out[k] = {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in cs.items()}
scan = next((k for k in out if "k_scan<" in k), None)
summary = {"kernels": out}
stats = {row["Name"].split("(")[0]: row for row in csv.DictReader(open(dst + "_kernel_stats.csv"))}
scan_row = next((row for name, row in stats.items() if "k_scan<" in name), None)
if scan_row:
    summary["scan_avg_launch_ns"] = float(scan_row["AverageNs"])
    summary["scan_calls"] = int(scan_row["Calls"])
if scan and "FETCH_SIZE" in out[scan]:
    fetch_kib = out[scan]["FETCH_SIZE"]["mean"]
    write_kib = out[scan].get("WRITE_SIZE", {"mean": 0.0})["mean"]
    summary["scan_hbm_bytes_per_launch"] = {
        "fetch_raw_KiB": fetch_kib, "write_raw_KiB": write_kib,
        "fetch_corrected_bytes": fetch_kib * 1024 * 2, "write_bytes": write_kib * 1024,
        "total_bytes": fetch_kib * 1024 * 2 + write_kib * 1024,
        "note": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B); KiB -> bytes",
    }
if scan and "GRBM_GUI_ACTIVE" in out[scan] and scan_row:
    # the counter is summed over the 8 XCDs; PMC passes serialise kernels, the duration is the un-profiled trace's
    summary["scan_effective_clock_GHz"] = out[scan]["GRBM_GUI_ACTIVE"]["mean"] / 8.0 / float(scan_row["AverageNs"])
json.dump(summary, open(dst + "_pmc.json", "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in summary.items() if k != "kernels"}, indent=1))
for name, row in stats.items():
    print(f'{name[:60]:60s} calls={row["Calls"]:>5s} avg_ns={float(row["AverageNs"]):12.0f} pct={row["Percentage"]}')
if len(sys.argv) >= 6 and "scan_hbm_bytes_per_launch" in summary:
    sites, guides, dist = int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    pruned = None  # which plan the profiled launches followed: from the bench line of the kernel-trace pass
    for line in open(src / "kt.log", errors="replace"):
        if line.startswith("{"):
            pruned = json.loads(line).get("roofline", {}).get("pruned")
    path = pathlib.Path(__file__).resolve().parent.parent / "profiles" / "scan_traffic.json"
    rec = json.loads(path.read_text()) if path.exists() else {}
    points = [p for p in rec.get("points", [])
              if (p.get("sites"), p.get("guides"), p.get("distribution"), p.get("pruned")) != (sites, guides, dist, pruned)]
    import hashlib
    sha = hashlib.sha256((pathlib.Path(__file__).resolve().parent.parent / "crackling_amd" / "csrc" / "issl_kernels.hip").read_bytes()).hexdigest()[:16]
    points.append({"sites": sites, "guides": guides, "distribution": dist, "pruned": pruned, "kernels_sha16": sha,
                   "hbm_bytes_per_launch": summary["scan_hbm_bytes_per_launch"]["total_bytes"],
                   "fetch_corrected_bytes": summary["scan_hbm_bytes_per_launch"]["fetch_corrected_bytes"],
                   "write_bytes": summary["scan_hbm_bytes_per_launch"]["write_bytes"],
                   # the same launches in the rocprofv3 --kernel-trace pass (every launch of the process, the first cold ones
                   # included): bench.py prices roofline.frac_trace on it
                   "trace_avg_launch_ms": summary.get("scan_avg_launch_ns", 0.0) * 1e-6 or None,
                   "trace_calls": summary.get("scan_calls"),
                   "source": f"{dst}_pmc.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `python3 bench.py "
                             f"--no-cpu-baseline --no-extras`; FETCH_SIZE x2 per MI355X_MICROARCH.md, KiB->bytes)"})
    path.write_text(json.dumps({"points": points}, indent=1) + "\n")
