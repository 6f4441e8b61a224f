#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_scan.sh into the summary committed under profiles/.

    python tools/summarize_profile.py gpurun_out/prof2 profiles/r01_bench_n1

writes <dst>_kernel_stats.csv (copy of the --kernel-trace --stats table), <dst>_pmc.json (per-kernel means of
every collected counter) and prints the scan kernel's HBM traffic per launch with the gfx950 correction of
MI355X_MICROARCH.md (FETCH_SIZE counts 128-B requests of a wide coalesced stream as 64 B: x2; counter unit KiB)."""
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
if scan and "FETCH_SIZE" in out[scan]:
    fetch_kib = out[scan]["FETCH_SIZE"]["mean"]
    write_kib = out[scan].get("WRITE_SIZE", {"mean": 0.0})["mean"]
    summary["scan_hbm_bytes_per_launch"] = {
        "fetch_raw_KiB": fetch_kib, "write_raw_KiB": write_kib,
        "fetch_corrected_bytes": fetch_kib * 1024 * 2, "write_bytes": write_kib * 1024,
        "total_bytes": fetch_kib * 1024 * 2 + write_kib * 1024,
        "note": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B); KiB -> bytes",
    }
json.dump(summary, open(dst + "_pmc.json", "w"), indent=1, sort_keys=True)
print(json.dumps(summary.get("scan_hbm_bytes_per_launch"), indent=1))
for row in csv.DictReader(open(dst + "_kernel_stats.csv")):
    print(f'{row["Name"][:60]:60s} calls={row["Calls"]:>5s} avg_ns={float(row["AverageNs"]):12.0f} pct={row["Percentage"]}')
