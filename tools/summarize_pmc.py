#!/usr/bin/env python3
"""Per-kernel means of every counter found under a tools/profile_tail.sh output directory, as JSON.
    python tools/summarize_pmc.py gpurun_out/prof_tail profiles/r03_tail_pmc.json"""
import collections, csv, json, pathlib, sys
src = pathlib.Path(sys.argv[1])
out = collections.defaultdict(dict)
for f in sorted(src.glob("*/*counter_collection.csv")) + sorted(src.glob("*/*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            out[k][c] = {"mean": sum(v) / len(v), "n": len(v)}
for f in list(src.glob("kt/*kernel_stats.csv")) + list(src.glob("kt/*/*kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        out[r["Name"].split("(")[0]]["avg_ns"] = float(r["AverageNs"]); out[r["Name"].split("(")[0]]["calls"] = int(r["Calls"])
json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("avg_ns", 0) if isinstance(kv[1].get("avg_ns", 0), float) else 0)[:14]:
    print(k[:50], {c: (round(x["mean"]) if isinstance(x, dict) else x) for c, x in v.items()})
