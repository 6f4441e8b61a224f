#!/usr/bin/env python3
"""Regenerate profiles/rNN_batch_size_sweep.json on the GPU box (run through gpurun):

    python tools/sweep_batch.py gpurun_out/sweep.json [--big]

bench.py at 64..10k guides per step on the default 50M-line index, then tools/quick_perf.py at 100k guides on 50M,
300M (BASELINE configs[2]) and, with --big, 1G sites.  Every point is a child process (one GPU process at a time)."""
import json, subprocess, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parent.parent
out = sys.argv[1]
big = "--big" in sys.argv
res = {"what": "bench.py --guides G --steps 100 --no-cpu-baseline on one MI355X, 50M-line index (48.78M distinct sites); "
               "small G = HBM-bound regime (each bucket tile serves one guide), large G = VALU-bound", "by_guides": {}}
for g in (64, 256, 1024, 4096, 10000):
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--guides", str(g), "--steps", "100", "--no-cpu-baseline"],
                       stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True, text=True)
    d = json.loads(p.stdout.strip().splitlines()[-1])
    res["by_guides"][str(g)] = {"guides_per_s": d["value"], "ms_per_step": d["ms_per_step"], "scan_ms": d["kernel_ms"]["scan"],
                                "algorithmic_GBps": d["roofline"]["achieved"], "kernel_ms": d["kernel_ms"]}
    print(g, d["value"], d["ms_per_step"], flush=True)
    json.dump(res, open(out, "w"), indent=1)
points = [("guides_100k_50M_sites", 50_000_000), ("config3", 300_000_000)] + ([("sites_1e9", 1_000_000_000)] if big else [])
for name, sites in points:
    tmp = out + "." + name
    subprocess.run([sys.executable, str(ROOT / "tools" / "quick_perf.py"), "--sites", str(sites), "--guides", "100000",
                    "--thr", "75", "--reps", "4", "--json", tmp], check=True, stdout=sys.stderr)
    res[name] = json.load(open(tmp)); pathlib.Path(tmp).unlink()
    print(name, res[name]["scan_ms"], res[name]["scan_Tcmp_per_s"], flush=True)
    json.dump(res, open(out, "w"), indent=1)
