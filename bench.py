#!/usr/bin/env python3
"""Benchmark of the ISSL off-target scoring hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (guide binning, XOR/popcount scan, exact check of the noted candidates, hit grouping,
ordered MIT/CFD replay) over one batch of synthetic guides; index, guides and scores are resident in HBM.

Workloads (BASELINE.json):
  N = 1   configs[2]: 100 000 guides per step vs a 300 M-line synthetic human-scale ISSL index (the configuration the
          metric's target is quoted on).  north_star's 10 k-guide point and a 64-guide point (the HBM-bound regime of
          the scan) are measured on the same index after the timed region and reported as extras.
  N > 1   configs[3]: ONE batch of 1 000 000 guides per step, sharded over the N ranks in interleaved chunks
          (crackling_amd/sharding.py), same index replicated: built by rank 0, broadcast over RCCL/xGMI, scores gathered
          on rank 0 inside the timed region.  "scaling": "strong" -- the total work per step is fixed.
The index comes from tests/synth.random_sites_fast (sorted, ~2.5 % duplicated lines) and is built ON THE DEVICE
(issl_index_build_on_device), so the set-up stays under a minute.

Rank 0 prints ONE JSON line; see DESIGN.md section 3 for the accounting.  `roofline` describes the scan kernel (still the
longest of the step).  By default the scan is PRUNED: it compares a guide only with the successor-byte
groups of its buckets that can hold a hit, ~1/13.5 of the reference's comparisons on this workload, same results.
`frac` = useful VALU issue cycles of the comparisons the kernel MADE / available ones, with the 2.4 GHz peak clock as
denominator (the clock the chip holds under this load is lower, see profiles/); `hbm_frac` = measured HBM bytes of the
launch / time / 8 TB/s.  The SURVEY 8(d) algorithmic-bytes figure (8 B per comparison of the REFERENCE, no credit for
reuse or pruning) is kept as `algorithmic_over_hbm_peak`; `extras.whole_bucket_scan` runs the same batch with pruning
off; `hbm_regime` is north_star's own point, 10 000 guides per step, where the pruned scan is bound by HBM (two guides
per group: `hbm_pmc_frac` = PMC-measured bytes of a launch / its time / 8 TB/s).
`cpu_baseline` times the CPU oracle (oracle/issl_oracle.c, the restatement of the reference's OpenMP scorer) on a
bounded sample of the same workload -- baseline only, rank 0, N=1.
"""
import argparse
import json
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s HBM3E spec peak
PEAK_CLOCK_HZ = 2.4e9       # peak engine clock; under the scan's load the chip holds less (profiles/: GRBM_GUI_ACTIVE)
N_SIMD = 256 * 4
VALU_CYCLES_PER_INSTR = 2   # SIMD-32: a wave64 VALU instruction holds the issue port for two cycles
BROADCAST_LIMIT_S = 600.0   # N > 1: a rank whose image broadcast has not finished by then exits non-zero (watchdog below)


def valu_per_2048_cmp(pruned, max_dist):
    """Useful wave64 VALU instructions per guide and 2048 candidates in the scan's loop, counted in the compiled kernel
    (tools/isa_stats.py): whole buckets compare 16 positions -- 32 for the mismatch planes, 29 for
    the count, 1 hit test = 62.  The pruned scan compares 12 (the successor slice's four are known from the group a
    guide was placed in): 24 + 17 + 1 = 42 for the twelve of thirteen placements with one mismatch there (budget 3),
    24 + 22 + 1 = 47 for the thirteenth (budget 4); max_dist <= 2 places every guide once per bucket (budget = max_dist:
    24 + 19..20 + 1)."""
    if not pruned:
        return 62.0
    if pruned == 1:
        return 45.0
    return (12 * 42 + 47) / 13.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def kernels_sha16():
    """What the PMC points of profiles/scan_traffic.json are stamped with: the scan kernel's source."""
    import hashlib
    return hashlib.sha256((ROOT / "crackling_amd" / "csrc" / "issl_kernels.hip").read_bytes()).hexdigest()[:16]


def traffic_point(sites, guides, dist, pruned, key="hbm_bytes_per_launch"):
    """(HBM bytes per k_scan launch of this workload from a PMC profile -- or, key="trace_avg_launch_ms", the mean launch
    duration of the rocprofv3 kernel trace of the same passes --, where it came from), or (None, why not).  A point measured
    on another build of the kernels (its `kernels_sha16` differs from the source in this tree) is stale and is not used."""
    prof = ROOT / "profiles" / "scan_traffic.json"
    try:
        for rec in json.loads(prof.read_text()).get("points", []):
            if (rec.get("sites"), rec.get("guides"), rec.get("distribution"), rec.get("pruned")) == (sites, guides, dist, pruned):
                if rec.get("kernels_sha16") != kernels_sha16():
                    return None, f"stale: profiles/scan_traffic.json holds a point of kernel build {rec.get('kernels_sha16')}, this is {kernels_sha16()}"
                return rec.get(key), rec.get("source")
    except Exception as e:  # noqa: BLE001
        return None, f"{type(e).__name__}: {e}"
    return None, "no PMC point for this workload in profiles/scan_traffic.json"


def rccl_version(torch):
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception as e:  # noqa: BLE001
        return f"unknown ({type(e).__name__})"


def host_description():
    info = {"nproc": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
    info["cgroup_cpu_max"] = None
    info["effective_cores"] = info["affinity"]
    try:
        info["cgroup_cpu_max"] = open("/sys/fs/cgroup/cpu.max").read().strip()
        quota, period = info["cgroup_cpu_max"].split()
        if quota != "max":  # CPU time the job may use per period: more runnable threads than this only queue
            info["effective_cores"] = max(1, min(info["affinity"], int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                info["cpu_model"] = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return info


def cpu_baseline(issl_path, guides, gpu_scores, max_dist, thr, method, budget_s):
    """Oracle (port of the reference OpenMP scorer) on the host cores: thread sweep at threshold 0 (full scan, no
    early exit: the setting comparable with the roofline run) and at the product threshold, bounded samples."""
    import oracle_util as ou
    host = host_description()
    t0 = time.perf_counter()
    ix = ou.OracleIndex(issl_path)
    load_s = time.perf_counter() - t0
    affinity, cores = host["affinity"], host["effective_cores"]
    # thread counts up to 4x the CPU share of the job (beyond that the cgroup quota only queues threads: 256 threads on a
    # 16-core share measured half the rate of 32), always including the share itself and a 2x oversubscription
    sweep_threads = sorted({t for t in (8, 16, 32, 64, 128, cores, 2 * cores, affinity) if t <= min(affinity, 4 * cores)})
    # one guide on one thread: what a guide costs (sizes the samples)
    t0 = time.perf_counter()
    ix.score(guides[:1], max_dist, 0.0, method, threads=1)
    per_guide_s = max(time.perf_counter() - t0, 1e-4)
    per_point_s = budget_s / (2 * len(sweep_threads))
    sweep = {}
    best = {}
    at = 0
    parity = True
    for label, t_val in (("thr0", 0.0), (f"thr{thr:g}", thr)):
        sweep[label] = {}
        for threads in sweep_threads:
            n = int(min(len(guides), max(threads, min(threads, cores) * per_point_s / per_guide_s)))
            if at + n > len(guides):  # small batches: the samples of later points start over at the first guide
                at = 0
            sample = guides[at:at + n]
            t0 = time.perf_counter()
            mit, cfd = ix.score(sample, max_dist, t_val, method, threads=threads)
            dt = time.perf_counter() - t0
            rate = n / dt
            sweep[label][str(threads)] = {"guides": n, "seconds": round(dt, 3), "guides_per_s": rate}
            if t_val == thr:  # the GPU batch was scored at this threshold: same guides must give the same bits
                gm, gc = gpu_scores
                parity = parity and bool(np.array_equal(gm[at:at + n].view(np.uint64), mit.view(np.uint64)) and
                                         np.array_equal(gc[at:at + n].view(np.uint64), cfd.view(np.uint64)))
            if label not in best or rate > best[label][1]:
                best[label] = (threads, rate, n, dt)
            at += n
    ix.close()
    key = f"thr{thr:g}"
    threads, rate, n, dt = best[key]
    port = {
        "value": rate, "unit": "guides/s", "cores": threads, "kind": "port",
        "sample": f"{n} guides of the batch, same index, OpenMP over guides with {threads} threads (best of the sweep "
                  f"{sweep_threads}), threshold {thr:g}, {dt:.2f} s wall (scan+score only, index already in memory)",
        "value_thr0": best["thr0"][1], "cores_thr0": best["thr0"][0],
        "single_thread_seconds_per_guide_thr0": per_guide_s,
        "sweep": sweep, "host": host, "index_load_s": load_s, "parity_on_sample": parity,
        "note": "thr0 = no early exit (isslScoreOfftargets.cpp:326: maximum_sum = +inf), the full five-bucket scan the "
                "GPU always does; the product threshold lets the CPU stop early on promiscuous guides.  kind 'port': the "
                "oracle's restatement of the reference scorer; timed beside the compiled reference in the build container it "
                "took 1.54x (threshold 0) / 0.78x (threshold 75) the reference's scoring time (profiles/r03_port_vs_reference_cpu.json, "
                "noisy VM): read the value as the reference's to within that factor",
    }
    ref = reference_baseline(issl_path, guides, gpu_scores, max_dist, thr, method, threads, rate)
    if ref is None:
        return port
    ref["host"] = host
    ref["port"] = {k: port[k] for k in ("value", "cores", "sample", "value_thr0", "cores_thr0", "single_thread_seconds_per_guide_thr0",
                                         "sweep", "index_load_s", "parity_on_sample", "note")}
    ref["parity_on_sample"] = bool(ref["parity_on_sample"] and parity)
    return ref


def reference_baseline(issl_path, guides, gpu_scores, max_dist, thr, method, threads, port_rate, seconds=10.0):
    """The compiled reference itself (oracle/_ref/isslScoreOfftargets: the reference's own sources, built by oracle/Makefile
    where they lie, never copied) on a sample of the batch, when the binary is there: `kind: "reference"`.  It loads the
    index on every call (it has no other mode), so it runs twice -- with the sample and with ONE guide -- and the scoring
    time is the difference.  None when the binary is absent or anything goes wrong: the port's figure stands then."""
    import subprocess
    exe = ROOT / "oracle" / "_ref" / "isslScoreOfftargets"
    if not exe.exists():
        return None
    try:
        import crackling_amd as ca
        sys.path.insert(0, str(ROOT / "tools"))
        from cli_end_to_end import write_query
        n = int(min(len(guides), max(threads, port_rate * seconds)))
        tmp = os.path.dirname(issl_path)
        q, q1 = f"{tmp}/bench_{os.getpid()}_ref.q", f"{tmp}/bench_{os.getpid()}_ref1.q"
        write_query(q, guides[:n])
        write_query(q1, guides[:1])
        env = dict(os.environ, OMP_NUM_THREADS=str(threads))
        try:
            def run(query):
                t0 = time.perf_counter()
                r = subprocess.run([str(exe), issl_path, query, str(max_dist), f"{thr:g}", method], capture_output=True, env=env, timeout=180)
                if r.returncode != 0:
                    raise RuntimeError(f"reference exited {r.returncode}: {r.stderr.decode(errors='replace')[-200:]}")
                return time.perf_counter() - t0, r.stdout
            t_load, _ = run(q1)
            t_full, out = run(q)
        finally:
            for f in (q, q1):
                if os.path.exists(f):
                    os.unlink(f)
        scoring = max(t_full - t_load, 1e-6)
        gm, gc = gpu_scores
        same = out == ca.format_scores_native(guides[:n], gm[:n], gc[:n], method)
        return {
            "value": n / scoring, "unit": "guides/s", "cores": threads, "kind": "reference",
            "sample": f"the first {n} guides of the batch through oracle/_ref/isslScoreOfftargets (the compiled reference, Makefile flags "
                      f"-O3 -std=c++11 -fopenmp -mpopcnt), OMP_NUM_THREADS={threads}, `<issl> <query> {max_dist} {thr:g} {method}`: "
                      f"{t_full:.2f} s wall, minus {t_load:.2f} s for a one-guide query (index load) = {scoring:.2f} s of scoring",
            "wall_s": t_full, "load_only_s": t_load, "scoring_s": scoring,
            "parity_on_sample": bool(same),
            "note": "stdout of the reference on the sample compared byte for byte with the GPU's scores as text (parity_on_sample); "
                    "`port` = the same sweep with the oracle's restatement (thread sweep, threshold 0 and the product threshold)",
        }
    except Exception as e:  # noqa: BLE001
        log(f"[bench] reference baseline not measured: {type(e).__name__}: {e}")
        return None


def main():
    # stdout carries exactly one JSON line: anything a library prints there (RCCL's version banner does) is sent
    # to stderr instead, and the line is written to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--spinup-ms", type=float, default=50.0,
                    help="untimed back-to-back scoring before the warm-up steps: the GPU needs some tens of ms of load "
                         "to reach its sustained clocks")
    ap.add_argument("--sites", type=int, default=300_000_000, help="lines of the synthetic site list")
    ap.add_argument("--guides", type=int, default=None,
                    help="guides per step over ALL ranks (default: 100 000 at N=1 = configs[2], 1 000 000 at N>1 = configs[3])")
    ap.add_argument("--threshold", type=float, default=75.0)
    ap.add_argument("--max-dist", type=int, default=4)
    ap.add_argument("--method", default="and")
    ap.add_argument("--dist", choices=["uniform", "markov"], default="uniform",
                    help="site distribution: iid uniform bases (BASELINE configs) or an AT-rich order-3 Markov chain")
    ap.add_argument("--chunk", type=int, default=4096, help="guides per interleaved shard chunk (N>1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=16.0, help="CPU seconds for the port's thread sweep (the compiled reference takes ~10 s of scoring + two index loads on top)")
    ap.add_argument("--no-extras", action="store_true", help="skip the 10k-guide, 64-guide and host-pointer points")
    ap.add_argument("--no-cli", action="store_true", help="skip extras.cli_end_to_end (bin/isslScoreOfftargets as Crackling runs it)")
    ap.add_argument("--dump-scores", default=None,
                    help="rank 0 writes the scores of the last timed step, in input order, to this .npz (tests: the N>1 "
                         "path against the single-process result)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    n_total = a.guides if a.guides else (100_000 if world == 1 else 1_000_000)

    import crackling_amd as ca  # loads libissl_hip.so (fails loudly if missing)
    import torch
    import torch.distributed as dist
    from crackling_amd import sharding
    from synth import random_sites, random_sites_fast, random_guides_fast, markov_sites_fast

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU path to time)")
    # ISSL_BENCH_DEVICE / ISSL_BENCH_BACKEND: rehearsal of the N>1 path on a box with ONE GPU (tests/test_multirank_one_gpu.py):
    # every rank on the same device, collectives over gloo (RCCL refuses two ranks on one GPU)
    dev_id = int(os.environ.get("ISSL_BENCH_DEVICE", local_rank))
    backend = os.environ.get("ISSL_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_id)
    dev = torch.device("cuda", dev_id)
    # Under torch.distributed.run (RANK set) the process group is used even for one rank, so that the RCCL
    # path (broadcast of the image, gather of the scores, barrier, max-reduce) runs on a 1-GPU box too.
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # ---- index: rank 0 builds it on its GPU, the image is broadcast over RCCL/xGMI ----------------------------------
    timings = {}
    index = None
    sigs = None
    if rank == 0:
        t0 = time.perf_counter()
        if a.dist == "markov":
            sigs, occ = markov_sites_fast(a.sites, seed=20261003, threads=min(16, os.cpu_count() or 8))
        elif a.sites >= 20_000_000:
            sigs, occ = random_sites_fast(a.sites, seed=20261003, threads=min(16, os.cpu_count() or 8))
        else:
            sigs, occ = random_sites(a.sites, seed=20261003)
        timings["synth_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        index = ca.IsslIndex.build_on_device(sigs, occ, device=dev_id)
        timings["device_build_s"] = time.perf_counter() - t0
        del occ
    if use_dist:
        # Watchdog: a broadcast that hangs (a rank that never joined, a dead link) must end the job with a message and a
        # non-zero status instead of running into the launcher's limit.  A timer thread, never an exec.
        import threading

        def _stuck():
            log(f"[bench] rank {rank}: image broadcast not finished after {BROADCAST_LIMIT_S:.0f} s -- giving up "
                f"(world {world}, backend {backend}, device {dev_id})")
            os._exit(3)
        watchdog = threading.Timer(BROADCAST_LIMIT_S, _stuck)
        watchdog.daemon = True
        watchdog.start()
        log(f"[bench] rank {rank}/{world}: device {dev_id}, backend {backend}, entering the image broadcast")
        bstats = {}
        index, timings["broadcast_s"] = sharding.broadcast_image(dist, torch, index, dev, stats=bstats)
        timings["hbm_used_peak_bytes_in_broadcast"] = bstats.get("hbm_used_peak_bytes")
        watchdog.cancel()
        log(f"[bench] rank {rank}: image attached after {timings['broadcast_s']:.2f} s of broadcast; HBM in use at the high-water mark "
            f"of the broadcast on this rank: {(bstats.get('hbm_used_peak_bytes') or 0) / 1e9:.1f} GB (image {bstats.get('image_bytes', 0) / 1e9:.1f} GB"
            f"{'; rank 0 holds the built image and the tensor it is copied into' if rank == 0 else ''})")
    hdr = index.header
    image_bytes = index.device_bytes()
    if rank == 0:
        log(f"[bench] index: {hdr['n_sites']} distinct sites ({a.sites} lines), image {image_bytes/1e9:.2f} GB, {timings}")

    # ---- guides: one batch of n_total guides, every rank takes its interleaved shard; resident in HBM ---------------
    if rank == 0:
        all_guides = random_guides_fast(sigs, n_total, seed=777)
        g_all = torch.from_numpy(all_guides.view(np.int64)).to(dev)
    else:
        g_all = torch.empty(n_total, dtype=torch.int64, device=dev)
    cdev = "cpu" if backend == "gloo" else dev   # where the small control tensors of the collectives live
    if use_dist:
        if backend == "gloo":
            host = g_all.cpu()
            dist.broadcast(host, 0)
            g_all.copy_(host)
        else:
            dist.broadcast(g_all, 0)
    layout = sharding.ShardLayout(torch, n_total, world, a.chunk if world > 1 else None, dev)
    d_guides = g_all[layout.index_tensors[rank]].contiguous()
    n_mine = int(d_guides.numel())
    guides = d_guides.cpu().numpy().view(np.uint64)
    d_mit = torch.empty(n_mine, dtype=torch.float64, device=dev)
    d_cfd = torch.empty(n_mine, dtype=torch.float64, device=dev)
    # every step of the timed region writes its own output buffers (steps run back to back inside the library)
    out_mit = torch.empty(a.steps, n_mine, dtype=torch.float64, device=dev)
    out_cfd = torch.empty(a.steps, n_mine, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def step(g=d_guides, m=d_mit, c=d_cfd):
        index.score_device_async(g, m, c, a.max_dist, a.threshold, a.method, stream=None)

    def settle(fn):
        """First batches on an index may have to grow the scratch buffers: repeat until a batch goes through."""
        fn()
        while not index.finish(stream):
            fn()

    lanes = index.get_option("lanes")
    index.set_option("scan_events", 1)  # the HIP event pair around every k_scan launch (roofline.frac_events)
    settle(step)
    settle(step)  # (both lanes' scratch buffers)
    if a.spinup_ms > 0:  # part of the set-up, reported as config.spinup_ms
        t_spin = time.perf_counter()
        while (time.perf_counter() - t_spin) * 1e3 < a.spinup_ms:
            step()
            index.finish(stream)
    for _ in range(max(a.warmup, 1)):
        settle(step)
    if use_dist:  # warms the point-to-point channels the gather uses
        sharding.gather_scores(dist, torch, layout, out_mit, out_cfd, device=dev)
    barrier()
    # Timed region: K steps enqueued back to back on the library's internal stream (no host round trip between
    # steps), one synchronisation at the end, then ONE gather of all scores to rank 0 (16 B per guide and step).
    # The library keeps a HIP event pair around every k_scan launch on the stream the kernel runs on;
    # stats()["ms_scan"] is their mean over the K launches.  Inputs are resident and ready (barrier above), hence
    # no stream dependency on the way in.
    # (scan_events = 1: the pair around EVERY launch, as the roofline contract asks; the library's default records it around the
    # first batch after a finish only -- an event record costs ~5 us of stream time)
    t0 = time.perf_counter()
    for i in range(a.steps):
        index.score_device_async(d_guides, out_mit[i], out_cfd[i], a.max_dist, a.threshold, a.method, stream=None)
    if not index.finish(stream):
        raise SystemExit("scratch buffers grew inside the timed region: warm-up too short")
    t_scored = time.perf_counter()
    gathered = (None, None)
    if use_dist:  # final gather of the scores, back in input order on rank 0
        gathered = sharding.gather_scores(dist, torch, layout, out_mit, out_cfd, device=dev)
    barrier()
    elapsed = time.perf_counter() - t0
    gather_s = time.perf_counter() - t_scored
    if a.dump_scores and rank == 0:  # scores of the last timed step in input order
        last = (gathered[0][-1], gathered[1][-1]) if use_dist else (out_mit[-1], out_cfd[-1])
        np.savez(a.dump_scores, guides=all_guides, mit=last[0].cpu().numpy(), cfd=last[1].cpu().numpy())
    st = index.stats()
    scan_ms = st["ms_scan"]
    # Stage breakdown (bin / verify / group / replay) from a short UNTIMED pass: the timed region records only the event
    # pair around the scan, because every further event record costs ~4 us of stream time.
    # One lane for it: in the timed region consecutive batches alternate between two workspaces / streams, so that the
    # short kernels of one batch run beside the scan of the next; a kernel's time is then not its own.
    index.set_option("stage_timing", 1).set_option("lanes", 1)
    for _ in range(3):
        step()
    index.finish(stream)
    index.set_option("stage_timing", 0).set_option("lanes", lanes)
    stages = index.stats()
    # The same K steps once more as the library runs them when nobody asks for the event pairs (its default: one pair per burst):
    # reported beside the line, never as `value`.
    default_events_ms = None
    if world == 1:
        index.set_option("scan_events", 2)
        t1 = time.perf_counter()
        for i in range(a.steps):
            index.score_device_async(d_guides, out_mit[i], out_cfd[i], a.max_dist, a.threshold, a.method, stream=None)
        if index.finish(stream):
            default_events_ms = (time.perf_counter() - t1) * 1e3 / a.steps
        index.set_option("scan_events", 1)
    per_rank = None
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        mine_t = torch.tensor([scan_ms, stages["ms_total"], float(n_mine), float(st["candidates"]), timings.get("broadcast_s", 0.0),
                               float(image_bytes), float(dev_id), float(timings.get("hbm_used_peak_bytes_in_broadcast") or 0)],
                              dtype=torch.float64, device=cdev)
        allr = [torch.empty_like(mine_t) for _ in range(world)] if rank == 0 else None
        dist.gather(mine_t, allr, dst=0)
        if rank == 0:
            per_rank = [{"rank": r, "scan_ms": float(x[0]), "pipeline_ms": float(x[1]), "guides": int(x[2]),
                         "comparisons": int(x[3]), "broadcast_s": float(x[4]), "image_bytes": int(x[5]), "device": int(x[6]),
                         "hbm_used_peak_bytes_in_broadcast": int(x[7])}
                        for r, x in enumerate(allr)]

    extras = {}
    if world > 1 and not a.no_extras:
        # the weak-scaling companion of the strong figure above: every rank scores configs[2]'s per-GPU batch (100 000
        # guides, or its whole shard if that is smaller) for a few steps, no gather; max over ranks
        n_weak = min(100_000, n_mine)
        g = d_guides[:n_weak].contiguous()
        m = torch.empty(n_weak, dtype=torch.float64, device=dev)
        c = torch.empty_like(m)
        settle(lambda: step(g, m, c))
        settle(lambda: step(g, m, c))
        weak_steps = 10
        barrier()
        t1 = time.perf_counter()
        for _ in range(weak_steps):
            step(g, m, c)
        weak_ok = index.finish(stream)  # False: a batch ran out of scratch space and would have to be re-run
        barrier()
        tt = torch.tensor([time.perf_counter() - t1 if weak_ok else float("inf")], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        if rank == 0 and float(tt.item()) != float("inf"):
            dt = float(tt.item())
            extras["weak_scaling_point"] = {"guides_per_gpu_per_step": n_weak, "steps": weak_steps, "ms_per_step": dt * 1e3 / weak_steps,
                                            "guides_per_s_all_gpus": world * n_weak * weak_steps / dt,
                                            "note": "every rank its own batch of this size per step, barrier + max over ranks, no gather"}
    if rank == 0 and world == 1 and not a.no_extras:
        try:  # extra measurement points must never cost the line itself
            # north_star's point: 10 000 guides per step against the same index
            index.set_option("scan_events", 2)  # the library's default: no event pair inside a burst (the kernel's own stamps time the scan)
            for label, n_small, reps in (("north_star_10k_guides", 10_000, 30), ("hbm_regime_64_guides", 64, 200)):
                if n_small >= n_mine:
                    continue
                g = d_guides[:n_small].contiguous()
                m = torch.empty(n_small, dtype=torch.float64, device=dev)
                c = torch.empty_like(m)
                settle(lambda: step(g, m, c))
                settle(lambda: step(g, m, c))
                for _ in range(5):
                    step(g, m, c)
                index.finish(stream)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(reps):
                    step(g, m, c)
                if not index.finish(stream):  # a batch would have to be re-run: no number for this point
                    continue
                dt = time.perf_counter() - t1
                s2 = index.stats()
                pmc, _src = traffic_point(a.sites, n_small, a.dist, s2["pruned"])
                extras[label] = {
                    "hbm_pmc_bytes_per_launch": pmc, "hbm_pmc_frac": (pmc / s2["ms_scan"] / 1e6 / HBM_PEAK_GBS) if pmc else None,
                    "pruned": s2["pruned"],
                    "guides_per_step": n_small, "steps": reps, "ms_per_step": dt * 1e3 / reps, "guides_per_s": n_small * reps / dt,
                    "scan_ms": s2["ms_scan"], "comparisons_per_launch": s2["candidates"],
                    "scan_Tcmp_per_s": s2["candidates"] / s2["ms_scan"] / 1e9,
                    "scan_units_per_launch": s2["scan_tiles"],
                    "algorithmic_GBps": 8.0 * s2["candidates"] / s2["ms_scan"] / 1e6,
                }
            index.set_option("scan_events", 1)
            # the same step sustained: the timed region above is 20 steps = 50 ms, shorter than the chip's power management
            # takes to settle and than a 5-s utilisation sampler's period -- here ~6 s of back-to-back steps in bursts of 50
            # (one synchronisation per burst), same inputs, outputs into a ring of the timed region's buffers
            try:
                sus_steps, t1 = 0, time.perf_counter()
                while time.perf_counter() - t1 < 6.0:
                    for i in range(50):
                        index.score_device_async(d_guides, out_mit[i % a.steps], out_cfd[i % a.steps], a.max_dist, a.threshold, a.method, stream=None)
                    if not index.finish(stream):
                        raise RuntimeError("scratch buffers grew inside the sustained run")
                    sus_steps += 50
                dt = time.perf_counter() - t1
                extras["sustained"] = {"steps": sus_steps, "seconds": dt, "ms_per_step": dt * 1e3 / sus_steps,
                                       "guides_per_s": n_mine * sus_steps / dt, "scan_ms_last_burst": index.stats()["ms_scan"],
                                       "note": "the default step run back to back for ~6 s (bursts of 50 steps, one synchronisation per burst): "
                                               "what the chip sustains once clocks and power have settled"}
            except Exception as e:  # noqa: BLE001
                extras["sustained"] = {"error": f"{type(e).__name__}: {e}"}
            # two workspaces, consecutive batches alternating between them (options lanes = 2 / 3): a software pipeline --
            # 2: the short kernels of one batch beside the scan of the next; 3: only a batch's binning beside the batch before it
            for label, lanes_opt, note in (
                    ("two_lanes", 2, "option lanes=2 (not the default): a software pipeline over two workspaces -- the scans of "
                                     "consecutive batches one after the other, verify / group / replay of a batch on a high-priority "
                                     "stream beside the next batch's scan; every kernel then shares the chip (scan_ms is the launch's "
                                     "own span)"),
                    ("bin_ahead", 3, "option lanes=3 (not the default): two workspaces; only the binning of a batch (seven short launches) "
                                     "runs beside the batch before it, its scan waits for that batch's replay")):
                index.set_option("lanes", lanes_opt)
                try:
                    def step2(i):  # consecutive batches are in flight together: every one its own output buffers
                        index.score_device_async(d_guides, out_mit[i % a.steps], out_cfd[i % a.steps], a.max_dist, a.threshold, a.method, stream=None)
                    settle(step)
                    settle(step)
                    for i in range(4):
                        step2(i)
                    index.finish(stream)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    reps = a.steps
                    for i in range(reps):
                        step2(i)
                    ok2 = index.finish(stream)
                    dt = time.perf_counter() - t1
                    s2 = index.stats()
                    if ok2:  # (a RETRY would mean batches to re-run: no number then)
                        same = bool(torch.equal(out_mit[(reps - 1) % a.steps], out_mit[0]) and torch.equal(out_cfd[(reps - 1) % a.steps], out_cfd[0]))
                        extras[label] = {
                            "guides_per_step": n_mine, "steps": reps, "ms_per_step": dt * 1e3 / reps, "guides_per_s": n_mine * reps / dt,
                            "scan_ms": s2["ms_scan"], "scan_ms_events": s2["ms_scan_events"], "scores_equal_across_steps": same, "note": note,
                        }
                finally:
                    index.set_option("lanes", lanes)
            # the same batch with the pruned scan switched off: every bucket of every guide compared in full, which is
            # what the reference's loop (:344) does and what round 1 and the first half of round 2 measured
            index.set_option("prune", 0)
            try:
                settle(step)
                settle(step)
                for _ in range(2):
                    step()
                index.finish(stream)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                reps = 5
                for _ in range(reps):
                    step()
                if not index.finish(stream):
                    raise RuntimeError("whole-bucket point: scratch buffers grew inside the measurement")
                dt = time.perf_counter() - t1
                s2 = index.stats()
                extras["whole_bucket_scan"] = {
                    "guides_per_step": n_mine, "steps": reps, "ms_per_step": dt * 1e3 / reps, "guides_per_s": n_mine * reps / dt,
                    "scan_ms": s2["ms_scan"], "comparisons_per_launch": s2["candidates"],
                    "scan_Tcmp_per_s": s2["candidates"] / s2["ms_scan"] / 1e9,
                    "valu_frac": s2["candidates"] / 2048.0 * valu_per_2048_cmp(0, a.max_dist) * VALU_CYCLES_PER_INSTR / (N_SIMD * PEAK_CLOCK_HZ * s2["ms_scan"] * 1e-3),
                    "scan_units_per_launch": s2["scan_tiles"],
                    "note": "option prune=0: the scan kernel works through whole buckets (13.5 x the comparisons on this workload)",
                }
            finally:
                index.set_option("prune", -1)
                settle(step)
                settle(step)
            # the caller-visible host entry point: guides from host memory, scores back to host memory, one sync per call
            index.score(guides, a.max_dist, a.threshold, a.method)   # (untimed: the handle's first call pins its staging memory)
            t1 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                hm, hc = index.score(guides, a.max_dist, a.threshold, a.method)
            dt = (time.perf_counter() - t1) / reps
            extras["host_pointer_entry"] = {"ms_per_step": dt * 1e3, "guides_per_s": n_mine / dt,
                                            "note": "issl_score(): guides in and scores out over PCIe, one synchronisation per call"}
            # Batches of half a million guides -- the pieces `issl_score` cuts a page of Crackling's into (the most that get hit slots):
            # a unit the scan fetches serves five times the guides, and a step costs less per guide than the line's own 100 000
            try:
                n_big = 500_000
                if a.sites >= 50_000_000 and n_mine <= n_big // 2:
                    gb = torch.from_numpy(random_guides_fast(sigs, n_big, seed=778).view(np.int64)).to(dev)
                    mb = torch.empty(n_big, dtype=torch.float64, device=dev)
                    cb = torch.empty_like(mb)
                    for _ in range(3):
                        step(gb, mb, cb)
                        index.finish(stream)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    reps = 6
                    for _ in range(reps):
                        step(gb, mb, cb)
                    if index.finish(stream):
                        dt = time.perf_counter() - t1
                        s2 = index.stats()
                        extras["batch_500k_guides"] = {"guides_per_step": n_big, "steps": reps, "ms_per_step": dt * 1e3 / reps,
                                                       "ms_per_100k_guides": dt * 1e3 / reps / (n_big / 1e5), "guides_per_s": n_big * reps / dt,
                                                       "scan_ms": s2["ms_scan"], "note": "not the line's workload (BASELINE configs[2] is 100 000 guides per step): the same "
                                                       "index scored in batches of 500 000 guides"}
                    del gb, mb, cb
            except Exception as e:  # noqa: BLE001
                extras["batch_500k_guides"] = {"error": f"{type(e).__name__}: {e}"}
            # the same line on a SKEWED index of the same size (`--dist markov` run by itself gives the full record): an
            # AT-rich order-3 Markov chain, seven times the hits, most guides leave through the early exit -- the tail's
            # kernels carry the step there
            if a.dist == "uniform" and a.sites >= 50_000_000:
                t1 = time.perf_counter()
                sk_sigs, sk_occ = markov_sites_fast(a.sites, seed=20261003, threads=min(16, os.cpu_count() or 8))
                sk_index = ca.IsslIndex.build_on_device(sk_sigs, sk_occ, device=dev_id)
                sk_guides = random_guides_fast(sk_sigs, n_mine, seed=777)
                sk_setup = time.perf_counter() - t1
                try:
                    sk_d = torch.from_numpy(sk_guides.view(np.int64)).to(dev)

                    def sk_step(i=0):
                        sk_index.score_device_async(sk_d, out_mit[i % a.steps], out_cfd[i % a.steps], a.max_dist, a.threshold, a.method, stream=None)

                    for _ in range(4):  # (scratch buffers settle: a batch that needs larger ones is repeated)
                        sk_step()
                        while not sk_index.finish(stream):
                            sk_step()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    reps = max(5, min(a.steps, 10))
                    for i in range(reps):
                        sk_step(i)
                    ok_sk = sk_index.finish(stream)
                    dt = time.perf_counter() - t1
                    sk_index.set_option("stage_timing", 1)
                    for _ in range(2):
                        sk_step()
                    sk_index.finish(stream)
                    s2 = sk_index.stats()
                    if ok_sk:
                        extras["skewed_index"] = {
                            "distribution": "markov", "distinct_sites": int(len(sk_sigs)), "guides_per_step": n_mine, "steps": reps,
                            "ms_per_step": dt * 1e3 / reps, "guides_per_s": n_mine * reps / dt, "hits_per_step": s2["hits"],
                            "kernel_ms": {"bin": s2["ms_bin"], "scan": s2["ms_scan"], "verify": s2["ms_verify"], "group": s2["ms_group"],
                                          "replay": s2["ms_replay"], "pipeline": s2["ms_total"]},
                            "setup_s": sk_setup,
                        }
                finally:
                    sk_index.close()
                    del sk_sigs, sk_occ
        except Exception as e:  # noqa: BLE001
            extras["error"] = f"{type(e).__name__}: {e}"

    if rank == 0:
        ms_per_step = elapsed * 1e3 / a.steps
        value = n_total * a.steps / elapsed
        cmp_per_launch = st["candidates"]            # comparisons the scan kernel made (and counted)
        ref_cmp = st["reference_comparisons"]        # comparisons the reference makes for the same batch (SURVEY 8d's unit)
        algo_bytes = 8.0 * ref_cmp + 8.0 * st["hits"] + 24.0 * n_mine
        t_scan = scan_ms * 1e-3
        valu_per = valu_per_2048_cmp(st["pruned"], a.max_dist)
        useful_valu_cycles = cmp_per_launch / 2048.0 * valu_per * VALU_CYCLES_PER_INSTR
        lane_ops = cmp_per_launch * valu_per / 32.0     # 64 lanes x instructions
        traffic, traffic_src = traffic_point(a.sites, n_mine, a.dist, st["pruned"])
        trace_ms, _trace_src = traffic_point(a.sites, n_mine, a.dist, st["pruned"], key="trace_avg_launch_ms")
        regime = extras.get("north_star_10k_guides") or {}
        out = {
            "metric": "guides/sec (whole node) + achieved HBM GB/s, 20bp/<=4mm ISSL scan",
            "value": value,
            "unit": "guides/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": ms_per_step,
            "ms_per_step_without_event_pairs": default_events_ms,  # the library's default (scan_events = 2): what a caller's back-to-back batches take
            "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": f"{n_total} guides per step ({'sharded over ' + str(world) + ' GPUs in interleaved chunks of ' + str(a.chunk) if world > 1 else 'one GPU'}) "
                            f"vs {a.sites}-line ({hdr['n_sites']} distinct sites) "
                            f"{'uniform' if a.dist == 'uniform' else 'AT-rich order-3 Markov'} synthetic ISSL index, 20 bp, slice width 8, "
                            f"<= {a.max_dist} mismatches, MIT+CFD ('{a.method}', threshold {a.threshold:g}) = BASELINE configs[{2 if world == 1 else 3}]"
                            f"{'' if (a.sites, n_total) in ((300_000_000, 100_000), (300_000_000, 1_000_000)) else ' shape at other sizes'}; "
                            f"index image, guides and scores resident in HBM (host-pointer entry point: extras.host_pointer_entry)",
                "guides_per_step_total": n_total, "guides_per_gpu": n_mine, "sites": a.sites, "distribution": a.dist,
                "distinct_sites": hdr["n_sites"], "image_bytes": image_bytes, "max_dist": a.max_dist, "threshold": a.threshold,
                "method": a.method, "spinup_ms": a.spinup_ms, "shard_chunk": a.chunk if world > 1 else None,
                "parallelism": f"interleaved guide shards x{world}, replicated index (RCCL broadcast, gather of scores)" if world > 1 else "single GPU",
            },
            "roofline": {
                "bound": "valu",
                "kernel": f"k_scan<{min(a.max_dist, 4) if a.max_dist <= 4 else -1}>",
                "achieved": lane_ops / t_scan / 1e12,
                "peak": N_SIMD * 32 * PEAK_CLOCK_HZ / 1e12,
                "unit": f"TOP/s (32-bit VALU lane-ops; {valu_per:.1f} wave64 instructions per guide and 2048 candidates)",
                "valu_per_2048_comparisons": valu_per,
                "frac": useful_valu_cycles / (N_SIMD * PEAK_CLOCK_HZ * t_scan),
                # the same fraction on the other two clocks of the launch: the HIP event pair around it (measured live, like `frac`)
                # and the mean duration in the committed rocprofv3 --kernel-trace of this build (profiles/scan_traffic.json; null
                # when the trace is of another build of the kernels) -- the trace's mean includes the process's first, cold launches
                "frac_events": useful_valu_cycles / (N_SIMD * PEAK_CLOCK_HZ * st["ms_scan_events"] * 1e-3) if st["ms_scan_events"] else None,
                "frac_trace": useful_valu_cycles / (N_SIMD * PEAK_CLOCK_HZ * trace_ms * 1e-3) if trace_ms else None,
                "trace_avg_launch_ms": trace_ms,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "scan_units_per_launch": st["scan_tiles"],
                "avg_launch_ms": scan_ms,
                "avg_launch_ms_events": st["ms_scan_events"],
                "avg_launch_ms_alone": stages["ms_scan"],
                "frac_alone": useful_valu_cycles / (N_SIMD * PEAK_CLOCK_HZ * stages["ms_scan"] * 1e-3),
                "lanes": lanes,
                "pruned": st["pruned"],
                "comparisons_per_launch": cmp_per_launch,
                "planned_comparisons": st["planned_comparisons"],
                "reference_comparisons_per_launch": ref_cmp,
                "comparisons_per_s": cmp_per_launch / t_scan,
                "reference_comparisons_per_s": ref_cmp / t_scan,
                "hbm_frac": (traffic / t_scan / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "algorithmic_bytes_per_launch": algo_bytes,
                "algorithmic_GBps": algo_bytes / t_scan / 1e9,
                "algorithmic_over_hbm_peak": algo_bytes / t_scan / 1e9 / HBM_PEAK_GBS,
                # north_star's own point (10 000 guides per step, same index), where the scan is HBM-bound: flat copies of
                # extras.north_star_10k_guides so that a reader of `roofline` alone has them
                "hbm_regime_guides_per_step": regime.get("guides_per_step"),
                "hbm_regime_scan_ms": regime.get("scan_ms"),
                "hbm_regime_pmc_bytes_per_launch": regime.get("hbm_pmc_bytes_per_launch"),
                "hbm_regime_frac": regime.get("hbm_pmc_frac"),
                "kernels_sha16": kernels_sha16(),
                "hbm_regime": extras.get("north_star_10k_guides"),
                "note": "avg_launch_ms is the launch's own span (first workgroup in to last workgroup out, stamped by the kernel "
                        "on the 100 MHz constant clock), avg_launch_ms_events the HIP event pair around the launch on its "
                        "stream; they differ only when a second lane shares the chip (extras.two_lanes).  pruned != 0: every bucket is stored ordered by the byte of the next slice and a guide is compared only "
                        "with the 13 of 256 groups of its five buckets that can hold a site within 4 mismatches (pigeonhole over "
                        "the cyclic successor slice; same hits, bit-identical scores), so the kernel makes "
                        "comparisons_per_launch, not reference_comparisons_per_launch, and compares 12 of the 16 positions (a group's "
                        "successor-slice bases are known when a guide is placed in it).  frac = (comparisons made / 2048 x "
                        "valu_per_2048_comparisons VALU instructions x 2 cycles) / (1024 SIMDs x 2.4 GHz x launch time), comparisons "
                        "counted by the kernel itself; a unit now serves ~20 guides instead of ~400, so unit fetches (hbm_frac) and "
                        "the per-unit set-up share the time with the VALU work.  algorithmic_* is SURVEY 8(d)'s figure (8 B per "
                        "comparison OF THE REFERENCE, no credit for reuse or pruning) and is not a fraction of anything "
                        "physical; extras.whole_bucket_scan is the same kernel working through whole buckets (frac ~0.75); hbm_regime is the "
                        "same kernel at north_star's 10 000 guides per step, where it is HBM-bound (hbm_pmc_frac)",
            },
            "kernel_ms": {"bin": stages["ms_bin"], "scan": stages["ms_scan"], "verify": stages["ms_verify"], "group": stages["ms_group"],
                          "replay": stages["ms_replay"], "pipeline": stages["ms_total"]},
            "hits_per_step": st["hits"],
            "setup_s": timings,
            "gather_s": gather_s if use_dist else None,
            "per_rank": per_rank,
            "collectives": ({"backend": backend, "world": world,
                             "rccl_version": rccl_version(torch) if backend == "nccl" else None,
                             "image_bytes": image_bytes, "broadcast_piece_bytes": 1 << 30,
                             "broadcast_GBps": image_bytes / timings["broadcast_s"] / 1e9 if timings.get("broadcast_s") else None,
                             "broadcast_limit_s": BROADCAST_LIMIT_S} if use_dist else None),
            "extras": extras or None,
        }
        want_cli = world == 1 and not a.no_cli and not a.no_extras
        if world == 1 and (not a.no_cpu_baseline or want_cli):
            # neither leg must ever cost the line itself: on any failure it is reported inside the line
            t1 = time.perf_counter()
            issl_path = None
            try:
                import shutil
                need = 48 * hdr["n_sites"] + (1 << 20)  # .issl bytes: 8 B per site + 40 B of slice lists (+ header, tables)
                tmp = next((d for d in ("/dev/shm", "/tmp", str(ROOT / "gpurun_out"))
                            if os.path.isdir(d) and os.access(d, os.W_OK) and shutil.disk_usage(d).free > 1.2 * need), None)
                if tmp is None:
                    raise RuntimeError(f"no scratch directory with {need / 1e9:.0f} GB free for the .issl")
                issl_path = f"{tmp}/bench_{os.getpid()}.issl"
                index.write(issl_path)
                write_s = time.perf_counter() - t1
                if not a.no_cpu_baseline:
                    try:
                        gpu_scores = (out_mit[-1].cpu().numpy(), out_cfd[-1].cpu().numpy())
                        out["cpu_baseline"] = cpu_baseline(issl_path, guides, gpu_scores, a.max_dist, a.threshold, a.method, a.cpu_budget_s)
                        out["cpu_baseline"]["issl_write_s"] = write_s
                    except Exception as e:  # noqa: BLE001
                        out["cpu_baseline"] = {"value": None, "unit": "guides/s", "cores": 0, "kind": "port", "sample": "not measured",
                                               "error": f"{type(e).__name__}: {e}"}
                    log(f"[bench] cpu baseline leg took {time.perf_counter()-t1:.1f} s")
                if want_cli:
                    # The drop-in where Crackling meets it (Crackling.py:767-778, config.ini:106-112): `bin/isslScoreOfftargets
                    # <issl> <query> 4 75 and > out` as a fresh child process per page, one-shot and through the resident
                    # server, a 1 M-guide and a 10 k-guide page, stdout compared byte for byte with the in-process result.
                    t2 = time.perf_counter()
                    try:
                        sys.path.insert(0, str(ROOT / "tools"))
                        import cli_end_to_end as e2e
                        pages, want = [], {}
                        for label, n_page, seed in (("page_1m_guides", 1_000_000, 4321), ("page_10k_guides", 10_000, 4322)):
                            g = random_guides_fast(sigs, n_page, seed=seed)
                            pm, pc = index.score(g, a.max_dist, a.threshold, a.method)
                            want[label] = ca.format_scores_native(g, pm, pc, a.method)
                            pages.append((label, g))
                        cli = e2e.measure(issl_path, pages, tmp, expected=want.get, server=True, log=log)
                        cli["what"] = ("bin/isslScoreOfftargets <issl> <query> 4 75 and > out as a fresh child process per page (wall = fork to exit as "
                                       "the caller sees it), .issl and query in " + tmp + "; one_shot: the process opens, uploads and scores by itself; "
                                       "resident: ISSL_SERVER points at `--serve` (index already uploaded there); timing = the process's ISSL_TIMING line")
                        cli["seconds"] = time.perf_counter() - t2
                        extras["cli_end_to_end"] = cli
                        out["extras"] = extras
                    except Exception as e:  # noqa: BLE001
                        extras["cli_end_to_end"] = {"error": f"{type(e).__name__}: {e}"}
                        out["extras"] = extras
                    log(f"[bench] cli leg took {time.perf_counter()-t2:.1f} s")
            except Exception as e:  # noqa: BLE001
                if not a.no_cpu_baseline and "cpu_baseline" not in out:
                    out["cpu_baseline"] = {"value": None, "unit": "guides/s", "cores": 0, "kind": "port", "sample": "not measured",
                                           "error": f"{type(e).__name__}: {e}"}
                if want_cli and "cli_end_to_end" not in extras:
                    extras["cli_end_to_end"] = {"error": f"{type(e).__name__}: {e}"}
                    out["extras"] = extras
            finally:
                if issl_path and os.path.exists(issl_path):
                    os.unlink(issl_path)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
