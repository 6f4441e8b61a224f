"""Scale points: BASELINE configs[2] (100 k guides x 300 M-line index, one GPU), configs[3]'s shape on one GPU (1 M
guides over three replicas of that index) and hit lists at that size, checked against the CPU oracle.

The device-built tests run at configs[2]'s full size or not at all (they fail on an MI355X that cannot hold it and are
skipped on a smaller GPU; the size is part of the test id and is printed behind the test summary); the host-built test
keeps 20 M lines (it writes the .issl for the oracle).  The bigger points of profiles/ are the same tests with
ISSL_SCALE_SITES / ISSL_SCALE_GUIDES / ISSL_SCALE_JSON set; BASELINE configs[4] itself -- 3 G lines on ONE MI355X -- is
tests/test_scale_beyond_hbm.py (a module of its own: it wants the whole HBM).  The .issl for the
oracle goes to ISSL_SCALE_TMP (default: pytest's tmp dir; use /dev/shm for files larger than the disk)."""
import json
import os
import pathlib
import time

import numpy as np
import pytest

import crackling_amd as ca
import oracle_util as ou
from synth import (random_sites_fast, markov_sites_fast, random_guides, random_guides_fast, text_order_key, check_comparisons,
                   random_sites_device, neighbours_device, DeviceSites)


def test_fast_generator_is_sorted_and_distinct():
    sigs, occ = random_sites_fast(300_000, seed=3, threads=4)
    key = text_order_key(sigs)
    assert (np.diff(key.astype(np.int64)) > 0).all() and set(np.unique(occ)) <= {1, 2}
    assert abs(int(occ.sum()) - 300_000) < 3000 and 0.015 < (occ == 2).mean() < 0.035


def _memory_limit_bytes():
    """Smaller of the cgroup limit of this container and the machine's RAM (a pod that exceeds the former is killed)."""
    limit = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES")
    for f in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            limit = min(limit, int(open(f).read().strip()))
        except (OSError, ValueError):
            pass
    return limit


@pytest.mark.gpu
def test_scale_point_matches_oracle_on_a_sample(tmp_path):
    n_lines = int(os.environ.get("ISSL_SCALE_SITES", 20_000_000))
    n_guides = int(os.environ.get("ISSL_SCALE_GUIDES", 20_000))
    n_check = int(os.environ.get("ISSL_SCALE_CHECK", 48))
    tmp = pathlib.Path(os.environ.get("ISSL_SCALE_TMP", tmp_path))
    # host bytes per site: 12 generated (+12 while the chunks are concatenated), 48 host index, 48 for the oracle's
    # own copy of the .issl, 48 more when the file itself lives in memory (/dev/shm)
    need = n_lines * (72 + 48 + (48 if str(tmp).startswith("/dev/shm") else 0))
    if need > 0.7 * _memory_limit_bytes():
        pytest.skip(f"needs ~{need / 1e9:.0f} GB of host memory, limit is {_memory_limit_bytes() / 1e9:.0f} GB")
    t = time.time(); sigs, occ = random_sites_fast(n_lines, seed=11, threads=min(32, os.cpu_count() or 8)); t_synth = time.time() - t
    guides = random_guides(sigs, n_guides, seed=12)
    print(f"synth {t_synth:.1f}s distinct={len(sigs)}", flush=True)
    t = time.time(); ix = ca.IsslIndex.build_from_sites(sigs, occ); t_build = time.time() - t
    print(f"build {t_build:.1f}s", flush=True)
    del sigs, occ
    t = time.time(); ix.upload(0); t_upload = time.time() - t
    print(f"upload {t_upload:.1f}s image {ix.device_bytes() / 1e9:.1f} GB", flush=True)
    best = None
    for rep in range(4):
        t = time.time(); mit, cfd = ix.score(guides, 4, 75.0, "and"); wall = time.time() - t
        st = ix.stats()
        print(f"rep{rep} wall {wall * 1e3:.1f} ms scan {st['ms_scan']:.2f} ms", flush=True)
        if rep and (best is None or st["ms_scan"] < best[1]["ms_scan"]):
            best = (wall, st)
    wall, st = best
    assert st["reference_comparisons"] == ix.count_candidates(guides)
    hdr = ix.header
    path = tmp / f"scale_{os.getpid()}.issl"
    try:
        t = time.time(); ix.write(path); t_write = time.time() - t
        issl_gb = path.stat().st_size / 1e9
        print(f"write {t_write:.1f}s ({issl_gb:.1f} GB)", flush=True)
        t = time.time(); oracle = ou.OracleIndex(path); t_load = time.time() - t
        print(f"oracle load {t_load:.1f}s", flush=True)
        pick = np.linspace(0, n_guides - 1, n_check).astype(np.int64)
        t = time.time(); omit, ocfd = oracle.score(guides[pick], 4, 75.0, "and", threads=min(n_check, os.cpu_count() or 8))
        t_oracle = time.time() - t
        oracle.close()
    finally:
        path.unlink(missing_ok=True)
    assert np.array_equal(mit[pick].view(np.uint64), omit.view(np.uint64))
    assert np.array_equal(cfd[pick].view(np.uint64), ocfd.view(np.uint64))
    summary = {
        "what": f"tests/test_scale.py: {n_guides} guides vs a {n_lines}-line synthetic index on one MI355X, 'and' thr 75 "
                f"max_dist 4; {n_check} guides re-scored by the CPU oracle on the same .issl: bit-identical",
        "distinct_sites": int(hdr["n_sites"]), "image_GB": ix.device_bytes() / 1e9, "issl_GB": issl_gb,
        "synth_s": t_synth, "host_build_s": t_build, "upload_s": t_upload, "wall_ms": wall * 1e3,
        "scan_ms": st["ms_scan"], "verify_ms": st["ms_verify"], "group_ms": st["ms_group"], "replay_ms": st["ms_replay"],
        "pipeline_ms": st["ms_total"], "comparisons": st["candidates"], "hits": st["hits"],
        "scan_Tcmp_per_s": st["candidates"] / st["ms_scan"] / 1e9, "algorithmic_TBps": 8.0 * st["candidates"] / st["ms_scan"] / 1e9,
        "guides_per_s_kernels": n_guides / st["ms_total"] * 1e3,
        "oracle_sample": {"guides": n_check, "seconds": t_oracle, "threads": min(n_check, os.cpu_count() or 8),
                          "guides_per_s": n_check / t_oracle},
    }
    print(json.dumps(summary), flush=True)
    if os.environ.get("ISSL_SCALE_JSON"):
        json.dump(summary, open(os.environ["ISSL_SCALE_JSON"], "w"), indent=1)


def _neighbours(sigs, guide, max_dist, chunk=1 << 26):
    """Indices of the sites within max_dist mismatches of guide: brute force over the whole table (numpy, chunked)."""
    found = []
    g = np.uint64(guide)
    even = np.uint64(0x5555555555555555)
    for lo in range(0, len(sigs), chunk):
        x = sigs[lo:lo + chunk] ^ g
        x |= x >> np.uint64(1)
        x &= even
        hit = np.flatnonzero(np.bitwise_count(x) <= max_dist)
        if len(hit):
            found.append(hit + lo)
    return np.concatenate(found) if found else np.empty(0, dtype=np.int64)


def _free_hbm_bytes():
    import torch
    free, _total = torch.cuda.mem_get_info(0)
    return free


class ScalePoint:
    """One device-built index shared by the tests below (built once per module: synthesis dominates the cost).  The size
    is the one asked for or the point does not run: it FAILS when the box is an MI355X-class GPU (>= 40 GB of free HBM)
    that cannot hold it -- a green suite means the full size ran -- and is skipped, with the reason, on a smaller GPU."""

    def __init__(self, n_lines, n_guides, options=None, n_check=None, what="configs[2]", dist="uniform", device_synth=False):
        """device_synth: the site table never exists on the host -- drawn chunk by chunk into device tensors (the same
        sites random_sites_fast makes), the index built from them (issl_index_build_from_device_sites), the checker's
        brute force run over them with torch arithmetic."""
        from concurrent.futures import ThreadPoolExecutor
        import conftest
        self.threads = min(32, os.cpu_count() or 8)
        options = dict(options or {})
        lists_cold = options.get("host_cold") == 1
        bare = options.get("keep_lists") == 0
        # HBM per site: sorted layout 152 B, compact with the lists in host memory or without lists 52 B; + 8 B of temporaries
        # while either is built (the sort's keys and the slice list being worked on live in the image's own scan section:
        # finish_upload); + 12 B of input tensors with device_synth.  Host: 12 B/site generated (twice while the chunks are
        # concatenated) + brute-force temporaries, + 40 B/site pinned for host-resident lists -- or, with device_synth, the
        # chunks in flight
        per_site_hbm = (60 if (lists_cold or bare) else 160) + (12 if device_synth else 0)
        host_need = (0 if device_synth else n_lines * (24 + (40 if lists_cold else 0))) + 16e9
        import torch
        torch.cuda.empty_cache()
        free_hbm, limit = _free_hbm_bytes(), _memory_limit_bytes()
        if free_hbm < 40e9:
            pytest.skip(f"{what}: {free_hbm / 1e9:.0f} GB of free HBM -- not the MI355X this point is sized for")
        if host_need > 0.8 * limit or free_hbm < n_lines * per_site_hbm + 8e9:
            pytest.fail(f"{what}: {n_lines} lines need ~{host_need / 1e9:.0f} GB of host memory (limit {limit / 1e9:.0f}) and "
                        f"~{n_lines * per_site_hbm / 1e9:.0f} GB of HBM (free {free_hbm / 1e9:.0f}); the point does not shrink")
        self.n_lines, self.n_guides = n_lines, n_guides
        t = time.time()
        if device_synth:
            d_sigs, d_occ = random_sites_device(self.n_lines, seed=11, threads=self.threads)
            self.sigs, self.occ = DeviceSites(d_sigs), DeviceSites(d_occ, np.uint32)
        elif dist == "markov":  # AT-rich order-3 Markov chain: a skewed index (bench.py --dist markov)
            self.sigs, self.occ = markov_sites_fast(self.n_lines, seed=20261003)
        else:
            self.sigs, self.occ = random_sites_fast(self.n_lines, seed=11, threads=self.threads)
        self.t_synth = time.time() - t
        self.guides = random_guides_fast(self.sigs, self.n_guides, seed=12)
        # HBM low-water mark of the construction: free memory sampled every 20 ms while the build runs (the call releases the GIL)
        import threading
        n_lines_sum = int(d_occ.sum(dtype=torch.int64)) if device_synth else None
        torch.cuda.synchronize()
        self.free_before_build = _free_hbm_bytes()
        low = [self.free_before_build]
        stop = threading.Event()

        def watch():
            while not stop.is_set():
                low[0] = min(low[0], _free_hbm_bytes())
                stop.wait(0.02)
        watcher = threading.Thread(target=watch, daemon=True)
        watcher.start()
        t = time.time()
        try:
            if device_synth:
                self.ix = ca.IsslIndex.build_from_device_sites(d_sigs, d_occ, n_lines_sum, device=0, options=options)
            else:
                self.ix = ca.IsslIndex.build_on_device(self.sigs, self.occ, device=0, options=options)
        finally:
            stop.set()
            watcher.join()
        self.t_build = time.time() - t
        self.build_peak_bytes = self.free_before_build - low[0]   # image + temporaries at the high-water mark of the build
        note = (f"{what}: {self.n_lines} lines ({len(self.sigs)} distinct sites) x {self.n_guides} guides, image "
                f"{self.ix.device_bytes() / 1e9:.1f} GB in HBM + {self.ix.cold()[1] / 1e9:.1f} GB pinned, sorted={self.ix.get_option('is_sorted')} "
                f"compact={self.ix.get_option('is_compact')}")
        note += f", {self.build_peak_bytes / 1e9:.1f} GB at the high-water mark of its construction"
        conftest.SCALE_NOTES.append(note)
        print(f"scale point {note}; synth {self.t_synth:.1f}s, built on the device in {self.t_build:.1f}s", flush=True)
        # what the construction may take beyond the image it leaves behind: 8 B per site (the radix passes' second buffer)
        # + their block histograms (1 KiB per 4096 sites, twice: 0.5 B per site), flags and allocator granularity -- the
        # documented "image + 8 B/site" (README, include/issl_hip.h)
        if self.ix.get_option("is_sorted") == 1:
            assert self.build_peak_bytes <= self.ix.device_bytes() + 8.5 * len(self.sigs) + (1 << 30), \
                (self.build_peak_bytes, self.ix.device_bytes(), len(self.sigs))
        # the checker: for a sample of guides every site within 4 mismatches, by brute force over the site table; those
        # sites (+ bystanders), with their occurrences and in the same relative order, form a small index for the oracle
        self.n_check = n_check or int(os.environ.get("ISSL_SCALE_CHECK", 64))
        self.pick = np.linspace(0, self.n_guides - 1, self.n_check).astype(np.int64)
        t = time.time()
        if device_synth:
            near = [neighbours_device(self.sigs.d, g, 4) for g in self.guides[self.pick]]
        else:
            with ThreadPoolExecutor(max_workers=min(self.threads, 16)) as pool:
                near = list(pool.map(lambda g: _neighbours(self.sigs, g, 4), self.guides[self.pick]))
        self.t_brute = time.time() - t
        self.keep = np.unique(np.concatenate(near + [np.arange(0, len(self.sigs), max(1, len(self.sigs) // 5000))]))
        print(f"brute force for {self.n_check} guides: {self.t_brute:.1f}s, {sum(len(x) for x in near)} sites within 4 mismatches", flush=True)

    def oracle_on_neighbourhoods(self, tmp_path, thr, want_hits=False, method="and", max_dist=4):
        """max_dist <= 4: the neighbourhood index holds every site within 4 mismatches of the sampled guides."""
        mini = ca.IsslIndex.build_from_sites(self.sigs[self.keep], self.occ[self.keep])
        path = tmp_path / "mini.issl"
        mini.write(path)
        mini.close()
        oracle = ou.OracleIndex(path)
        out = oracle.score(self.guides[self.pick], max_dist, thr, method, want_hits=want_hits)
        oracle.close()
        return out


SCALE_SIZE = (int(os.environ.get("ISSL_SCALE_SITES", 300_000_000)), int(os.environ.get("ISSL_SCALE_GUIDES", 100_000)))


@pytest.fixture(scope="module", params=[SCALE_SIZE], ids=lambda p: f"{p[0]}lines-{p[1]}guides")
def scale(request):
    options = {"compact": 1, "host_cold": 1} if os.environ.get("ISSL_SCALE_LAYOUT") == "compact_cold" else None
    sp = ScalePoint(request.param[0], request.param[1], options=options)
    yield sp
    sp.ix.close()


def _score_and_check(scale, tmp_path, what):
    """Whole buckets (the reference's loop), then the planner's choice: a sample bit-identical to the oracle, ALL scores of
    the batch identical between the two scans, comparison counters consistent; returns the summary."""
    ix, guides = scale.ix, scale.guides
    omit, ocfd = scale.oracle_on_neighbourhoods(tmp_path, 75.0)
    assert (omit < 100).any()  # the sample does meet off-targets
    runs = {}
    for prune in (0, -1):  # every bucket of every guide scanned (the reference's loop), then the planner's choice
        ix.set_option("prune", prune)
        best = None
        for rep in range(3):
            t = time.time(); mit, cfd = ix.score(guides, 4, 75.0, "and"); wall = time.time() - t
            st = ix.stats()
            print(f"prune={prune} rep{rep} wall {wall * 1e3:.1f} ms scan {st['ms_scan']:.2f} ms pruned={st['pruned']}", flush=True)
            if rep and (best is None or st["ms_total"] < best[1]["ms_total"]):
                best = (wall, st)
        check_comparisons(ix, guides, prune)
        assert np.array_equal(mit[scale.pick].view(np.uint64), omit.view(np.uint64)), prune
        assert np.array_equal(cfd[scale.pick].view(np.uint64), ocfd.view(np.uint64)), prune
        if prune == 0:
            full_mit, full_cfd = mit, cfd
        else:  # ALL guides of the batch: the pruned scan finds what the scan of the whole buckets finds
            assert np.array_equal(mit.view(np.uint64), full_mit.view(np.uint64))
            assert np.array_equal(cfd.view(np.uint64), full_cfd.view(np.uint64))
        runs[prune] = best
    wall, st = runs[-1]
    full_st = runs[0][1]
    assert st["pruned"] == 2   # at these sizes the planner prunes
    summary = {
        "what": f"{what}: {scale.n_guides} guides vs a {scale.n_lines}-line synthetic "
                f"index built on one MI355X (issl_index_build_on_device), 'and' thr 75 max_dist 4; {scale.n_check} guides checked "
                f"bit-for-bit against the CPU oracle on the index of their brute-force neighbourhoods; all {scale.n_guides} scores "
                f"identical between the whole-bucket and the pruned scan",
        "distinct_sites": int(len(scale.sigs)), "image_GB": ix.device_bytes() / 1e9, "pinned_host_GB": ix.cold()[1] / 1e9,
        "synth_s": scale.t_synth,
        "device_build_s": scale.t_build, "wall_ms": wall * 1e3, "scan_ms": st["ms_scan"], "verify_ms": st["ms_verify"],
        "bin_ms": st["ms_bin"], "group_ms": st["ms_group"], "replay_ms": st["ms_replay"], "pipeline_ms": st["ms_total"],
        "pruned": st["pruned"], "comparisons": st["candidates"], "reference_comparisons": st["reference_comparisons"],
        "hits": st["hits"], "scan_launches": st["scan_launches"],
        "scan_Tcmp_per_s": st["candidates"] / st["ms_scan"] / 1e9,
        "algorithmic_TBps": 8.0 * st["candidates"] / st["ms_scan"] / 1e9,
        "guides_per_s_kernels": scale.n_guides / st["ms_total"] * 1e3, "brute_force_check_s": scale.t_brute,
        "cold_sections": ix.get_option("cold_sections"), "is_sorted": ix.get_option("is_sorted"), "is_compact": ix.get_option("is_compact"),
        "full_scan": {"scan_ms": full_st["ms_scan"], "pipeline_ms": full_st["ms_total"], "comparisons": full_st["candidates"],
                      "scan_Tcmp_per_s": full_st["candidates"] / full_st["ms_scan"] / 1e9,
                      "guides_per_s_kernels": scale.n_guides / full_st["ms_total"] * 1e3},
    }
    print(json.dumps(summary), flush=True)
    import conftest
    conftest.SCALE_NOTES.append(f"{what.split('::')[-1]}: {scale.n_lines} lines x {scale.n_guides} guides: {st['ms_total']:.2f} ms per batch pruned "
                                f"(bin {st['ms_bin']:.2f}, scan {st['ms_scan']:.2f}, verify {st['ms_verify']:.2f}, group {st['ms_group']:.2f}, replay "
                                f"{st['ms_replay']:.2f}; {st['hits']} hits), {full_st['ms_total']:.1f} ms with whole buckets; all {scale.n_guides} scores equal")
    return summary


def _hit_lists_match(scale, tmp_path):
    for thr in (0.0, 75.0):
        got = scale.ix.dump_hits(scale.guides[scale.pick], 4, thr, "and")
        _, _, want = scale.oracle_on_neighbourhoods(tmp_path, thr, want_hits=True)
        assert len(got) == len(want) and len(got) > scale.n_check // 2, (thr, len(got), len(want))
        assert np.array_equal(got[:, [0, 1, 4, 5]], want[:, [0, 1, 4, 5]]), thr           # guide, slice, dist, occ
        assert np.array_equal(scale.sigs[got[:, 3]], scale.sigs[scale.keep][want[:, 3]]), thr  # the same sites


@pytest.mark.gpu
def test_device_built_scale_point(scale, tmp_path):
    """BASELINE configs[2] on one MI355X: index built ON the GPU (issl_index_build_on_device: no 48 B/site host arrays),
    100 k guides scored; a sample is bit-identical to the CPU oracle on the index of its brute-force neighbourhoods --
    sites farther away contribute nothing, and the scoring order (slice, position in bucket) of the survivors is
    unchanged.  The scan's own comparison counter must equal the bucket-table arithmetic: every bucket was scanned."""
    summary = _score_and_check(scale, tmp_path, "tests/test_scale.py::test_device_built_scale_point")
    if os.environ.get("ISSL_SCALE_JSON"):
        json.dump(summary, open(os.environ["ISSL_SCALE_JSON"], "w"), indent=1)


@pytest.mark.gpu
def test_wide_sample_against_the_full_oracle_at_scale(scale, tmp_path):
    """configs[2] once more with the checker that needs no neighbourhood trick: the whole index written out as an .issl
    (14 GB, byte-compatible with the reference's), loaded by the CPU oracle, and ISSL_SCALE_WIDE (2000) guides spread
    over the batch scored by it in full -- every bucket, every candidate, the seen-bitmap, the early exit -- scores at the
    product threshold and hit lists without early exit, against the GPU's.  ~50 ms per guide and thread on the oracle."""
    import shutil
    n_wide = int(os.environ.get("ISSL_SCALE_WIDE", 2000))
    need = 48 * len(scale.sigs) + (1 << 20)
    tmp = next((d for d in (os.environ.get("ISSL_SCALE_TMP"), "/dev/shm", str(tmp_path))
                if d and os.path.isdir(d) and os.access(d, os.W_OK) and shutil.disk_usage(d).free > 1.2 * need), None)
    if tmp is None or 2.2 * need > 0.8 * _memory_limit_bytes():
        pytest.fail(f"no room for the oracle's copy of the index ({need / 1e9:.0f} GB as a file + as much in memory)")
    ix, guides = scale.ix, scale.guides
    pick = np.unique(np.linspace(0, len(guides) - 1, n_wide).astype(np.int64))
    ix.set_option("prune", -1)
    mit, cfd = ix.score(guides, 4, 75.0, "and")
    assert ix.stats()["pruned"] == 2
    hits = ix.dump_hits(guides[pick[::20]], 4, 0.0, "and")
    ix.set_option("prune", 1)
    mit5, cfd5 = ix.score(guides, 5, 75.0, "and")
    st5 = ix.stats()
    assert st5["pruned"] == 3 and st5["candidates"] < 0.40 * st5["reference_comparisons"]   # 67 of 256 groups, in whole units of 2048
    ix.set_option("prune", -1)
    path = pathlib.Path(tmp) / f"scale_wide_{os.getpid()}.issl"
    try:
        t = time.time(); ix.write(path); t_write = time.time() - t
        t = time.time(); oracle = ou.OracleIndex(path); t_load = time.time() - t
        threads = min(64, max(8, 2 * (os.cpu_count() or 8)))
        t = time.time(); omit, ocfd = oracle.score(guides[pick], 4, 75.0, "and", threads=threads); t_score = time.time() - t
        t = time.time(); _, _, ohits = oracle.score(guides[pick[::20]], 4, 0.0, "and", want_hits=True, threads=threads); t_hits = time.time() - t
        omit5, ocfd5 = oracle.score(guides[pick[::8]], 5, 75.0, "and", threads=threads)
        oracle.close()
    finally:
        path.unlink(missing_ok=True)
    import conftest
    conftest.SCALE_NOTES.append(f"configs[2] wide sample: {len(pick)} guides (+ hit lists of {len(pick[::20])}) scored in full by the CPU oracle "
                                f"on the written {need / 1e9:.1f} GB .issl: write {t_write:.1f}s, load {t_load:.1f}s, oracle {t_score:.1f}s + {t_hits:.1f}s")
    assert np.array_equal(mit[pick].view(np.uint64), omit.view(np.uint64))
    assert np.array_equal(cfd[pick].view(np.uint64), ocfd.view(np.uint64))
    assert hits.shape == ohits.shape and np.array_equal(hits, ohits)   # same index on both sides: ids and list positions too
    assert (omit < 100).sum() > len(pick) // 2
    assert np.array_equal(mit5[pick[::8]].view(np.uint64), omit5.view(np.uint64))   # max_dist 5: the pruned scan's third mode
    assert np.array_equal(cfd5[pick[::8]].view(np.uint64), ocfd5.view(np.uint64))


@pytest.mark.gpu
def test_hit_lists_at_scale(scale, tmp_path):
    """Bit-exact hit lists at configs[2]'s size: for the sampled guides, (slice, site, distance, occurrences) of every
    scored off-target in scoring order, without and with early exit, against the oracle on the neighbourhood index
    (site ids and bucket positions differ between the two indexes; the sites and their order do not)."""
    _hit_lists_match(scale, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("method,thr,max_dist", [("mit", 0.0, 4), ("cfd", 75.0, 3), ("or", 50.0, 4), ("avg", 90.0, 2),
                                                 ("and", 100.0, 0), ("xyz", 75.0, 4)])
def test_methods_thresholds_distances_at_scale(scale, tmp_path, method, thr, max_dist):
    """The other score methods, exit thresholds and compiled distance tests of the scan at configs[2]'s size: the sampled
    guides scored alone (a 64-guide batch: the HBM-bound regime of the scan) against the oracle on the neighbourhood
    index, scores and hit lists."""
    sample = scale.guides[scale.pick]
    mit, cfd = scale.ix.score(sample, max_dist, thr, method)
    omit, ocfd, ohits = scale.oracle_on_neighbourhoods(tmp_path, thr, want_hits=True, method=method, max_dist=max_dist)
    assert np.array_equal(mit.view(np.uint64), omit.view(np.uint64)), (method, thr, max_dist)
    assert np.array_equal(cfd.view(np.uint64), ocfd.view(np.uint64)), (method, thr, max_dist)
    check_comparisons(scale.ix, sample)
    hits = scale.ix.dump_hits(sample, max_dist, thr, method)
    assert len(hits) == len(ohits)
    assert np.array_equal(hits[:, [0, 1, 4, 5]], ohits[:, [0, 1, 4, 5]])
    assert np.array_equal(scale.sigs[hits[:, 3]], scale.sigs[scale.keep][ohits[:, 3]])


@pytest.mark.gpu
def test_config3_shape_on_one_gpu(scale):
    """BASELINE configs[3]'s shape (1 M guides against replicas of the 300 M-site index, results gathered in input
    order) on the one GPU of the test box: three replicas on device 0 (peer-copy path), the batch handed out as a queue
    of chunks.  The first third of the batch is a 'repeat region' (near-copies of a few sites: many hits, slow replay),
    the way Crackling's genome-ordered pages are skewed: contiguous thirds would leave replica 0 with all of it."""
    n = 1_000_000 if scale.n_lines >= 100_000_000 else 120_000
    guides = random_guides_fast(scale.sigs, n, seed=99)
    rng = np.random.default_rng(5)
    dense = scale.sigs[rng.integers(0, len(scale.sigs), size=8)]
    third = n // 3
    guides[:third] = dense[rng.integers(0, 8, size=third)] ^ (rng.integers(0, 4, size=third, dtype=np.uint64) << np.uint64(2 * 19))
    want_m, want_c = scale.ix.score(guides, 4, 75.0, "and")
    node = ca.IsslNode(scale.ix, devices=[0, 0, 0])
    try:
        assert node.info()["n_devices"] == 3
        node.score(guides, 4, 75.0, "and")                    # sizes the replicas' scratch buffers for chunks of THIS size (the
                                                              # queue's chunk grows with the batch) and lets every handle see a
                                                              # piece go through at once: replica 0 is the warm handle above,
                                                              # the other two are new
        t = time.time(); mit, cfd = node.score(guides, 4, 75.0, "and"); wall = time.time() - t
        busy, done = node.shard_times()
    finally:
        node.close()
    assert np.array_equal(mit.view(np.uint64), want_m.view(np.uint64)) and np.array_equal(cfd.view(np.uint64), want_c.view(np.uint64))
    assert sum(done) == n and min(done) > 0
    print(f"node of 3 replicas: {n} guides in {wall * 1e3:.1f} ms, busy ms per replica {[round(b, 1) for b in busy]}, guides {done}", flush=True)
    assert max(busy) / min(busy) <= 1.2, busy


@pytest.mark.gpu
def test_cold_sections_in_host_memory_at_scale(monkeypatch, tmp_path):
    """The layout for indexes larger than the HBM (sites and slice lists in pinned host memory, scan stream in HBM),
    forced at 20 M lines: same bits as the all-HBM image of the same index, device-built and from host arrays."""
    sigs, occ = random_sites_fast(20_000_000, seed=21, threads=min(32, os.cpu_count() or 8))
    guides = random_guides_fast(sigs, 20_000, seed=22)
    hot = ca.IsslIndex.build_on_device(sigs, occ, device=0)
    assert hot.get_option("cold_on_host") == 0
    want = hot.score(guides, 4, 75.0, "and")
    want_hits = hot.dump_hits(guides[:200], 4, 0.0, "and")
    hot_bytes = hot.device_bytes()
    hot.close()
    monkeypatch.setenv("ISSL_FORCE_HOST_COLD", "1")      # read when a handle is created
    t = time.time(); cold = ca.IsslIndex.build_on_device(sigs, occ, device=0); t_build = time.time() - t
    assert cold.get_option("cold_on_host") == 1 and cold.cold()[1] >= 48 * len(sigs) and cold.device_bytes() < 0.4 * hot_bytes
    for rep in range(2):
        t = time.time(); got = cold.score(guides, 4, 75.0, "and"); wall = time.time() - t
    st = cold.stats()
    print(f"host-cold image: built in {t_build:.1f}s, {cold.device_bytes() / 1e9:.2f} GB in HBM + {cold.cold()[1] / 1e9:.2f} GB pinned; "
          f"20k guides in {wall * 1e3:.1f} ms (scan {st['ms_scan']:.2f}, verify {st['ms_verify']:.2f}, replay {st['ms_replay']:.2f})", flush=True)
    assert np.array_equal(got[0].view(np.uint64), want[0].view(np.uint64)) and np.array_equal(got[1].view(np.uint64), want[1].view(np.uint64))
    assert np.array_equal(cold.dump_hits(guides[:200], 4, 0.0, "and"), want_hits)
    # written back out (the lists come from the pinned host copy) and uploaded again from that file's host arrays
    path = tmp_path / "cold.issl"
    cold.write(path)
    cold.close()
    again = ca.IsslIndex.open(path).upload(0)
    assert again.get_option("cold_on_host") == 1
    got2 = again.score(guides[:5000], 4, 75.0, "and")
    assert np.array_equal(got2[0].view(np.uint64), want[0][:5000].view(np.uint64))
    again.close()


@pytest.mark.gpu
@pytest.mark.parametrize("options", [None, {"compact": 1}, {"compact": 1, "host_cold": 1}, {"keep_lists": 0}, {"sorted_layout": 0},
                                     {"sorted_layout": 0, "inline_sigs": 0}, {"sorted_layout": 0, "host_cold": 1}],
                         ids=["sorted", "compact", "compact-lists-cold", "compact-no-lists", "list-order", "list-order-no-inline", "list-order-cold"])
def test_many_hit_guides_several_per_replay_workgroup(tmp_path, options):
    """More guides of every many-hit class than the replay kernel of that class has workgroups: 2300 guides with 2049 ..
    16384 hits (k_replay_big<256>: 2048 workgroups) and 560 with more (k_replay_big<1024>: 512), besides 2300 of
    k_replay_mid's, so that every workgroup takes a second guide after its first.  The batch is scored whole, in pieces
    small enough that no workgroup meets two guides, and -- a sample -- by the oracle on the guides' neighbourhoods."""
    rng = np.random.default_rng(20261004)
    classes = [(2300, 900, 513, 2048), (2300, 3600, 2049, 16384), (560, 24000, 16385, 1 << 30)]  # guides, draws, hits from .. to
    centres = rng.integers(0, 1 << 40, size=sum(c[0] for c in classes), dtype=np.uint64)

    def variants(c, count, k_from):
        out = np.repeat(c, count)
        k = rng.integers(k_from, 5, size=len(out))
        for j in range(4):
            pos = rng.integers(0, 20, size=len(out)).astype(np.uint64)
            sub = rng.integers(1, 4, size=len(out)).astype(np.uint64)
            out = np.where(j < k, out ^ (sub << (np.uint64(2) * pos)), out)
        return out

    parts, at = [rng.integers(0, 1 << 40, size=200000, dtype=np.uint64)], 0
    for n, draws, _, _ in classes:
        parts.append(variants(centres[at:at + n], draws, 3 if draws > 20000 else 1)); at += n
    sig = np.unique(np.concatenate(parts))
    sig = sig[np.argsort(text_order_key(sig), kind="stable")]
    occ = rng.integers(1, 3, size=len(sig)).astype(np.uint32)
    order = rng.permutation(len(centres))   # the classes mixed through the batch
    guides = centres[order]
    ix = ca.IsslIndex.build_on_device(sig, occ, device=0, options=options)
    try:
        assert ix.get_option("is_sorted") == (0 if options and options.get("sorted_layout") == 0 else 1)
        assert ix.get_option("is_compact") == (1 if options and (options.get("compact") == 1 or options.get("keep_lists") == 0) else 0)
        assert (ix.get_option("cold_on_host") == 1) == bool(options and options.get("host_cold") == 1)
        mit, cfd = ix.score(guides, 4, 0.0, "and")
        hits = ix.stats()["hits"]
        got = ix.dump_hits(guides[:512], 4, 0.0, "and")
        per_guide = np.bincount(got[:, 0], minlength=512)
        at = 0
        for n, _, lo, hi in classes:   # every guide of the sample lies in its class
            mine = np.flatnonzero((order[:512] >= at) & (order[:512] < at + n)); at += n
            assert len(mine) and (per_guide[mine] >= lo).all() and (per_guide[mine] <= hi).all(), (lo, hi, per_guide[mine].min(), per_guide[mine].max())
        for thr, method in ((0.0, "and"), (75.0, "and"), (90.0, "or"), (50.0, "mit")):
            mit, cfd = ix.score(guides, 4, thr, method)
            for rep in range(2):
                m2, c2 = ix.score(guides, 4, thr, method)
                assert np.array_equal(m2.view(np.uint64), mit.view(np.uint64)) and np.array_equal(c2.view(np.uint64), cfd.view(np.uint64)), (thr, method, rep)
            pm = np.empty(len(guides)); pc = np.empty(len(guides))
            piece = 64
            for lo in range(0, len(guides), piece):
                pm[lo:lo + piece], pc[lo:lo + piece] = ix.score(guides[lo:lo + piece], 4, thr, method)
            bad = np.flatnonzero((pm.view(np.uint64) != mit.view(np.uint64)) | (pc.view(np.uint64) != cfd.view(np.uint64)))
            assert len(bad) == 0, (thr, method, len(bad), bad[:8].tolist())
            for slots in (0, 2):   # every hit through the grouping pass (no per-guide slots) / slots for 2048 hits per guide: the same scores
                ix.set_option("hit_slots", slots)
                m0, c0 = ix.score(guides, 4, thr, method)
                assert np.array_equal(m0.view(np.uint64), mit.view(np.uint64)) and np.array_equal(c0.view(np.uint64), cfd.view(np.uint64)), (thr, method, slots)
            ix.set_option("hit_slots", 1)
            # the oracle on the neighbourhoods of a sample (brute force over the site table)
            pick = np.linspace(0, len(guides) - 1, 36).astype(np.int64)
            if thr in (0.0, 75.0):
                near = [_neighbours(sig, g, 4) for g in guides[pick]]
                keep = np.unique(np.concatenate(near + [np.arange(0, len(sig), 4000)]))
                mini = ca.IsslIndex.build_from_sites(sig[keep], occ[keep])
                path = tmp_path / "mini_many.issl"
                mini.write(path); mini.close()
                oracle = ou.OracleIndex(path)
                omit, ocfd = oracle.score(guides[pick], 4, thr, method)
                oracle.close()
                assert np.array_equal(mit[pick].view(np.uint64), omit.view(np.uint64)), (thr, method)
                assert np.array_equal(cfd[pick].view(np.uint64), ocfd.view(np.uint64)), (thr, method)
        # the same guides at the end of a batch of more than 2^18: the three-kernel prefix sum over the hit counts instead of
        # the one-workgroup one (with and without hit slots)
        filler = rng.integers(0, 1 << 40, size=(1 << 18) + 1000, dtype=np.uint64)
        long_batch = np.concatenate([filler, guides])
        want_m, want_c = ix.score(guides, 4, 75.0, "and")
        for slots in (1, 0, 2):
            ix.set_option("hit_slots", slots)
            lm, lc = ix.score(long_batch, 4, 75.0, "and")
            assert np.array_equal(lm[len(filler):].view(np.uint64), want_m.view(np.uint64)), slots
            assert np.array_equal(lc[len(filler):].view(np.uint64), want_c.view(np.uint64)), slots
        ix.set_option("hit_slots", 1)
        print(f"many-hit batch: {len(guides)} guides, {len(sig)} sites, {hits} hits", flush=True)
    finally:
        ix.close()


@pytest.mark.gpu
def test_skewed_index_at_scale(tmp_path):
    """configs[2]'s size on a skewed index (the AT-rich Markov chain of `bench.py --dist markov`: seven times the hits,
    43 % of the guides with more than 512 of them, 58 % leave through the early exit): the replay kernels for many-hit
    guides carry the load here and their workgroups take many guides each.  A sample against the oracle; the whole batch
    scored repeatedly, in pieces, with whole buckets and as back-to-back asynchronous batches on one and two lanes (what
    bench.py times) -- all bit-identical."""
    import torch
    sp = ScalePoint(SCALE_SIZE[0], SCALE_SIZE[1], what="configs[2] on a skewed index", dist="markov")
    ix, guides = sp.ix, sp.guides
    try:
        omit, ocfd = sp.oracle_on_neighbourhoods(tmp_path, 75.0)
        mit, cfd = ix.score(guides, 4, 75.0, "and")
        st = ix.stats()
        assert np.array_equal(mit[sp.pick].view(np.uint64), omit.view(np.uint64))
        assert np.array_equal(cfd[sp.pick].view(np.uint64), ocfd.view(np.uint64))
        assert st["pruned"] == 2 and st["hits"] > 300 * len(guides)
        want = (mit.view(np.uint64).copy(), cfd.view(np.uint64).copy())

        def same(m, c, what):
            bad = np.flatnonzero((m.view(np.uint64) != want[0]) | (c.view(np.uint64) != want[1]))
            assert len(bad) == 0, (what, len(bad), bad[:8].tolist())

        for rep in range(3):
            same(*ix.score(guides, 4, 75.0, "and"), f"rep {rep}")
        piece = 5000
        pm = np.empty(len(guides)); pc = np.empty(len(guides))
        for lo in range(0, len(guides), piece):
            pm[lo:lo + piece], pc[lo:lo + piece] = ix.score(guides[lo:lo + piece], 4, 75.0, "and")
        same(pm, pc, "pieces")
        ix.set_option("prune", 0)
        same(*ix.score(guides, 4, 75.0, "and"), "whole buckets")
        ix.set_option("prune", -1).set_option("hit_slots", 0)
        same(*ix.score(guides, 4, 75.0, "and"), "no hit slots")
        ix.set_option("hit_slots", 1)
        for method, thr, max_dist in (("or", 90.0, 4), ("mit", 50.0, 4), ("cfd", 75.0, 3), ("avg", 60.0, 4), ("and", 0.0, 4)):
            wm, wc = ix.score(guides, max_dist, thr, method)   # other exits, other distances: whole batch = pieces = oracle sample
            for lo in range(0, len(guides), 20000):
                qm, qc = ix.score(guides[lo:lo + 20000], max_dist, thr, method)
                assert np.array_equal(qm.view(np.uint64), wm[lo:lo + 20000].view(np.uint64)), (method, thr, lo)
                assert np.array_equal(qc.view(np.uint64), wc[lo:lo + 20000].view(np.uint64)), (method, thr, lo)
            om, oc = sp.oracle_on_neighbourhoods(tmp_path, thr, method=method, max_dist=max_dist)
            assert np.array_equal(wm[sp.pick].view(np.uint64), om.view(np.uint64)), (method, thr)
            assert np.array_equal(wc[sp.pick].view(np.uint64), oc.view(np.uint64)), (method, thr)
        d_g = torch.from_numpy(guides.view(np.int64)).cuda()
        steps = 6
        out_m = torch.empty(steps, len(guides), dtype=torch.float64, device="cuda:0"); out_c = torch.empty_like(out_m)
        for lanes in (1, 2, 3):
            ix.set_option("lanes", lanes)
            out_m.zero_(); out_c.zero_()
            while True:
                for i in range(steps):
                    ix.score_device_async(d_g, out_m[i], out_c[i], 4, 75.0, "and", stream=None)
                if ix.finish(None):
                    break
            hm, hc = out_m.cpu().numpy(), out_c.cpu().numpy()
            for i in range(steps):
                same(hm[i], hc[i], f"lanes {lanes} step {i}")
        ix.set_option("lanes", 1)
    finally:
        ix.close()


def test_device_side_site_generator_makes_the_host_generators_sites():
    """random_sites_device (what the 3 G-line point is built from) against random_sites_fast (the generator of every other
    scale point) -- the same sites and counts -- and the torch brute force against the numpy one.  CPU tensors: no GPU."""
    sigs, occ = random_sites_fast(400_000, seed=11, threads=4)
    d_sigs, d_occ = random_sites_device(400_000, seed=11, device="cpu", threads=4)
    assert np.array_equal(d_sigs.numpy().view(np.uint64), sigs) and np.array_equal(d_occ.numpy().view(np.uint32), occ)
    proxy = DeviceSites(d_sigs)
    assert len(proxy) == len(sigs) and np.array_equal(proxy[[5, 0, 399]], sigs[[5, 0, 399]])
    g = int(sigs[1234]) ^ (3 << 10) ^ (1 << 30)
    for max_dist in (0, 2, 4, 9):
        assert np.array_equal(neighbours_device(d_sigs, g, max_dist, chunk=70_000), _neighbours(sigs, np.uint64(g), max_dist, chunk=90_000))


@pytest.mark.gpu
def test_compact_image_with_lists_in_host_memory_at_scale(tmp_path):
    """The other way to keep a large index's lists out of the HBM (host_cold=1: pinned, mapped host memory, shared by
    the GPUs of a node), at 300 M lines: same checks."""
    n_lines = 300_000_000
    if n_lines * 64 + 16e9 > 0.8 * _memory_limit_bytes():
        pytest.skip(f"host memory limit {_memory_limit_bytes() / 1e9:.0f} GB: no room for the pinned slice lists of {n_lines} lines")
    sp = ScalePoint(n_lines, int(os.environ.get("ISSL_SCALE_GUIDES", 100_000)), options={"compact": 1, "host_cold": 1},
                    n_check=int(os.environ.get("ISSL_SCALE_CHECK", 32)), what="compact image, lists in host memory")
    try:
        assert sp.ix.get_option("is_compact") == 1 and sp.ix.get_option("cold_sections") == 1
        assert sp.ix.cold()[1] >= 40 * len(sp.sigs) and sp.ix.device_bytes() < 60 * len(sp.sigs) + (64 << 20)
        _score_and_check(sp, tmp_path, "tests/test_scale.py::test_compact_image_with_lists_in_host_memory_at_scale")
        _hit_lists_match(sp, tmp_path)
    finally:
        sp.ix.close()
