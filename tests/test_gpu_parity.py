"""GPU parity tests: the HIP path (through the C ABI of libissl_hip.so) against
  * the golden vectors of the compiled reference (stdout text, hit lists),
  * the CPU oracle on seeded synthetic inputs (BASELINE.json configs[0]: 1k guides x 1M sites),
  * size-independent properties at a larger size.
Integer results (hit lists, counts) must be bit-exact; MIT/CFD doubles are compared bit-exactly too
(north_star allows 1e-6: the ordered replay makes them identical in practice, tolerance stated below)."""
import os
import pathlib
import subprocess

import numpy as np
import pytest

import crackling_amd as ca
import oracle_util as ou
from synth import random_sites, random_guides, sigs_to_text, text_order_key, check_comparisons

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent
FLOAT_TOL = 1e-6  # north_star tolerance for CFD/MIT; we additionally require bit-equality below


@pytest.fixture(scope="module")
def dev_index_cache():
    cache = {}
    yield cache
    for ix in cache.values():
        ix.close()


def _dev(cache, golden):
    if golden.name not in cache:
        cache[golden.name] = ca.IsslIndex.open(golden.issl).upload(0)
    return cache[golden.name]


# The scan either works through whole buckets (what the reference's loop :344 does) or, on the sorted image and for
# max_dist <= 4, only through the successor-byte groups of a bucket that can hold a hit; by default the planner picks
# per batch (tiny indexes: whole buckets).  prune=1 forces the pruned scan, prune=0 forbids it.
PRUNE = pytest.mark.parametrize("prune", [-1, 0, 1], ids=["auto", "full", "pruned"])


@PRUNE
def test_scores_match_reference_stdout(golden, dev_index_cache, prune):
    ix = _dev(dev_index_cache, golden)
    assert ix.get_option("is_sorted") == 1
    sigs = ca.encode_guides([g.encode() for g in golden.guides])
    ix.set_option("prune", prune)
    try:
        for key, want in golden.expected.items():
            method, thr, dist = key.split("|")
            mit, cfd = ix.score(sigs, int(dist), float(thr), method)
            got = ca.format_scores(sigs, mit, cfd, method)
            assert got == want, key
            st = check_comparisons(ix, sigs, prune)
            if prune == 1 and 0 <= int(dist) <= 4:
                assert st["pruned"] == (1 if int(dist) <= 2 else 2), key
    finally:
        ix.set_option("prune", -1)


@PRUNE
def test_hit_lists_match_reference(golden, dev_index_cache, prune):
    ix = _dev(dev_index_cache, golden)
    sigs = ca.encode_guides([g.encode() for g in golden.guides])
    ix.set_option("prune", prune)
    try:
        for thr in golden.hit_thresholds():
            hits = ix.dump_hits(sigs, 4, float(thr), "and")
            want = golden.hits(thr)
            assert hits.shape == want.shape, (thr, hits.shape, want.shape)
            assert np.array_equal(hits, want), thr
    finally:
        ix.set_option("prune", -1)


def test_score_table_with_duplicate_odd_and_missing_masks(golden_oddtable):
    """A local-MIT table no builder writes (duplicate masks: the first pair wins, isslScoreOfftargets.cpp:188-197; masks
    with odd bits; patterns that are missing: 0.0, :394).  The image then carries no dense 2^20-entry table and the
    replay looks the masks up by binary search (mit_lookup) -- stdout and hit lists of the compiled reference."""
    g = golden_oddtable
    ix = ca.IsslIndex.open(g.issl).upload(0)
    assert ix.get_option("dense_mit") == 0
    sigs = ca.encode_guides([s.encode() for s in g.guides])
    for key, want in g.expected.items():
        method, thr, dist = key.split("|")
        mit, cfd = ix.score(sigs, int(dist), float(thr), method)
        assert ca.format_scores(sigs, mit, cfd, method) == want, key
    for thr in g.hit_thresholds():
        assert np.array_equal(ix.dump_hits(sigs, 4, float(thr), "and"), g.hits(thr)), thr
    ix.close()
    # the same bytes through the CLI
    exe = ROOT / "bin" / "isslScoreOfftargets"
    r = subprocess.run([str(exe), str(g.issl), str(g.guides_txt), "4", "75", "mit"], capture_output=True)
    assert r.returncode == 0 and r.stdout.decode() == g.expected["mit|75|4"]


def test_cold_sections_in_host_memory(golden, monkeypatch):
    """The layout for an index larger than the free HBM, forced on the golden indexes: scan stream and tables in HBM,
    site table and slice lists in pinned host memory read across PCIe by k_verify / k_replay.  Same stdout and hit
    lists as the reference; a node of two replicas shares the one host copy."""
    monkeypatch.setenv("ISSL_FORCE_HOST_COLD", "1")      # read when the handle is created
    ix = ca.IsslIndex.open(golden.issl)
    all_hbm = ix.device_bytes()
    ix.upload(0)
    assert ix.get_option("cold_on_host") == 1 and ix.get_option("has_inline_sigs") == 0
    host_ptr, cold_bytes = ix.cold()
    assert host_ptr and cold_bytes >= 48 * ix.header["n_sites"] and ix.device_bytes() <= all_hbm
    sigs = ca.encode_guides([g.encode() for g in golden.guides])
    for key, want in golden.expected.items():
        method, thr, dist = key.split("|")
        mit, cfd = ix.score(sigs, int(dist), float(thr), method)
        assert ca.format_scores(sigs, mit, cfd, method) == want, key
    for thr in golden.hit_thresholds():
        assert np.array_equal(ix.dump_hits(sigs, 4, float(thr), "and"), golden.hits(thr)), thr
    node = ca.IsslNode(ix, devices=[0, 0])
    mit, cfd = node.score(sigs, 4, 75.0, "and")
    assert ca.format_scores(sigs, mit, cfd, "and") == golden.expected["and|75|4"]
    node.close()
    ix.close()
    # an image with host-resident cold sections cannot be adopted without them
    monkeypatch.delenv("ISSL_FORCE_HOST_COLD")


def test_image_replication_entry_points(golden_uniform, monkeypatch):
    """issl_index_copy_image_to (how a device-built index gets into the tensor a framework broadcasts), attach of the
    copy, option errors, and the refusal to adopt a host-cold image without its cold sections."""
    import torch
    g = golden_uniform
    sigs = ca.encode_guides([s.encode() for s in g.guides])
    from synth import random_sites
    ssig, occ = random_sites(30000, seed=3)
    built = ca.IsslIndex.build_on_device(ssig, occ, device=0)
    want = built.score(sigs, 4, 75.0, "and")
    n = built.device_bytes()
    raw = torch.empty(n + 256, dtype=torch.uint8, device="cuda:0")
    off = (-raw.data_ptr()) % 256
    image = built.copy_image_to_tensor(raw[off:off + n])
    with pytest.raises(ca.IsslError):
        built.copy_image_to_tensor(raw[off:off + n - 1])          # too small
    built.close()
    twin = ca.IsslIndex.attach_tensor(image)
    got = twin.score(sigs, 4, 75.0, "and")
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert twin.get_option("cold_on_host") == 0 and twin.cold() == (None, 0)
    # options: unknown key, out of range, change while batches are in flight
    for key, value in (("nope", 1), ("scan_blocks", 0), ("item_guides", 4), ("inline_sigs", 2)):
        with pytest.raises(ca.IsslError):
            twin.set_option(key, value)
    d_g = torch.from_numpy(sigs.view(np.int64)).cuda()
    d_m = torch.empty(len(sigs), dtype=torch.float64, device="cuda:0"); d_c = torch.empty_like(d_m)
    twin.score_device_async(d_g, d_m, d_c, 4, 75.0, "and")
    with pytest.raises(ca.IsslError):
        twin.set_option("scan_blocks", 512)
    assert twin.finish()
    twin.set_option("scan_blocks", 512)
    assert twin.get_option("scan_blocks") == 512
    assert np.array_equal(twin.score(sigs, 4, 75.0, "and")[0], want[0])
    twin.close()
    # a host-cold image: the hot part alone is not an index
    monkeypatch.setenv("ISSL_FORCE_HOST_COLD", "1")
    cold = ca.IsslIndex.open(g.issl).upload(0)
    n = cold.device_bytes()
    raw2 = torch.empty(n + 256, dtype=torch.uint8, device="cuda:0")
    off2 = (-raw2.data_ptr()) % 256
    hot = cold.copy_image_to_tensor(raw2[off2:off2 + n])
    with pytest.raises(ca.IsslError) as e:
        ca.IsslIndex.attach_tensor(hot)
    assert "cold" in str(e.value)
    cold.close()


@pytest.mark.parametrize("sites", [["ACGTACGTACGTACGTACGT"], ["ACGTACGTACGTACGTACGT"] * 3 + ["TTTTTTTTTTTTTTTTTTTT"]])
def test_degenerate_indexes(sites, tmp_path):
    """One site, one site three times + another: every layout, against the oracle on the same file.  (An index without
    sites is not one: the reference scorer stops with "loading off-target sequences failed", isslScoreOfftargets.cpp
    :201-204, and the builder here refuses an empty list.)"""
    with pytest.raises(ca.IsslError):
        ca.IsslIndex.build_from_text(b"")
    text = "".join(s + "\n" for s in sorted(sites)).encode()
    path = tmp_path / "tiny.issl"
    host = ca.IsslIndex.build_from_text(text)
    host.write(path)
    assert path.read_bytes() == ou.build_issl(text)
    oracle = ou.OracleIndex(path)
    guides = ca.encode_guides(["ACGTACGTACGTACGTACGT", "ACGTACGTACGTACGTACGA", "TTTTTTTTTTTTTTTTTTTT", "GGGGGGGGGGGGGGGGGGGG"])
    for layout in ({"inline_sigs": 1, "host_cold": 0}, {"inline_sigs": 0, "host_cold": 0}, {"host_cold": 1}):
        ix = ca.IsslIndex.open(path)
        for key, value in layout.items():
            ix.set_option(key, value)
        ix.upload(0)
        for method, thr, dist in (("and", 75.0, 4), ("mit", 0.0, 0), ("cfd", 100.0, 2)):
            mit, cfd = ix.score(guides, dist, thr, method)
            omit, ocfd, ohits = oracle.score(guides, dist, thr, method, want_hits=True)
            assert np.array_equal(mit.view(np.uint64), omit.view(np.uint64)), (layout, method)
            assert np.array_equal(cfd.view(np.uint64), ocfd.view(np.uint64)), (layout, method)
            assert np.array_equal(ix.dump_hits(guides, dist, thr, method), ohits), (layout, method)
        assert ix.score(guides[:0], 4, 75.0, "and")[0].shape == (0,)
        ix.close()
    if sites:
        sig = ca.encode_guides(sorted(set(sites)))
        occ = np.array([sites.count(s) for s in sorted(set(sites))], dtype=np.uint32)
        built = ca.IsslIndex.build_on_device(sig, occ, device=0)
        mit, cfd = built.score(guides, 4, 75.0, "and")
        omit, ocfd = oracle.score(guides, 4, 75.0, "and")
        assert np.array_equal(mit.view(np.uint64), omit.view(np.uint64)) and np.array_equal(cfd.view(np.uint64), ocfd.view(np.uint64))
        built.close()
    oracle.close()


def test_cli_stdout_is_byte_identical(golden):
    exe = ROOT / "bin" / "isslScoreOfftargets"
    for key in ("and|75|4", "mit|0|4", "cfd|75|4", "xyz|0|4"):
        if key not in golden.expected:
            continue
        method, thr, dist = key.split("|")
        r = subprocess.run([str(exe), str(golden.issl), str(golden.guides_txt), dist, thr, method],
                           capture_output=True)
        assert r.returncode == 0, r.stderr.decode()
        assert r.stdout.decode() == golden.expected[key], key
    # the process leaves through _exit once its text is written, with the runtime started on a thread of its own beside the
    # host-side parsing; the tidy variants (everything released, runtime started in line) print the same
    key = "and|75|4"
    for extra in ({"ISSL_TIDY_EXIT": "1"}, {"ISSL_NO_WARMUP": "1"}, {"ISSL_TIMING": "1"}):
        r = subprocess.run([str(exe), str(golden.issl), str(golden.guides_txt), "4", "75", "and"], capture_output=True,
                           env=dict(os.environ, **extra))
        assert r.returncode == 0 and r.stdout.decode() == golden.expected[key], (extra, r.stderr.decode())
        if "ISSL_TIMING" in extra:
            import json as _json
            t = _json.loads(r.stderr.decode().strip().splitlines()[-1])
            assert {"start_ms", "open_ms", "runtime_ms", "upload_ms", "query_ms", "score_ms", "format_ms", "write_ms", "total_ms"} <= set(t)
            assert t["guides"] == len(golden.guides) and t["resident"] is False


def test_crackling_caller_roundtrip(golden_uniform):
    """The caller side of the boundary (Crackling.py:747-786) with our binary in place of the reference's."""
    targets = [g + "AGG" for g in golden_uniform.guides if set(g) <= set("ACGT")]
    text = ca.run_scorer_binary(str(ROOT / "bin" / "isslScoreOfftargets"), str(golden_uniform.issl), targets, 4, 75, "and")
    parsed = ca.parse_scorer_output(text)
    want = ca.parse_scorer_output(golden_uniform.expected["and|75|4"])
    assert parsed == {k: want[k] for k in parsed} and len(parsed) == len(set(t[:20] for t in targets))


@pytest.fixture(scope="module")
def config0(tmp_path_factory):
    """BASELINE.json configs[0]: 1k guides vs 1M-site synthetic index, slice width 8."""
    sigs, occ = random_sites(1_000_000, seed=2026)
    guides = random_guides(sigs, 1000, seed=2027)
    ix = ca.IsslIndex.build_from_sites(sigs, occ)
    p = tmp_path_factory.mktemp("cfg0") / "cfg0.issl"
    ix.write(p)
    ix.upload(0)
    oracle = ou.OracleIndex(p)
    yield ix, oracle, sigs, guides
    ix.close()
    oracle.close()


@pytest.mark.parametrize("method,thr,dist", [("and", 0.0, 4), ("and", 75.0, 4), ("or", 75.0, 4), ("avg", 50.0, 3),
                                              ("mit", 75.0, 4), ("cfd", 90.0, 2), ("and", 0.0, 0)])
@PRUNE
def test_config0_scores_match_oracle(config0, method, thr, dist, prune):
    ix, oracle, sigs, guides = config0
    ix.set_option("prune", prune)
    try:
        mit, cfd = ix.score(guides, dist, thr, method)
    finally:
        ix.set_option("prune", -1)
    omit, ocfd = oracle.score(guides, dist, thr, method)
    assert np.allclose(mit, omit, rtol=0, atol=FLOAT_TOL) and np.allclose(cfd, ocfd, rtol=0, atol=FLOAT_TOL)
    assert np.array_equal(mit.view(np.uint64), omit.view(np.uint64)), "MIT not bit-identical"
    assert np.array_equal(cfd.view(np.uint64), ocfd.view(np.uint64)), "CFD not bit-identical"
    check_comparisons(ix, guides, prune)


@PRUNE
def test_config0_hit_lists_match_oracle(config0, prune):
    ix, oracle, sigs, guides = config0
    ix.set_option("prune", prune)
    try:
        for dist, thr in ((4, 0.0), (4, 75.0), (3, 0.0), (2, 0.0), (1, 0.0), (0, 0.0)):
            hits = ix.dump_hits(guides, dist, thr, "and")
            _, _, ohits = oracle.score(guides, dist, thr, "and", want_hits=True)
            assert np.array_equal(hits, ohits), (dist, thr)
            if thr == 0.0:
                assert ix.stats()["hits"] == len(ohits)
    finally:
        ix.set_option("prune", -1)


def test_config0_properties(config0):
    ix, oracle, sigs, guides = config0
    mit, cfd = ix.score(guides, 4, 75.0, "and")
    # permutation of the batch permutes the result
    perm = np.random.default_rng(1).permutation(len(guides))
    pm, pc = ix.score(guides[perm], 4, 75.0, "and")
    assert np.array_equal(pm, mit[perm]) and np.array_equal(pc, cfd[perm])
    # splitting the batch does not change any score (guides are independent, :316-509)
    a = ix.score(guides[:333], 4, 75.0, "and"); b = ix.score(guides[333:], 4, 75.0, "and")
    assert np.array_equal(np.concatenate([a[0], b[0]]), mit) and np.array_equal(np.concatenate([a[1], b[1]]), cfd)
    # duplicates of one guide all get its score
    rep = np.repeat(guides[:7], 300)
    rm, rc = ix.score(rep, 4, 75.0, "and")
    assert np.array_equal(rm, np.repeat(mit[:7], 300)) and np.array_equal(rc, np.repeat(cfd[:7], 300))
    # an exact site scores at most 100*100/(100+occ... ) : dist-0 hit adds occ to CFD only
    site = sigs[:50]
    sm, sc = ix.score(site, 0, 0.0, "and")
    assert np.all(sm == 100.0) and np.all(sc < 100.0)
    # scores lie in (0, 100]
    assert np.all((mit > 0) & (mit <= 100)) and np.all((cfd > 0) & (cfd <= 100))
    # empty batch
    em, ec = ix.score(np.empty(0, dtype=np.uint64))
    assert len(em) == 0 and len(ec) == 0


def test_large_max_dist_and_single_guide(config0):
    ix, oracle, sigs, guides = config0
    for dist in (6, 20):
        mit, cfd = ix.score(guides[:64], dist, 0.0, "and")
        omit, ocfd = oracle.score(guides[:64], dist, 0.0, "and")
        assert np.array_equal(mit.view(np.uint64), omit.view(np.uint64)), dist
        assert np.array_equal(cfd.view(np.uint64), ocfd.view(np.uint64)), dist
    mit, cfd = ix.score(guides[5:6], 4, 75.0, "or")
    omit, ocfd = oracle.score(guides[5:6], 4, 75.0, "or")
    assert mit[0] == omit[0] and cfd[0] == ocfd[0]


def test_max_dist_5_takes_the_pruned_scan_too(config0):
    """max_dist 5 (the reference takes any, isslScoreOfftargets.cpp:109,382): a hit the reference can find has an exactly
    matching slice, and one of those is followed by a slice with at most two mismatches -- 67 of a bucket's 256
    successor-byte groups, three classes of guides with budgets 5, 4, 3 on the other 12 positions (pruned == 3).  Scores and
    hit lists against the oracle, pruned = whole buckets = planner's choice; the local MIT table has no masks of five
    mismatches: such hits add 0.0 (operator[], :394)."""
    ix, oracle, sigs, guides = config0
    try:
        got = {}
        for prune in (1, 0, -1):
            ix.set_option("prune", prune)
            for thr in (0.0, 75.0):
                mit, cfd = ix.score(guides, 5, thr, "and")
                st = check_comparisons(ix, guides, prune)
                if prune == 1:
                    assert st["pruned"] == 3   # (forced: on a 1 M-site index the planner would not prune)
                omit, ocfd = oracle.score(guides, 5, thr, "and")
                assert np.array_equal(mit.view(np.uint64), omit.view(np.uint64)), (prune, thr)
                assert np.array_equal(cfd.view(np.uint64), ocfd.view(np.uint64)), (prune, thr)
            hits = ix.dump_hits(guides[:300], 5, 0.0, "and")
            _, _, ohits = oracle.score(guides[:300], 5, 0.0, "and", want_hits=True)
            assert np.array_equal(hits, ohits), prune
            got[prune] = hits
        assert (got[1][:, 4] == 5).sum() > 100   # hits at distance five are there
        ix.set_option("prune", 1)
        for method, thr in (("mit", 50.0), ("cfd", 90.0), ("or", 75.0), ("avg", 60.0)):
            mit, cfd = ix.score(guides, 5, thr, method)
            omit, ocfd = oracle.score(guides, 5, thr, method)
            assert np.array_equal(mit.view(np.uint64), omit.view(np.uint64)) and np.array_equal(cfd.view(np.uint64), ocfd.view(np.uint64)), method
    finally:
        ix.set_option("prune", -1)


def test_runtime_threshold_build_of_the_scan_kernel(config0):
    """The scan kernel is compiled with max_dist 0..4 as constants and once with a runtime threshold (used for
    max_dist > 4).  Force the runtime-threshold build for small distances too and compare."""
    ix, oracle, sigs, guides = config0
    ix.set_option("scan_generic", 1)
    try:
        for dist in (0, 1, 2, 3, 4, 5, 16):
            mit, cfd = ix.score(guides[:256], dist, 0.0, "and")
            omit, ocfd = oracle.score(guides[:256], dist, 0.0, "and")
            assert np.array_equal(mit.view(np.uint64), omit.view(np.uint64)), dist
            assert np.array_equal(cfd.view(np.uint64), ocfd.view(np.uint64)), dist
    finally:
        ix.set_option("scan_generic", 0)
    assert ix.get_option("scan_generic") == 0
    for dist in (0, 1, 2, 3, 4):
        mit, cfd = ix.score(guides[:256], dist, 0.0, "and")
        check_comparisons(ix, guides[:256])
        omit, ocfd = oracle.score(guides[:256], dist, 0.0, "and")
        assert np.array_equal(mit.view(np.uint64), omit.view(np.uint64)), dist
        assert np.array_equal(cfd.view(np.uint64), ocfd.view(np.uint64)), dist


def test_scheduling_knobs_do_not_change_results(config0):
    """Smaller scan items (more, finer work units; tiles re-read per item) and other launch sizes: same hit lists and
    scores, and the scan counts exactly the comparisons the bucket table predicts."""
    ix, oracle, sigs, guides = config0
    rng = np.random.default_rng(5)
    batch = np.concatenate([guides, guides[:1].repeat(700) ^ (rng.integers(0, 1 << 14, size=700, dtype=np.uint64) << np.uint64(20))])
    want = ix.dump_hits(batch, 4, 0.0, "and")
    wm, wc = ix.score(batch, 4, 75.0, "and")
    expected = ix.count_candidates(batch)
    try:
        for sched, blocks, prune in (("64", 1024, 0), ("8", 1024, 0), ("200", 77, 0), ("512", 4096, 0),
                                     ("64", 1024, 1), ("8", 77, 1), ("200", 4096, 1)):
            ix.set_option("item_guides", sched).set_option("scan_blocks", blocks).set_option("prune", prune)
            got = ix.dump_hits(batch, 4, 0.0, "and")
            assert np.array_equal(got, want), (sched, blocks, prune)
            gm, gc = ix.score(batch, 4, 75.0, "and")
            assert np.array_equal(gm, wm) and np.array_equal(gc, wc), (sched, blocks, prune)
            st = check_comparisons(ix, batch, prune)
            assert st["reference_comparisons"] == expected and st["pruned"] == 2 * prune, (sched, blocks, prune)
        ix.set_option("item_guides", 512).set_option("scan_blocks", 1024)
        for slots, prune in ((0, 0), (0, 1), (1, 1), (2, 1), (2, 0)):   # without / with the per-guide hit slots (Workspace::slot_hits), 2: the wide ones
            ix.set_option("hit_slots", slots).set_option("prune", prune)
            for thr in (75.0, 0.0):
                gm, gc = ix.score(batch, 4, thr, "and")
                om, oc = oracle.score(batch, 4, thr, "and")
                assert np.array_equal(gm.view(np.uint64), om.view(np.uint64)) and np.array_equal(gc.view(np.uint64), oc.view(np.uint64)), (slots, prune, thr)
    finally:
        ix.set_option("item_guides", 512).set_option("scan_blocks", 1024).set_option("prune", -1).set_option("hit_slots", 1)
    _, _, ohits = oracle.score(batch, 4, 0.0, "and", want_hits=True)
    assert np.array_equal(want, ohits)
    with pytest.raises(ca.IsslError):
        ix.set_option("scan_blocks", 0)
    with pytest.raises(ca.IsslError):
        ix.set_option("no_such_knob", 1)


def test_skewed_batch_and_poly_a(config0):
    """Ragged work: many guides in one bucket, homopolymers, guide words equal to the zero padding."""
    ix, oracle, sigs, guides = config0
    rng = np.random.default_rng(9)
    skew = guides[:1].repeat(1500) ^ (rng.integers(0, 1 << 16, size=1500, dtype=np.uint64) << np.uint64(24))
    batch = np.concatenate([skew, np.array([0, (1 << 40) - 1, 0x5555555555, 0xAAAAAAAAAA], dtype=np.uint64)])
    mit, cfd = ix.score(batch, 4, 0.0, "and")
    omit, ocfd = oracle.score(batch, 4, 0.0, "and")
    assert np.array_equal(mit.view(np.uint64), omit.view(np.uint64))
    assert np.array_equal(cfd.view(np.uint64), ocfd.view(np.uint64))


def test_image_attach_and_caller_owned_memory(golden_uniform):
    """issl_index_upload_into / issl_index_attach_image: the path a non-root rank takes after the broadcast."""
    import torch
    src = ca.IsslIndex.open(golden_uniform.issl)
    nbytes = src.device_bytes()
    buf = torch.empty(nbytes + 256, dtype=torch.uint8, device="cuda:0")
    off = (-buf.data_ptr()) % 256
    img = buf[off:off + nbytes]
    src.upload_into_tensor(img)
    clone = img.clone()  # stands in for the broadcast copy on another rank
    if clone.data_ptr() % 256:
        pytest.skip("allocator returned unaligned clone")
    att = ca.IsslIndex.attach_tensor(clone)
    assert att.header == src.header and np.array_equal(att.bucket_sizes(), src.bucket_sizes())
    sigs = ca.encode_guides([g.encode() for g in golden_uniform.guides])
    a = src.score(sigs, 4, 75.0, "and"); b = att.score(sigs, 4, 75.0, "and")
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert ca.format_scores(sigs, b[0], b[1], "and") == golden_uniform.expected["and|75|4"]
    # device-resident in/out
    d_g = torch.from_numpy(sigs.view(np.int64)).cuda()
    d_m = torch.empty(len(sigs), dtype=torch.float64, device="cuda:0"); d_c = torch.empty_like(d_m)
    att.score_device(d_g, d_m, d_c, 4, 75.0, "and")
    torch.cuda.synchronize()
    assert np.array_equal(d_m.cpu().numpy(), a[0]) and np.array_equal(d_c.cpu().numpy(), a[1])
    att.close(); src.close()


def test_large_batch_paths(config0, monkeypatch):
    """300k guides in one launch (three-kernel prefix sum, many items per bucket) and the host entry point's
    splitting of batches beyond 2^22 guides (checked with a small piece count through repetition)."""
    ix, oracle, sigs, guides = config0
    big = np.tile(guides, 300)  # 300k guides, every bucket ~1200 guides deep
    mit, cfd = ix.score(big, 4, 75.0, "and")
    assert ix.stats()["reference_comparisons"] == 300 * ix.count_candidates(guides)
    base = ix.score(guides, 4, 75.0, "and")
    assert np.array_equal(mit, np.tile(base[0], 300)) and np.array_equal(cfd, np.tile(base[1], 300))
    ix.set_option("prune", 1)  # the same 300k guides through the successor-byte groups: 19.5 M guide slots
    mit, cfd = ix.score(big, 4, 75.0, "and")
    assert ix.stats()["pruned"] == 2
    assert np.array_equal(mit, np.tile(base[0], 300)) and np.array_equal(cfd, np.tile(base[1], 300))
    ix.set_option("prune", -1)
    huge = np.tile(guides, 4300)  # 4.3M guides: equal pieces of at most 512 k guides (what gets hit slots) ...
    mit, cfd = ix.score(huge, 4, 75.0, "and")
    assert np.array_equal(mit, np.tile(base[0], 4300)) and np.array_equal(cfd, np.tile(base[1], 4300))
    assert ix.stats()["n_batches"] == 9 and ix.stats()["n_guides"] == len(huge)
    assert ix.stats()["reference_comparisons"] == 4300 * ix.count_candidates(guides)
    ix.set_option("hit_slots", 0)  # ... without them: of 2^20 guides while the pruned scan may be chosen, else of 2^22
    mit, cfd = ix.score(huge, 4, 75.0, "and")
    assert np.array_equal(mit, np.tile(base[0], 4300)) and np.array_equal(cfd, np.tile(base[1], 4300))
    assert ix.stats()["n_batches"] == 5 and ix.stats()["n_guides"] == len(huge)
    ix.set_option("prune", 0)
    mit, cfd = ix.score(huge, 4, 75.0, "and")
    assert np.array_equal(mit, np.tile(base[0], 4300)) and np.array_equal(cfd, np.tile(base[1], 4300))
    assert ix.stats()["n_batches"] == 2 and ix.stats()["n_guides"] == len(huge)
    ix.set_option("hit_slots", 1)


def test_one_site_index(tmp_path):
    """Smallest possible index: one site, 1279 empty buckets."""
    sig = np.array([0x123456789A], dtype=np.uint64)
    ix = ca.IsslIndex.build_from_sites(sig, np.array([7], dtype=np.uint32))
    p = tmp_path / "one.issl"; ix.write(p); ix.upload(0)
    oracle = ou.OracleIndex(p)
    guides = np.array([0x123456789A, 0x123456789B, 0x0, (1 << 40) - 1], dtype=np.uint64)
    for thr in (0.0, 75.0):
        mit, cfd = ix.score(guides, 4, thr, "and")
        omit, ocfd = oracle.score(guides, 4, thr, "and")
        assert np.array_equal(mit.view(np.uint64), omit.view(np.uint64)) and np.array_equal(cfd.view(np.uint64), ocfd.view(np.uint64))
    hits = ix.dump_hits(guides, 4, 0.0, "and")
    assert len(hits) == 2 and list(hits[0]) == [0, 0, 0, 0, 0, 7] and hits[1][4] == 1
    ix.close()


@pytest.mark.parametrize("lanes", [1, 2, 3])
def test_async_batches(config0, lanes):
    """issl_score_device_async / issl_score_wait / issl_score_finish: several batches in flight on the internal stream --
    or, with the lanes option = 2 / 3, alternating between two workspaces and streams (the batches then overlap: whole tails
    beside the next scan / only the binning of a batch beside the batch before it)."""
    import torch
    ix, oracle, sigs, guides = config0
    ix.set_option("lanes", lanes)
    try:
        _async_batches(ix, oracle, guides, torch)
        if lanes >= 2:   # many batches in flight on both lanes, one of them larger than anything before (workspace growth)
            rng = np.random.default_rng(9)
            big = np.concatenate([guides, guides ^ (rng.integers(0, 4, size=len(guides), dtype=np.uint64) << np.uint64(10))])
            batches = [big, guides[:77], guides[100:900], guides[:1], big[::-1].copy(), guides[5:505]] * 3
            outs = []
            for g in batches:
                d_g = torch.from_numpy(g.view(np.int64)).cuda()
                d_m = torch.empty(len(g), dtype=torch.float64, device="cuda:0"); d_c = torch.empty_like(d_m)
                outs.append((g, d_g, d_m, d_c))
            while True:
                for g, d_g, d_m, d_c in outs:
                    ix.score_device_async(d_g, d_m, d_c, 4, 75.0, "and")
                if ix.finish():
                    break
            assert ix.stats()["n_batches"] == len(batches)
            want = {}
            for g, d_g, d_m, d_c in outs:
                key = g.tobytes()
                if key not in want:
                    want[key] = oracle.score(g, 4, 75.0, "and")
                assert np.array_equal(d_m.cpu().numpy().view(np.uint64), want[key][0].view(np.uint64))
                assert np.array_equal(d_c.cpu().numpy().view(np.uint64), want[key][1].view(np.uint64))
    finally:
        ix.set_option("lanes", 1)


def _async_batches(ix, oracle, guides, torch):
    stream = torch.cuda.current_stream().cuda_stream
    parts = [guides[:300], guides[300:301], guides[301:]]
    outs = []
    for g in parts:
        d_g = torch.from_numpy(g.view(np.int64)).cuda()
        d_m = torch.empty(len(g), dtype=torch.float64, device="cuda:0"); d_c = torch.empty_like(d_m)
        ix.score_device_async(d_g, d_m, d_c, 4, 75.0, "and", stream=stream)
        outs.append((d_g, d_m, d_c))
    assert ix.finish(stream)
    assert ix.stats()["n_batches"] == 3
    # batches without an input dependency, consumer stream waits through issl_score_wait
    outs2 = []
    for g, (d_g, _, _) in zip(parts, outs):
        d_m = torch.zeros(len(g), dtype=torch.float64, device="cuda:0"); d_c = torch.zeros_like(d_m)
        ix.score_device_async(d_g, d_m, d_c, 4, 75.0, "and", stream=None)
        outs2.append((d_m, d_c))
    ix.wait(stream)
    copies = [(m.clone(), c.clone()) for m, c in outs2]  # on the torch stream, behind the wait
    assert ix.finish(stream)
    for (m, c), (_, m0, c0) in zip(copies, outs):
        assert torch.equal(m, m0) and torch.equal(c, c0)
    mit = np.concatenate([o[1].cpu().numpy() for o in outs]); cfd = np.concatenate([o[2].cpu().numpy() for o in outs])
    omit, ocfd = oracle.score(guides, 4, 75.0, "and")
    assert np.array_equal(mit.view(np.uint64), omit.view(np.uint64)) and np.array_equal(cfd.view(np.uint64), ocfd.view(np.uint64))
    # a batch that keeps the device busy for milliseconds: what the consumer stream copies behind issl_score_wait must be the
    # scores, not what the buffers held before (on one lane the end of a batch is recorded by the wait itself)
    long_g = torch.from_numpy(np.tile(guides, 200).view(np.int64)).cuda()
    d_m = torch.zeros(len(long_g), dtype=torch.float64, device="cuda:0"); d_c = torch.zeros_like(d_m)
    while True:   # (once through first: a batch of this size may have to grow the record buffers)
        ix.score_device_async(long_g, d_m, d_c, 4, 75.0, "and", stream=None)
        if ix.finish(stream):
            break
    for _ in range(4):   # (two lanes: the other workspace may still have to grow)
        d_m.zero_(); d_c.zero_()
        torch.cuda.synchronize()
        ix.score_device_async(long_g, d_m, d_c, 4, 75.0, "and", stream=None)
        ix.wait(stream)
        m1, c1 = d_m.clone(), d_c.clone()
        if ix.finish(stream):
            break
    else:
        raise AssertionError("the batch kept asking for larger buffers")
    assert torch.equal(m1, d_m) and torch.equal(c1, d_c)
    assert np.array_equal(m1.cpu().numpy().view(np.uint64), np.tile(omit, 200).view(np.uint64))


def test_guides_with_thousands_of_hits(tmp_path):
    """Dense neighbourhoods: guides with ~6000 and ~1500 scored off-targets (replay sorts in HBM, beyond the 512-key
    LDS buffer), one with ~300 (LDS sort) and one with a few dozen (rank path), each with and without early exit."""
    rng = np.random.default_rng(77)
    centres = rng.integers(0, 1 << 40, size=3, dtype=np.uint64)

    def neighbours(c, count, max_sub):
        out = set()
        while len(out) < count:
            s = int(c)
            for p in rng.choice(20, size=int(rng.integers(0, max_sub + 1)), replace=False):
                s ^= int(rng.integers(1, 4)) << (2 * int(p))
            out.add(s)
        return out

    sites = neighbours(centres[0], 6000, 4) | neighbours(centres[1], 1500, 3) | neighbours(centres[2], 40, 2)
    centres = np.append(centres, rng.integers(0, 1 << 40, dtype=np.uint64))
    sites |= neighbours(centres[3], 300, 3)  # 64 < hits <= 512: the LDS sort
    # one guide whose hits almost all sit in ONE slice list (substitutions outside the first four bases only):
    # more than 7680 in slice 0, the length beyond which k_replay_big sorts in HBM instead of LDS
    centres = np.append(centres, rng.integers(0, 1 << 40, dtype=np.uint64))
    wide = set()
    while len(wide) < 9000:
        s = int(centres[4])
        for p in rng.choice(16, size=int(rng.integers(1, 5)), replace=False):
            s ^= int(rng.integers(1, 4)) << (2 * (4 + int(p)))
        wide.add(s)
    sites |= wide
    # ... and one with more than 16384 hits, 20 000 of them in slice 0: the 1024-thread build of k_replay_big, whose LDS
    # takes 7680 hits of a slice -- the slice is walked in runs of id groups that fit (without early exit: all of them)
    centres = np.append(centres, rng.integers(0, 1 << 40, dtype=np.uint64))
    wider = set()
    while len(wider) < 20000:
        s = int(centres[5])
        for p in rng.choice(16, size=int(rng.integers(1, 5)), replace=False):
            s ^= int(rng.integers(1, 4)) << (2 * (4 + int(p)))
        wider.add(s)
    sites |= wider
    sites |= set(int(x) for x in rng.integers(0, 1 << 40, size=20000, dtype=np.uint64))
    sig = np.array(sorted(sites), dtype=np.uint64)
    sig = sig[np.argsort(text_order_key(sig), kind="stable")]
    occ = rng.integers(1, 4, size=len(sig)).astype(np.uint32)
    ix = ca.IsslIndex.build_from_sites(sig, occ)
    p = tmp_path / "dense.issl"
    ix.write(p)
    ix.upload(0)
    oracle = ou.OracleIndex(p)
    guides = np.concatenate([centres, centres ^ np.uint64(3)])
    for thr in (0.0, 75.0, 99.0):
        for method in ("and", "or", "mit", "cfd"):
            mit, cfd = ix.score(guides, 4, thr, method)
            omit, ocfd = oracle.score(guides, 4, thr, method)
            assert np.array_equal(mit.view(np.uint64), omit.view(np.uint64)), (thr, method)
            assert np.array_equal(cfd.view(np.uint64), ocfd.view(np.uint64)), (thr, method)
    hits = ix.dump_hits(guides, 4, 0.0, "and")
    _, _, ohits = oracle.score(guides, 4, 0.0, "and", want_hits=True)
    assert np.array_equal(hits, ohits)
    per_guide = np.bincount(hits[:, 0], minlength=len(guides))
    assert per_guide.max() > 2048 and per_guide[1] > 512 and 64 < per_guide[3] <= 512 and per_guide[2] <= 64
    assert np.count_nonzero((hits[:, 0] == 4) & (hits[:, 1] == 0)) > 7680  # a slice beyond the LDS of either build
    assert per_guide[5] > 16384 and np.count_nonzero((hits[:, 0] == 5) & (hits[:, 1] == 0)) > 2 * 7680
    for thr in (50.0, 75.0, 99.0):   # ... and the hits scored before the early exit (the head of such a slice is tried first)
        _, _, ohits = oracle.score(guides, 4, thr, "and", want_hits=True)
        assert np.array_equal(ix.dump_hits(guides, 4, thr, "and"), ohits), thr
    ix.close()


def test_replay_workgroups_that_take_several_mid_size_guides(tmp_path):
    """More guides with 513..2048 hits than k_replay_mid has workgroups (2048), so that a workgroup takes a second
    guide after its first -- and the first ones are guides it hands on to k_replay_big (one slice list longer than the
    1024 hits it ranks in LDS): what follows such a guide in the same workgroup must not see its leftovers."""
    rng = np.random.default_rng(4242)
    n_hand, n_mid = 128, 2100
    centres = rng.integers(0, 1 << 40, size=n_hand + n_mid, dtype=np.uint64)

    def variants(c, count, first_pos):
        """`count` signatures with 1..4 substitutions of `c` at positions first_pos..19 (per centre, vectorised)."""
        out = np.repeat(c, count)
        k = rng.integers(1, 5, size=len(out))
        for j in range(4):
            pos = rng.integers(first_pos, 20, size=len(out)).astype(np.uint64)
            sub = rng.integers(1, 4, size=len(out)).astype(np.uint64)
            out = np.where(j < k, out ^ (sub << (np.uint64(2) * pos)), out)
        return out

    sig = np.concatenate([variants(centres[:n_hand], 1800, 4),   # slice 0 of the guide matches exactly: one long list
                          variants(centres[:n_hand], 100, 0),
                          variants(centres[n_hand:], 800, 0),
                          rng.integers(0, 1 << 40, size=50000, dtype=np.uint64)])
    sig = np.unique(sig)
    sig = sig[np.argsort(text_order_key(sig), kind="stable")]
    occ = rng.integers(1, 3, size=len(sig)).astype(np.uint32)
    ix = ca.IsslIndex.build_from_sites(sig, occ)
    p = tmp_path / "mid.issl"
    ix.write(p)
    ix.upload(0)
    oracle = ou.OracleIndex(p)
    guides = centres
    hits = ix.dump_hits(guides, 4, 0.0, "and")
    per_guide = np.bincount(hits[:, 0], minlength=len(guides))
    in_slice0 = np.bincount(hits[hits[:, 1] == 0, 0], minlength=len(guides))
    assert (per_guide > 512).all() and (per_guide <= 2048).all()          # every guide is k_replay_mid's to begin with
    assert (in_slice0[:n_hand] > 1024).all() and (in_slice0[n_hand:] <= 1024).all()
    assert len(guides) > 2048 + n_hand                                      # guides 2048.. follow a handed-on one
    for thr in (75.0, 0.0):
        for rep in range(2):
            mit, cfd = ix.score(guides, 4, thr, "and")
            omit, ocfd = oracle.score(guides, 4, thr, "and")
            bad = np.flatnonzero((mit.view(np.uint64) != omit.view(np.uint64)) | (cfd.view(np.uint64) != ocfd.view(np.uint64)))
            assert len(bad) == 0, (thr, rep, bad[:10].tolist())
    ix.close()


def test_node_sharding_and_rccl_broadcast(golden_uniform, monkeypatch):
    """In-process multi-GPU orchestration on the one GPU of the test box: (a) two replicas on device 0 (peer-copy
    path: RCCL refuses a device listed twice) exercise sharding, host threads and the gather into the caller's
    arrays; (b) a one-device node with ISSL_FORCE_RCCL=1 runs ncclCommInitAll + ncclBroadcast for real."""
    sigs = ca.encode_guides([g.encode() for g in golden_uniform.guides])
    want = golden_uniform.expected["and|75|4"]
    ix = ca.IsslIndex.open(golden_uniform.issl)
    node = ca.IsslNode(ix, devices=[0, 0, 0])
    inf = node.info()
    assert inf["n_devices"] == 3 and inf["used_rccl"] == 0
    mit, cfd = node.score(sigs[:101], 4, 75.0, "and")  # ragged shards 34/34/33
    assert ca.format_scores(sigs[:101], mit, cfd, "and") == "".join(want.splitlines(True)[:101])
    mit, cfd = node.score(sigs, 4, 75.0, "and")
    assert ca.format_scores(sigs, mit, cfd, "and") == want
    mit, cfd = node.score(sigs[:2], 4, 75.0, "and")    # fewer guides than devices
    assert ca.format_scores(sigs[:2], mit, cfd, "and") == "".join(want.splitlines(True)[:2])
    node.close(); ix.close()
    monkeypatch.setenv("ISSL_FORCE_RCCL", "1")
    ix = ca.IsslIndex.open(golden_uniform.issl)
    node = ca.IsslNode(ix, devices=[0])
    assert node.info()["used_rccl"] == 1, "RCCL could not be loaded or the broadcast failed"
    mit, cfd = node.score(sigs, 4, 75.0, "and")
    assert ca.format_scores(sigs, mit, cfd, "and") == want
    node.close(); ix.close()


def test_cli_multi_device_env(golden_uniform):
    exe = ROOT / "bin" / "isslScoreOfftargets"
    env = dict(os.environ, ISSL_DEVICES="0,0", ISSL_TIMING="1")
    r = subprocess.run([str(exe), str(golden_uniform.issl), str(golden_uniform.guides_txt), "4", "75", "and"],
                       capture_output=True, env=env)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout.decode() == golden_uniform.expected["and|75|4"]
    assert b'"devices": 2' in r.stderr


def test_resident_server_mode(golden, tmp_path):
    """`isslScoreOfftargets --serve <socket>` keeps the index in HBM; invocations with ISSL_SERVER set get the same
    bytes from it (second call: resident hit); an unreachable server falls back to in-process scoring."""
    import time
    exe = str(ROOT / "bin" / "isslScoreOfftargets")
    sock = str(tmp_path / "issl.sock")
    server = subprocess.Popen([exe, "--serve", sock], stderr=subprocess.PIPE)
    try:
        for _ in range(100):
            if os.path.exists(sock):
                break
            time.sleep(0.05)
        assert os.path.exists(sock)
        env = dict(os.environ, ISSL_SERVER=sock, ISSL_TIMING="1")
        keys = [k for k in ("and|75|4", "mit|0|4", "xyz|0|4") if k in golden.expected]
        for i, key in enumerate(keys + keys[:1]):
            method, thr, dist = key.split("|")
            r = subprocess.run([exe, str(golden.issl), str(golden.guides_txt), dist, thr, method], capture_output=True,
                               env=env, cwd=str(tmp_path))
            assert r.returncode == 0, r.stderr.decode()
            assert r.stdout.decode() == golden.expected[key], key
            assert (b'"resident": true' in r.stderr) == (i > 0), r.stderr
        # stdout redirected into a file, as Crackling does it (Crackling.py:767): the descriptor travels to the server,
        # which writes the text itself (head line "OKFD"); appended behind what the file holds; ISSL_SERVER_NO_FD: through
        # the socket as for a pipe -- the same bytes every way
        out_file = tmp_path / "redirected.out"
        for mode, extra in (("wb", {}), ("ab", {}), ("wb", {"ISSL_SERVER_NO_FD": "1"})):
            before = out_file.read_bytes() if mode == "ab" else b""
            with open(out_file, mode) as fh:
                r = subprocess.run([exe, str(golden.issl), str(golden.guides_txt), "4", "75", "and"], stdout=fh, stderr=subprocess.PIPE,
                                   env=dict(env, **extra), cwd=str(tmp_path))
            assert r.returncode == 0, r.stderr.decode()
            assert out_file.read_bytes() == before + golden.expected["and|75|4"].encode(), (mode, extra)
            assert b'"resident": true' in r.stderr and b'"write_ms"' in r.stderr
        # errors travel back with exit status 1 and nothing on stdout
        r = subprocess.run([exe, str(tmp_path / "missing.issl"), str(golden.guides_txt), "4", "75", "and"],
                           capture_output=True, env=env)
        assert r.returncode == 1 and r.stdout == b"" and b"cannot open index" in r.stderr
        with open(out_file, "wb") as fh:   # ... also when stdout is a file that went along with the request
            r = subprocess.run([exe, str(tmp_path / "missing.issl"), str(golden.guides_txt), "4", "75", "and"], stdout=fh,
                               stderr=subprocess.PIPE, env=env)
        assert r.returncode == 1 and out_file.read_bytes() == b"" and b"cannot open index" in r.stderr
        bad = tmp_path / "bad.txt"; bad.write_text("ACGT\n")
        r = subprocess.run([exe, str(golden.issl), str(bad), "4", "75", "and"], capture_output=True, env=env)
        assert r.returncode == 1 and r.stdout == b"" and b"multiple of the expected line length" in r.stderr
    finally:
        subprocess.run([exe, "--stop", sock], capture_output=True)
        try:
            server.wait(timeout=20)
        except subprocess.TimeoutExpired:
            server.kill()
    assert server.returncode == 0
    # no server behind the socket path: the process scores by itself
    env = dict(os.environ, ISSL_SERVER=str(tmp_path / "nobody.sock"))
    r = subprocess.run([exe, str(golden.issl), str(golden.guides_txt), "4", "75", "and"], capture_output=True, env=env)
    assert r.returncode == 0 and r.stdout.decode() == golden.expected["and|75|4"]


def test_resident_server_survives_bad_clients(golden_uniform, tmp_path):
    """The server outlives its clients: one that connects and says nothing (receive timeout), one that sends a request
    and hangs up before the answer (SIGPIPE ignored, MSG_NOSIGNAL); its socket is private (0600); an existing path that
    is not a socket is never replaced."""
    import socket
    import stat
    import time
    g = golden_uniform
    exe = str(ROOT / "bin" / "isslScoreOfftargets")
    precious = tmp_path / "not_a_socket"
    precious.write_text("keep me")
    r = subprocess.run([exe, "--serve", str(precious)], capture_output=True, timeout=30)
    assert r.returncode == 1 and precious.read_text() == "keep me" and b"not a socket" in r.stderr
    sock = str(tmp_path / "issl.sock")
    server = subprocess.Popen([exe, "--serve", sock], stderr=subprocess.PIPE, env=dict(os.environ, ISSL_SERVER_TIMEOUT_S="1"))
    try:
        for _ in range(100):
            if os.path.exists(sock):
                break
            time.sleep(0.05)
        assert stat.S_IMODE(os.stat(sock).st_mode) == 0o600
        env = dict(os.environ, ISSL_SERVER=sock, ISSL_TIMING="1")
        args = [exe, str(g.issl), str(g.guides_txt), "4", "75", "and"]
        silent = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        silent.connect(sock)                                   # says nothing: the server drops it after 1 s
        t0 = time.time()
        r = subprocess.run(args, capture_output=True, env=env, timeout=60)
        assert r.returncode == 0 and r.stdout.decode() == g.expected["and|75|4"] and time.time() - t0 < 20
        silent.close()
        rude = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        rude.connect(sock)
        rude.sendall(f"SCORE\t{g.issl}\t{g.guides_txt}\t4\t75\tand\n".encode())
        rude.close()                                           # gone before the answer is written
        r = subprocess.run(args, capture_output=True, env=env, timeout=60)
        assert r.returncode == 0 and r.stdout.decode() == g.expected["and|75|4"]
        assert b'"resident": true' in r.stderr and server.poll() is None
    finally:
        subprocess.run([exe, "--stop", sock], capture_output=True)
        try:
            server.wait(timeout=20)
        except subprocess.TimeoutExpired:
            server.kill()
    assert server.returncode == 0


def test_raw_buffer_growth_reruns_the_batch(golden_uniform, monkeypatch):
    """Start with a raw-record buffer far too small for the batch (fewer chunks than scan waves): the library must
    notice, grow it and re-run, with the golden results."""
    monkeypatch.setenv("ISSL_RAW_CHUNKS", "7")
    ix = ca.IsslIndex.open(golden_uniform.issl).upload(0)
    sigs = ca.encode_guides([g.encode() for g in golden_uniform.guides])
    mit, cfd = ix.score(sigs, 4, 75.0, "and")
    st = ix.stats()
    assert st["scan_launches"] >= 2, st
    assert ca.format_scores(sigs, mit, cfd, "and") == golden_uniform.expected["and|75|4"]
    hits = ix.dump_hits(sigs, 4, 0.0, "and")
    assert np.array_equal(hits, golden_uniform.hits(0))
    assert ix.stats()["scan_launches"] == 1  # capacity is kept
    ix.close()


def test_hit_counter(golden):
    ix = ca.IsslIndex.open(golden.issl).upload(0)
    sigs = ca.encode_guides([g.encode() for g in golden.guides])
    ix.score(sigs, 4, 0.0, "and")
    st = ix.stats()
    assert st["scan_launches"] == 1 and st["hits"] == len(golden.hits(0))
    ix.close()


@pytest.fixture(scope="module")
def config1(tmp_path_factory):
    """BASELINE.json configs[1] at full size: 10k guides vs a 50M-line synthetic index (48.8M distinct sites)."""
    sigs, occ = random_sites(50_000_000, seed=20261003)
    guides = random_guides(sigs, 10_000, seed=777)
    ix = ca.IsslIndex.build_from_sites(sigs, occ)
    ix.upload(0)
    yield ix, sigs, occ, guides
    ix.close()


def test_config1_full_size_properties_and_oracle_sample(config1, tmp_path):
    """Full-size run of the bench workload: size-independent properties on all 10k guides and bit-exact parity with
    the CPU oracle on a sample (the oracle needs ~12 ms per guide and core at this index size)."""
    ix, sigs, occ, guides = config1
    mit, cfd = ix.score(guides, 4, 75.0, "and")
    st = ix.stats()
    check_comparisons(ix, guides)
    assert st["scan_launches"] == 1
    assert np.all((mit > 0) & (mit <= 100) & (cfd > 0) & (cfd <= 100))
    # permutation and split invariance
    perm = np.random.default_rng(3).permutation(len(guides))
    pm, pc = ix.score(guides[perm], 4, 75.0, "and")
    assert np.array_equal(pm, mit[perm]) and np.array_equal(pc, cfd[perm])
    a = ix.score(guides[:3333], 4, 75.0, "and"); b = ix.score(guides[3333:], 4, 75.0, "and")
    assert np.array_equal(np.concatenate([a[0], b[0]]), mit) and np.array_equal(np.concatenate([a[1], b[1]]), cfd)
    # a guide that IS a site: its exact match contributes occ to the CFD total and nothing to MIT (dist 0)
    hits = ix.dump_hits(sigs[1000:1064], 0, 0.0, "and")
    assert len(hits) == 64 and np.array_equal(hits[:, 3], np.arange(1000, 1064)) and np.all(hits[:, 4] == 0)
    assert np.array_equal(hits[:, 5], occ[1000:1064])
    # checksum of checksums: hits found with max_dist d are a subset of those with d+1, first-slice rule included
    h3 = ix.dump_hits(guides[:500], 3, 0.0, "and"); h4 = ix.dump_hits(guides[:500], 4, 0.0, "and")
    k3 = set(map(tuple, h3[:, [0, 3]])); k4 = set(map(tuple, h4[:, [0, 3]]))
    assert k3 <= k4 and len(k4) > len(k3) and np.all(h3[:, 4] <= 3)
    # oracle on a sample
    p = tmp_path / "cfg1.issl"
    ix.write(p)
    oracle = ou.OracleIndex(p)
    sample = np.concatenate([guides[:160], guides[-96:]])
    gm = np.concatenate([mit[:160], mit[-96:]]); gc = np.concatenate([cfd[:160], cfd[-96:]])
    omit, ocfd = oracle.score(sample, 4, 75.0, "and")
    assert np.array_equal(gm.view(np.uint64), omit.view(np.uint64)) and np.array_equal(gc.view(np.uint64), ocfd.view(np.uint64))
    _, _, ohits = oracle.score(sample[:64], 4, 0.0, "and", want_hits=True)
    assert np.array_equal(ix.dump_hits(sample[:64], 4, 0.0, "and"), ohits)
    oracle.close()
    os.unlink(p)


def test_index_built_on_device_is_byte_identical(golden, tmp_path):
    """issl_index_build_on_device: slice lists from one radix pass per slice on the GPU; written back out it is the
    reference builder's file, and it scores like the host-built index."""
    lines = golden.sites_txt.read_text().splitlines()
    sigs_all = ca.encode_guides(lines)
    # consecutive duplicates collapse (isslCreateIndex.cpp:189-193): the list is sorted
    keep = np.ones(len(sigs_all), dtype=bool)
    keep[1:] = sigs_all[1:] != sigs_all[:-1]
    first = np.flatnonzero(keep)
    sigs = sigs_all[first]
    occ = np.diff(np.append(first, len(sigs_all))).astype(np.uint32)
    ix = ca.IsslIndex.build_on_device(sigs, occ, device=0)
    out = tmp_path / "dev.issl"
    ix.write(out)
    assert out.read_bytes() == golden.issl.read_bytes()
    guides = ca.encode_guides(golden.guides)
    mit, cfd = ix.score(guides, 4, 75.0, "and")
    assert ca.format_scores(guides, mit, cfd, "and") == golden.expected["and|75|4"]
    with pytest.raises(ca.IsslError, match="built on the device"):
        ix.upload(1)   # its arrays exist only in the image on device 0
    ix.upload(0)       # same device: nothing to do
    ix.close()


@pytest.mark.parametrize("options", [None, {"compact": 1, "keep_lists": 0}])
def test_narrow_index_built_on_device_is_byte_identical(golden_width, tmp_path, options):
    """The device-side builder with 4- and 2-bit slices (one radix pass per slice over the slice's own bits): written back
    out it is the reference builder's file (tests/golden/width4, width2) -- also from an image that holds no slice lists,
    which are made again for the file --, and it scores like the host-built index."""
    g = golden_width
    width = int(g.name[5:])
    lines = g.sites_txt.read_text().splitlines()
    sigs_all = ca.encode_guides(lines)
    keep = np.ones(len(sigs_all), dtype=bool)
    keep[1:] = sigs_all[1:] != sigs_all[:-1]
    first = np.flatnonzero(keep)
    sigs = sigs_all[first]
    occ = np.diff(np.append(first, len(sigs_all))).astype(np.uint32)
    ix = ca.IsslIndex.build_on_device(sigs, occ, device=0, slice_width=width, options=options)
    assert ix.get_option("is_sorted") == 1 and ix.get_option("lists_absent") == (1 if options else 0)
    out = tmp_path / "dev.issl"
    ix.write(out)
    assert out.read_bytes() == g.issl.read_bytes()
    guides = ca.encode_guides(g.guides)
    for key in ("and|75|4", "and|0|2", "cfd|75|6"):
        method, thr, dist = key.split("|")
        mit, cfd = ix.score(guides, int(dist), float(thr), method)
        assert ca.format_scores(guides, mit, cfd, method) == g.expected[key], key
    for thr in g.hit_thresholds():
        assert np.array_equal(ix.dump_hits(guides, 4, float(thr), "and"), g.hits(thr)), thr
    ix.close()


def test_device_builder_matches_host_builder_on_config0(config0, tmp_path):
    """1 M sites (245 workgroups per radix pass): same .issl bytes as the host builder, same scores."""
    ix, oracle, sigs, guides = config0
    _, occ = random_sites(1_000_000, seed=2026)
    dev = ca.IsslIndex.build_on_device(sigs, occ, device=0)
    assert dev.device_bytes() == ix.device_bytes()
    assert np.array_equal(dev.bucket_sizes(), ix.bucket_sizes())
    a, b = tmp_path / "host.issl", tmp_path / "dev.issl"
    ix.write(a)
    dev.write(b)
    assert a.read_bytes() == b.read_bytes()
    mit, cfd = dev.score(guides, 4, 75.0, "and")
    wm, wc = ix.score(guides, 4, 75.0, "and")
    assert np.array_equal(mit, wm) and np.array_equal(cfd, wc)
    dev.close()


@pytest.mark.parametrize("n_in_bucket", [1, 31, 32, 33, 2047, 2048, 2049, 4096, 4097])
def test_bucket_lengths_around_group_and_tile_boundaries(tmp_path, n_in_bucket):
    """Buckets whose length sits on the edges of the scan layout (32 candidates per lane, 2048 per tile): every site of
    the index shares its first four bases, so slice 0 holds ONE bucket of exactly n sites (zero padding after it) and
    the other slices are spread; hits at the first and the last position of the bucket."""
    rng = np.random.default_rng(1000 + n_in_bucket)
    low = np.uint64(0b10_01_11_00)                                   # positions 0..3 of every site
    rest = np.unique(rng.integers(0, 1 << 32, size=n_in_bucket * 2, dtype=np.uint64))[:n_in_bucket]
    assert len(rest) == n_in_bucket
    sig = (rest << np.uint64(8)) | low
    sig = sig[np.argsort(text_order_key(sig), kind="stable")]
    occ = rng.integers(1, 5, size=len(sig)).astype(np.uint32)
    ix = ca.IsslIndex.build_from_sites(sig, occ)
    assert int(ix.bucket_sizes()[int(low)]) == n_in_bucket
    p = tmp_path / "edge.issl"
    ix.write(p)
    ix.upload(0)
    oracle = ou.OracleIndex(p)
    # guides: the first and the last site of the bucket (in bucket order = id order), each also with substitutions
    # outside and inside slice 0, plus guides made of padding-like words
    picks = np.array([sig[0], sig[-1], sig[len(sig) // 2]], dtype=np.uint64)
    guides = np.concatenate([picks, picks ^ np.uint64(1 << 20), picks ^ np.uint64((3 << 10) | (2 << 30)),
                             picks ^ np.uint64(1), np.array([0, low, (1 << 40) - 1], dtype=np.uint64)])
    for dist, thr in ((4, 0.0), (4, 75.0), (2, 0.0), (0, 0.0)):
        hits = ix.dump_hits(guides, dist, thr, "and")
        omit, ocfd, ohits = oracle.score(guides, dist, thr, "and", want_hits=True)
        assert np.array_equal(hits, ohits), (dist, thr)
        mit, cfd = ix.score(guides, dist, thr, "and")
        assert np.array_equal(mit.view(np.uint64), omit.view(np.uint64)) and np.array_equal(cfd.view(np.uint64), ocfd.view(np.uint64))
    assert len(ohits) >= 3
    ix.close()


def test_random_small_indexes_differential():
    """Differential sweep: 12 random small indexes (clustered so that guides have neighbours at every distance), random
    image layouts (with / without the in-list signatures, cold sections in host memory), methods, thresholds and
    distances, hit lists and scores against the oracle.  ISSL_FUZZ_TRIALS / ISSL_FUZZ_SEED run longer campaigns
    (profiles/r02_fuzz_campaign.txt)."""
    import tempfile
    trials = int(os.environ.get("ISSL_FUZZ_TRIALS", 12))
    rng = np.random.default_rng(int(os.environ.get("ISSL_FUZZ_SEED", 424242)))
    methods = ["and", "or", "avg", "mit", "cfd"]
    from test_layouts import LAYOUTS, SORTED
    names = list(LAYOUTS)
    with tempfile.TemporaryDirectory() as tmp:
        for trial in range(trials):
            n_centres = int(rng.integers(1, 40))
            centres = rng.integers(0, 1 << 40, size=n_centres, dtype=np.uint64)
            sites = set(int(c) for c in centres)
            for c in centres:
                for _ in range(int(rng.integers(0, 60))):
                    s = int(c)
                    for pos in rng.choice(20, size=int(rng.integers(1, 6)), replace=False):
                        s ^= int(rng.integers(1, 4)) << (2 * int(pos))
                    sites.add(s)
            sites |= set(int(x) for x in rng.integers(0, 1 << 40, size=int(rng.integers(0, 3000)), dtype=np.uint64))
            sig = np.array(sorted(sites), dtype=np.uint64)
            sig = sig[np.argsort(text_order_key(sig), kind="stable")]
            occ = rng.integers(1, 7, size=len(sig)).astype(np.uint32)
            width = (4, 2)[trial // 4 % 2] if trial % 4 == 3 else 8   # narrow slices: sorted layouts by the byte of the next two / four slices
            ix = ca.IsslIndex.build_from_sites(sig, occ, slice_width=width)
            path = os.path.join(tmp, f"t{trial}.issl")
            ix.write(path)
            name = names[int(rng.integers(0, len(names)))] if trial % 3 else SORTED[(trial // 3) % len(SORTED)]
            if width != 8 and name not in ("sorted", "compact", "list", "list_esig"):
                name = ("sorted", "compact")[trial // 8 % 2]
            layout = dict(LAYOUTS[name])
            if name in SORTED:
                layout["prune"] = int(rng.integers(-1, 2)) if trial % 2 else 1
            layout["hit_slots"] = int(rng.integers(0, 3))   # hits straight to per-guide slots (2: the wide ones), or all through the grouping pass
            for key, value in layout.items():
                ix.set_option(key, value)
            ix.upload(0)
            assert ix.get_option("is_sorted") == (1 if name in SORTED else 0)
            layout["name"] = name
            oracle = ou.OracleIndex(path)
            guides = np.concatenate([centres, centres ^ np.uint64(2 << 16), rng.integers(0, 1 << 40, size=5, dtype=np.uint64)])
            for _ in range(4):
                method = methods[int(rng.integers(0, 5))]
                thr = float(rng.choice([0.0, 30.0, 75.0, 95.0, 100.0]))
                dist = int(rng.integers(0, 6))
                hits = ix.dump_hits(guides, dist, thr, method)
                omit, ocfd, ohits = oracle.score(guides, dist, thr, method, want_hits=True)
                assert np.array_equal(hits, ohits), (trial, layout, method, thr, dist)
                mit, cfd = ix.score(guides, dist, thr, method)
                assert np.array_equal(mit.view(np.uint64), omit.view(np.uint64)), (trial, layout, method, thr, dist)
                assert np.array_equal(cfd.view(np.uint64), ocfd.view(np.uint64)), (trial, layout, method, thr, dist)
            oracle.close()
            ix.close()


@pytest.mark.parametrize("width", [4, 2])
def test_narrow_slices_on_the_sorted_layouts_against_the_oracle(tmp_path, width):
    """Ten 4-bit / twenty 2-bit slices (isslScoreOfftargets.cpp:261-270,330-341 take any width): the sorted layouts order every
    bucket by the byte of the next two / four slices and the pruned scan visits 13 (1, 67) of its 256 groups.  An index of 120 k sites -- dense
    neighbourhoods, so that guides have hits at every distance and in every slice, and groups of several scan windows -- on
    the sorted and the compact layout, pruned scan forced, whole buckets, planner's choice: hit lists and scores for max_dist
    0..6 against the oracle (5: 67 groups; 6: whole buckets), early exit on and off."""
    rng = np.random.default_rng(4040)
    centres = rng.integers(0, 1 << 40, size=150, dtype=np.uint64)
    sites = set(int(c) for c in centres)
    for c in centres:
        for _ in range(int(rng.integers(20, 400))):
            s = int(c)
            for pos in rng.choice(20, size=int(rng.integers(1, 7)), replace=False):
                s ^= int(rng.integers(1, 4)) << (2 * int(pos))
            sites.add(s)
    # one bucket (slice 0 = 0x5) with a few long successor-byte groups: several scan windows per group
    for _ in range(30000):
        sites.add(0x5 | (int(rng.integers(0, 3)) << 4) | (0x2 << 8) | (int(rng.integers(0, 1 << 28)) << 12))
    sites |= set(int(x) for x in rng.integers(0, 1 << 40, size=60000, dtype=np.uint64))
    sig = np.array(sorted(sites), dtype=np.uint64)
    sig = sig[np.argsort(text_order_key(sig), kind="stable")]
    occ = rng.integers(1, 5, size=len(sig)).astype(np.uint32)
    path = tmp_path / "w4.issl"
    ca.IsslIndex.build_from_sites(sig, occ, slice_width=width).write(path)
    oracle = ou.OracleIndex(path)
    extra = np.array([0x5 | (1 << 4) | (0x2 << 8) | (int(x) << 12) for x in rng.integers(0, 1 << 28, size=40)], dtype=np.uint64)
    guides = np.concatenate([centres, centres ^ np.uint64(3 << 10), extra, rng.integers(0, 1 << 40, size=30, dtype=np.uint64)])
    want = {}
    for dist in range(0, 7):
        for thr in (0.0, 75.0):
            want[dist, thr] = oracle.score(guides, dist, thr, "and", want_hits=True)
    assert len(want[4, 0.0][2]) > 20000 and len(want[4, 75.0][2]) < len(want[4, 0.0][2])
    for layout in ({"sorted_layout": 1, "compact": 0}, {"compact": 1}):
        ix = ca.IsslIndex.open(path)
        for key, value in layout.items():
            ix.set_option(key, value)
        ix.upload(0)
        assert ix.get_option("is_sorted") == 1 and ix.header["n_slices"] == 40 // width
        try:
            for prune in (1, 0, -1):
                ix.set_option("prune", prune)
                for (dist, thr), (omit, ocfd, ohits) in want.items():
                    hits = ix.dump_hits(guides, dist, thr, "and")
                    assert np.array_equal(hits, ohits), (layout, prune, dist, thr)
                    mit, cfd = ix.score(guides, dist, thr, "and")
                    st = ix.stats()
                    assert np.array_equal(mit.view(np.uint64), omit.view(np.uint64)), (layout, prune, dist, thr)
                    assert np.array_equal(cfd.view(np.uint64), ocfd.view(np.uint64)), (layout, prune, dist, thr)
                    if prune == 1:
                        assert st["pruned"] == (1 if dist <= 2 else 2 if dist <= 4 else 3 if dist == 5 else 0), (dist, st["pruned"])
                    if prune == 0:
                        assert st["pruned"] == 0 and st["candidates"] == st["reference_comparisons"]
            for method, thr, dist in (("mit", 50.0, 4), ("cfd", 90.0, 3), ("or", 75.0, 4), ("avg", 30.0, 2)):
                ix.set_option("prune", 1)
                mit, cfd = ix.score(guides, dist, thr, method)
                omit, ocfd = oracle.score(guides, dist, thr, method)
                assert np.array_equal(mit.view(np.uint64), omit.view(np.uint64)) and np.array_equal(cfd.view(np.uint64), ocfd.view(np.uint64)), (method, thr, dist)
        finally:
            ix.close()
    oracle.close()


def test_runtime_threshold_kernel_with_the_pruned_scan(config0):
    """The runtime-threshold build of the scan kernel (scan_generic) working through the successor-byte groups."""
    ix, oracle, sigs, guides = config0
    ix.set_option("scan_generic", 1).set_option("prune", 1)
    try:
        for dist in (0, 1, 2, 3, 4):
            hits = ix.dump_hits(guides[:300], dist, 0.0, "and")
            assert ix.stats()["pruned"] == (1 if dist <= 2 else 2)
            _, _, ohits = oracle.score(guides[:300], dist, 0.0, "and", want_hits=True)
            assert np.array_equal(hits, ohits), dist
    finally:
        ix.set_option("scan_generic", 0).set_option("prune", -1)


def test_pruned_scan_on_groups_of_many_windows(tmp_path):
    """Successor-byte groups that are several scan windows long, start in the middle of a tile and share tiles with their
    neighbours -- at a size the oracle scores in full: 70 k sites, most of them with the same first slice and one of three
    neighbouring values of the second one.  Hit lists and scores for max_dist 0..4, pruned and whole-bucket scan."""
    rng = np.random.default_rng(77)
    a, b = 0x5A, 0xC3
    centres = rng.integers(0, 1 << 24, size=200, dtype=np.uint64)

    def rest(n):   # the other 12 positions: a centre with up to three substitutions, so that guides have many neighbours
        r = centres[rng.integers(0, len(centres), size=n)]
        for _ in range(3):
            r = r ^ (rng.integers(0, 4, size=n, dtype=np.uint64) << (np.uint64(2) * rng.integers(0, 12, size=n).astype(np.uint64)))
        return r << np.uint64(16)

    sites = np.concatenate([
        rest(30_000) | np.uint64(a) | (np.uint64(b) << np.uint64(8)),
        rest(15_000) | np.uint64(a) | (np.uint64(b ^ 1) << np.uint64(8)),      # neighbouring group, one mismatch away
        rest(7_000) | np.uint64(a) | (np.uint64(b ^ 0x10) << np.uint64(8)),    # another single-mismatch group
        rest(5_000) | np.uint64(a) | (np.uint64(b ^ 0x11) << np.uint64(8)),    # two mismatches in the successor byte
        rng.integers(0, 1 << 40, size=13_000, dtype=np.uint64)])
    sig = np.unique(sites)
    sig = sig[np.argsort(text_order_key(sig), kind="stable")]
    occ = rng.integers(1, 4, size=len(sig)).astype(np.uint32)
    ix = ca.IsslIndex.build_from_sites(sig, occ)
    path = tmp_path / "windows.issl"
    ix.write(path)
    ix.upload(0)
    assert ix.get_option("is_sorted") == 1
    oracle = ou.OracleIndex(path)
    picks = sig[rng.integers(0, len(sig), size=400)]
    guides = picks.copy()
    for k in range(len(guides)):  # 0..4 substitutions anywhere
        for pos in rng.choice(20, size=int(rng.integers(0, 5)), replace=False):
            guides[k] ^= np.uint64(int(rng.integers(1, 4)) << (2 * int(pos)))
    try:
        for prune in (1, 0):
            ix.set_option("prune", prune)
            for dist, thr, method in ((4, 0.0, "and"), (4, 75.0, "and"), (3, 0.0, "or"), (2, 0.0, "mit"), (1, 50.0, "cfd"), (0, 0.0, "avg")):
                hits = ix.dump_hits(guides, dist, thr, method)
                st = ix.stats()
                assert st["pruned"] == (0 if prune == 0 else (1 if dist <= 2 else 2))
                omit, ocfd, ohits = oracle.score(guides, dist, thr, method, want_hits=True)
                assert np.array_equal(hits, ohits), (prune, dist, thr, method)
                if (dist, thr) == (4, 0.0):
                    assert len(ohits) > 20_000   # (the guides do have neighbours in the big groups)
                mit, cfd = ix.score(guides, dist, thr, method)
                assert np.array_equal(mit.view(np.uint64), omit.view(np.uint64)), (prune, dist, thr, method)
                assert np.array_equal(cfd.view(np.uint64), ocfd.view(np.uint64)), (prune, dist, thr, method)
            if prune == 1:
                assert st["scan_tiles"] > 5 * 13   # the big groups are several windows long
    finally:
        oracle.close()
        ix.close()


def test_resident_server_releases_the_least_recently_used_index(tmp_path):
    """Three indexes, room for two (ISSL_SERVER_HBM_BUDGET): asking for the third releases the one that was used longest
    ago, the others stay resident hits; every answer equals the reference's stdout (Crackling's per-page reload of
    config.ini:106-112 against a server that holds several genomes)."""
    import json
    import time
    from conftest import Golden
    sets = [Golden("uniform"), Golden("clustered"), Golden("bigocc")]
    sizes = []
    for g in sets:
        ix = ca.IsslIndex.open(g.issl)
        sizes.append(ix.device_bytes())
        ix.close()
    assert max(sizes) < 1.15 * min(sizes)
    # the server wants image + 24 B/site of temporaries + 10 GiB of workspace free before an upload: room for two images
    # and that reserve, not for three
    budget = int(2.5 * max(sizes)) + 24 * 8200 + (10 << 30)
    exe = str(ROOT / "bin" / "isslScoreOfftargets")
    sock = str(tmp_path / "issl.sock")
    server = subprocess.Popen([exe, "--serve", sock], stderr=subprocess.PIPE, env=dict(os.environ, ISSL_SERVER_HBM_BUDGET=str(budget)))
    try:
        for _ in range(100):
            if os.path.exists(sock):
                break
            time.sleep(0.05)
        env = dict(os.environ, ISSL_SERVER=sock, ISSL_TIMING="1")

        def ask(g, key="and|75|4"):
            method, thr, dist = key.split("|")
            r = subprocess.run([exe, str(g.issl), str(g.guides_txt), dist, thr, method], capture_output=True, env=env)
            assert r.returncode == 0 and r.stdout.decode() == g.expected[key], r.stderr.decode()
            return b'"resident": true' in r.stderr

        def status():
            r = subprocess.run([exe, "--status", sock], capture_output=True)
            assert r.returncode == 0, r.stderr
            return json.loads(r.stdout)
        a, b, c = sets
        assert not ask(a) and not ask(b) and ask(a)            # a, b resident; a used last
        assert status()["evictions"] == 0 and len(status()["resident"]) == 2
        assert not ask(c)                                       # no room for three: b (least recently used) goes
        st = status()
        assert st["evictions"] == 1 and [os.path.basename(os.path.dirname(x["issl"])) for x in st["resident"]] == ["uniform", "bigocc"]
        assert ask(a) and ask(c) and not ask(b)                 # b comes back, a (used before c) goes
        assert [os.path.basename(os.path.dirname(x["issl"])) for x in status()["resident"]] == ["bigocc", "clustered"]
    finally:
        subprocess.run([exe, "--stop", sock], capture_output=True)
        try:
            server.wait(timeout=20)
        except subprocess.TimeoutExpired:
            server.kill()
    assert server.returncode == 0


def test_lean_tail_prediction_and_its_retry():
    """A handle whose batches meet no guide with more than 512 hits stops launching the grouping pass and the many-hit replays
    (Workspace::lean_tail); a batch that then DOES meet such a guide is run again with the whole pipeline -- through issl_score by
    itself, through the asynchronous entry points as ISSL_E_RETRY.  tests/golden/signedtable: guides with 600 ... 1800 hits beside
    guides with a handful, reference stdout."""
    import torch
    from conftest import Golden
    g = Golden("signedtable")
    sigs = ca.encode_guides([s.encode() for s in g.guides])
    key = next(k for k in g.expected if k.startswith("and|"))
    method, thr, dist = key.split("|")
    ix = ca.IsslIndex.open(g.issl).upload(0)
    if ix.get_option("lean_tail") != 1 or ix.get_option("hit_slots") == 0:   # (a whole-suite run with ISSL_LEAN_TAIL=0 / ISSL_HIT_SLOTS=0 in the environment)
        ix.close()
        pytest.skip("the environment switches the lean tail off")
    try:
        counts = np.bincount(ix.dump_hits(sigs, int(dist), 0.0, method)[:, 0], minlength=len(sigs))
        few = np.flatnonzero(counts <= 400)
        assert len(few) >= 2 and (counts > 512).any()
        want = g.expected[key].splitlines(keepends=True)
        for rounds in range(2):
            for _ in range(2):   # batches without a many-hit guide: the lane turns lean
                mit, cfd = ix.score(sigs[few], int(dist), float(thr), method)
                assert ca.format_scores(sigs[few], mit, cfd, method) == "".join(want[i] for i in few)
            assert ix.stats()["scan_launches"] == 1
            mit, cfd = ix.score(sigs, int(dist), float(thr), method)   # mispredicted: run again inside the call
            assert ca.format_scores(sigs, mit, cfd, method) == g.expected[key]
            assert ix.stats()["scan_launches"] == 2
            mit, cfd = ix.score(sigs, int(dist), float(thr), method)   # the lane knows now
            assert ca.format_scores(sigs, mit, cfd, method) == g.expected[key] and ix.stats()["scan_launches"] == 1
        # asynchronous batches: finish() says "again" once
        for _ in range(2):
            ix.score(sigs[few], int(dist), float(thr), method)
        d_g = torch.from_numpy(sigs.view(np.int64)).cuda()
        d_m = torch.empty(len(sigs), dtype=torch.float64, device="cuda:0"); d_c = torch.empty_like(d_m)
        tries = 0
        while True:
            ix.score_device_async(d_g, d_m, d_c, int(dist), float(thr), method)
            tries += 1
            if ix.finish():
                break
        assert tries == 2
        assert ca.format_scores(sigs, d_m.cpu().numpy(), d_c.cpu().numpy(), method) == g.expected[key]
        ix.set_option("lean_tail", 0)   # the knob: never lean
        for _ in range(2):
            ix.score(sigs[few], int(dist), float(thr), method)
        ix.score(sigs, int(dist), float(thr), method)
        assert ix.stats()["scan_launches"] == 1
    finally:
        ix.close()


def test_one_workgroup_binning_of_small_batches(tmp_path):
    """A batch of up to 512 (guide, slice) pairs is binned by ONE workgroup in one launch (k_bin_small: every placement a group
    of its own) where the general path takes seven.  On the index of the many-windows test -- groups several units long that
    start inside tiles, guides that SHARE groups (the small path then scans those twice) -- batches of 1 ... 103 guides (the
    103rd no longer fits: general path) must give the oracle's hit lists and scores bit for bit with the path on and off,
    for max_dist 0 ... 4 (13 ways, 1 way) and the runtime-threshold kernel; then the same with an item list that is too short
    for the plan: the small path asks for the batch again (the ISSL_E_RETRY round inside issl_score), the general path scans
    whole buckets that once; both have the room afterwards."""
    rng = np.random.default_rng(78)
    a, b = 0x5A, 0xC3
    centres = rng.integers(0, 1 << 24, size=200, dtype=np.uint64)

    def rest(n):
        r = centres[rng.integers(0, len(centres), size=n)]
        for _ in range(3):
            r = r ^ (rng.integers(0, 4, size=n, dtype=np.uint64) << (np.uint64(2) * rng.integers(0, 12, size=n).astype(np.uint64)))
        return r << np.uint64(16)

    sites = np.concatenate([
        rest(30_000) | np.uint64(a) | (np.uint64(b) << np.uint64(8)),
        rest(15_000) | np.uint64(a) | (np.uint64(b ^ 1) << np.uint64(8)),
        rest(7_000) | np.uint64(a) | (np.uint64(b ^ 0x10) << np.uint64(8)),
        rng.integers(0, 1 << 40, size=13_000, dtype=np.uint64)])
    sig = np.unique(sites)
    sig = sig[np.argsort(text_order_key(sig), kind="stable")]
    occ = rng.integers(1, 4, size=len(sig)).astype(np.uint32)
    ix = ca.IsslIndex.build_from_sites(sig, occ)
    path = tmp_path / "small.issl"
    ix.write(path)
    ix.close()
    oracle = ou.OracleIndex(path)
    picks = sig[rng.integers(0, len(sig), size=103)]
    guides = picks.copy()
    for k in range(len(guides)):
        for pos in rng.choice(20, size=int(rng.integers(0, 5)), replace=False):
            guides[k] ^= np.uint64(int(rng.integers(1, 4)) << (2 * int(pos)))
    guides[5] = guides[4]          # the same guide twice, and
    guides[7] = guides[6] ^ np.uint64(1 << 38)  # two that share every group of four slices
    want = {}
    try:
        for small in (1, 0):
            ix = ca.IsslIndex.open(path)
            ix.set_option("prune", 1).set_option("small_bin", small)
            ix.upload(0)
            assert ix.get_option("is_sorted") == 1
            for n in (1, 2, 8, 64, 102, 103):
                for dist, thr, method in ((4, 0.0, "and"), (4, 75.0, "and"), (3, 0.0, "or"), (2, 0.0, "mit"), (0, 0.0, "avg")):
                    key = (n, dist, thr, method)
                    if key not in want:
                        want[key] = oracle.score(guides[:n], dist, thr, method, want_hits=True)
                    omit, ocfd, ohits = want[key]
                    hits = ix.dump_hits(guides[:n], dist, thr, method)
                    assert np.array_equal(hits, ohits), (small, key)
                    mit, cfd = ix.score(guides[:n], dist, thr, method)
                    st = ix.stats()
                    assert np.array_equal(mit.view(np.uint64), omit.view(np.uint64)), (small, key)
                    assert np.array_equal(cfd.view(np.uint64), ocfd.view(np.uint64)), (small, key)
                    assert st["pruned"] == (1 if dist <= 2 else 2) and st["scan_launches"] == 1
                    assert st["reference_comparisons"] == ix.count_candidates(guides[:n])
            ix.set_option("scan_generic", 1)   # the runtime-threshold build of the scan
            mit, cfd = ix.score(guides[:64], 4, 0.0, "and")
            assert np.array_equal(mit.view(np.uint64), want[(64, 4, 0.0, "and")][0].view(np.uint64))
            ix.set_option("scan_generic", 0)
            # other launch shapes of the scan: ranges, record chunks and the range search follow the knobs
            for knobs in ({"scan_blocks": 4096}, {"scan_blocks": 64}, {"scan_blocks": 1024, "scan_threads": 768}, {"scan_threads": 256}):
                for k, val in knobs.items():
                    ix.set_option(k, val)
                for n in (1, 64, 102):
                    mit, cfd = ix.score(guides[:n], 4, 0.0, "and")
                    assert np.array_equal(mit.view(np.uint64), want[(n, 4, 0.0, "and")][0].view(np.uint64)), (small, knobs, n)
                    assert np.array_equal(cfd.view(np.uint64), want[(n, 4, 0.0, "and")][1].view(np.uint64)), (small, knobs, n)
                    assert np.array_equal(ix.dump_hits(guides[:n], 4, 0.0, "and"), want[(n, 4, 0.0, "and")][2]), (small, knobs, n)
            ix.close()
        # an item list too short for the plan
        for small in (1, 0):
            ix = ca.IsslIndex.open(path)
            ix.set_option("prune", 1).set_option("small_bin", small).set_option("fine_items", 16)
            ix.upload(0)
            omit, ocfd, _ = want[(64, 4, 0.0, "and")]
            mit, cfd = ix.score(guides[:64], 4, 0.0, "and")
            st = ix.stats()
            assert np.array_equal(mit.view(np.uint64), omit.view(np.uint64)) and np.array_equal(cfd.view(np.uint64), ocfd.view(np.uint64))
            if small:
                assert st["scan_launches"] == 2 and st["pruned"] == 2     # asked for again, with room
            else:
                assert st["scan_launches"] == 1 and st["pruned"] == 0     # whole buckets that once
            mit, cfd = ix.score(guides[:64], 4, 0.0, "and")
            st = ix.stats()
            assert np.array_equal(mit.view(np.uint64), omit.view(np.uint64)) and np.array_equal(cfd.view(np.uint64), ocfd.view(np.uint64))
            assert st["scan_launches"] == 1 and st["pruned"] == 2
            ix.close()
    finally:
        oracle.close()


def test_cli_writes_a_large_page_where_its_stdout_points(golden_uniform, tmp_path):
    """A page whose text is several megabytes (the library formats it in pieces, on several threads), wherever the caller's
    stdout points: a fresh file, a file with something in front (the text starts at the descriptor's position, which ends behind
    the text), a descriptor in append mode, a pipe -- the same bytes as the in-process scores formatted by the library, and the
    same through the resident server, which writes to the client's descriptor itself.  (Writing the pieces with pwrite from
    several threads was measured and dropped: 12.6 ms against 7.1 for the 41 MB of a 1 M-guide page -- writers of one file
    wait for each other; profiles/r05_cli_ab_pwrite.log.)"""
    import sys
    sys.path.insert(0, str(ROOT / "tools"))
    import cli_end_to_end as e2e
    exe = ROOT / "bin" / "isslScoreOfftargets"
    g = golden_uniform
    rng = np.random.default_rng(5)
    base = ca.encode_guides([s.encode() for s in g.guides if set(s) <= set("ACGT")])
    guides = np.concatenate([base, rng.integers(0, 1 << 40, size=130_000, dtype=np.uint64)])
    query = tmp_path / "big.q"
    e2e.write_query(query, guides)
    with ca.IsslIndex.open(g.issl) as ix:
        ix.upload(0)
        mit, cfd = ix.score(guides, 4, 75.0, "and")
        want = ca.format_scores(guides, mit, cfd, "and").encode()
    assert len(want) > (4 << 20)
    args = [str(exe), str(g.issl), str(query), "4", "75", "and"]

    def run(env=None):
        fresh = tmp_path / "fresh.out"
        with open(fresh, "wb") as fh:
            assert subprocess.run(args, stdout=fh, env=env).returncode == 0
        assert fresh.read_bytes() == want
        behind = tmp_path / "behind.out"
        with open(behind, "wb") as fh:
            fh.write(b"# header\n")
            fh.flush()
            assert subprocess.run(args, stdout=fh, env=env).returncode == 0
            fh.write(b"# trailer\n")   # (our own position is shared with the child's: it must sit behind the text)
        assert behind.read_bytes() == b"# header\n" + want + b"# trailer\n"
        appended = tmp_path / "appended.out"
        appended.write_bytes(b"# first\n")
        with open(appended, "ab") as fh:
            assert subprocess.run(args, stdout=fh, env=env).returncode == 0
        assert appended.read_bytes() == b"# first\n" + want
        r = subprocess.run(args, stdout=subprocess.PIPE, env=env)
        assert r.returncode == 0 and r.stdout == want

    run()
    sock = str(tmp_path / "srv.sock")
    import time
    srv = subprocess.Popen([str(exe), "--serve", sock], stderr=subprocess.PIPE)
    try:
        for _ in range(100):
            if os.path.exists(sock):
                break
            time.sleep(0.1)
        assert os.path.exists(sock)
        run(dict(os.environ, ISSL_SERVER=sock))
    finally:
        srv.terminate()
        srv.wait(timeout=30)
