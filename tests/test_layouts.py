"""Every image layout against every golden set of the compiled reference (stdout text and hit lists), through the C ABI.

Layouts:
  sorted        stream ordered by the successor slice's byte, 16-byte stream records          (pruned scan)
  compact       the same order, 4-byte site ids per stream position                           (pruned scan)
  compact_cold  ... with the slice lists in pinned host memory                                (pruned scan)
  compact_bare  ... without slice lists at all: 52 B/site, what a 3 G-site index gets         (pruned scan)
  list_esig     stream in list order, in-list signatures
  list          stream in list order
  host_cold     stream in list order, site table and lists in pinned host memory (occ8 + plane rebuild)
The sorted layouts score in (slice, site id) order, the list-order ones in (slice, list position) order; on indexes whose
lists ascend by id -- every builder's -- the two are the reference's order (isslScoreOfftargets.cpp:330-348)."""
import pathlib

import numpy as np
import pytest

import crackling_amd as ca
import oracle_util as ou
from conftest import Golden

pytestmark = pytest.mark.gpu

LAYOUTS = {
    "sorted": {"sorted_layout": 1, "compact": 0, "host_cold": 0},
    "compact": {"compact": 1, "host_cold": 0, "keep_lists": 1},
    "compact_cold": {"compact": 1, "host_cold": 1},
    "compact_bare": {"compact": 1, "keep_lists": 0},
    "list_esig": {"sorted_layout": 0, "inline_sigs": 1, "host_cold": 0},
    "list": {"sorted_layout": 0, "inline_sigs": 0, "host_cold": 0},
    "host_cold": {"sorted_layout": 0, "host_cold": 1},
}
SORTED = ("sorted", "compact", "compact_cold", "compact_bare")
SETS = ["uniform", "clustered", "edge", "oddtable", "bigocc", "signedtable"]


def _open(path_or_bytes, layout):
    ix = ca.IsslIndex.from_bytes(path_or_bytes) if isinstance(path_or_bytes, (bytes, bytearray)) else ca.IsslIndex.open(path_or_bytes)
    for key, value in LAYOUTS[layout].items():
        ix.set_option(key, value)
    return ix


def _check_layout(ix, layout):
    assert ix.get_option("is_sorted") == (1 if layout in SORTED else 0)
    assert ix.get_option("is_compact") == (1 if layout in ("compact", "compact_cold", "compact_bare") else 0)
    assert ix.get_option("lists_absent") == (1 if layout == "compact_bare" else 0)
    assert ix.get_option("cold_sections") == {"compact_cold": 1, "host_cold": 3}.get(layout, 0)
    assert ix.get_option("has_inline_sigs") == (1 if layout == "list_esig" else 0)


@pytest.mark.parametrize("layout", list(LAYOUTS))
@pytest.mark.parametrize("name", SETS)
def test_every_layout_reproduces_the_reference(name, layout):
    g = Golden(name)
    ix = _open(g.issl, layout).upload(0)
    _check_layout(ix, layout)
    sigs = ca.encode_guides([s.encode() for s in g.guides])
    try:
        for prune in ((0, 1) if layout in SORTED else (-1,)):
            ix.set_option("prune", prune)
            for key, want in g.expected.items():
                method, thr, dist = key.split("|")
                mit, cfd = ix.score(sigs, int(dist), float(thr), method)
                assert ca.format_scores(sigs, mit, cfd, method) == want, (key, prune)
                if prune == 1 and 0 <= int(dist) <= 4:
                    assert ix.stats()["pruned"] == (1 if int(dist) <= 2 else 2)
            for thr in g.hit_thresholds():
                hits = ix.dump_hits(sigs, 4, float(thr), "and")
                want = g.hits(thr)
                assert hits.shape == want.shape, (thr, prune, hits.shape, want.shape)
                assert np.array_equal(hits, want), (thr, prune)
    finally:
        ix.close()


@pytest.mark.parametrize("layout", [None] + list(LAYOUTS))
@pytest.mark.parametrize("name", ["width4", "width2"])
def test_narrow_slices_reproduce_the_reference(name, layout):
    """Indexes with 4- and 2-bit slices (10 / 20 slices per site; isslScoreOfftargets.cpp:261-270,330-341 is generic in
    both).  List-order images scan whole buckets -- the scan word keeps 16 of the 18 / 19 positions outside the slice, the
    exact test decides.  Both widths also take the sorted layouts (round 4; the default): every bucket ordered by the byte
    of the next two / four slices, and the pruned scan visits 13 (1, 67) of a bucket's 256 groups -- with <= 4 mismatches
    some exact slice is followed by four positions with at most one mismatch between them (enumerated in
    tests/test_oracle_golden.py).  Stdout and hit lists of the compiled reference, max distances 2, 4 and 6 (6: whole buckets
    on every layout); the one layout a narrow width cannot take is refused."""
    g = Golden(name)
    width = int(name[5:])
    if layout == "host_cold":   # rebuilds signatures from the stream's 16 positions + the bucket's byte: 8-bit slices only
        bad = _open(g.issl, layout)
        with pytest.raises(ca.IsslError):
            bad.upload(0)
        bad.close()
        return
    ix = ca.IsslIndex.open(g.issl) if layout is None else _open(g.issl, layout)
    ix.upload(0)
    is_sorted = layout is None or layout in SORTED
    if layout is None:
        assert ix.get_option("is_sorted") == 1 and ix.get_option("is_compact") == 0 and ix.get_option("cold_sections") == 0
    else:
        _check_layout(ix, layout)
    assert ix.header["slice_width"] == width and ix.header["n_slices"] == 40 // width
    sigs = ca.encode_guides([s.encode() for s in g.guides])
    try:
        for prune in ((0, 1, -1) if is_sorted else (-1,)):
            ix.set_option("prune", prune)
            for key, want in g.expected.items():
                method, thr, dist = key.split("|")
                mit, cfd = ix.score(sigs, int(dist), float(thr), method)
                assert ca.format_scores(sigs, mit, cfd, method) == want, (key, prune)
                st = ix.stats()
                assert st["reference_comparisons"] == ix.count_candidates(sigs)
                if prune == 1 and 0 <= int(dist) <= 5:
                    assert st["pruned"] == (1 if int(dist) <= 2 else 2 if int(dist) <= 4 else 3), (key, st["pruned"])
                if prune == 0 or not is_sorted or int(dist) > 5:
                    assert st["pruned"] == 0 and st["candidates"] == st["reference_comparisons"]
            for thr in g.hit_thresholds():
                assert np.array_equal(ix.dump_hits(sigs, 4, float(thr), "and"), g.hits(thr)), (thr, prune)
    finally:
        ix.close()
    if layout is None:
        odd = ca.IsslIndex.build_from_text(g.sites_txt.read_bytes(), slice_width=5)   # 5-bit slices cut positions in two
        with pytest.raises(ca.IsslError) as e:
            odd.upload(0)
        assert "unsupported index geometry" in str(e.value)
        odd.close()
        exe = pathlib.Path(__file__).resolve().parent.parent / "bin" / "isslScoreOfftargets"
        import subprocess
        r = subprocess.run([str(exe), str(g.issl), str(g.guides_txt), "4", "75", "and"], capture_output=True)
        assert r.returncode == 0 and r.stdout.decode() == g.expected["and|75|4"]


def test_compact_image_with_host_lists_built_on_the_device_and_shared_by_a_node(golden_uniform, tmp_path):
    """The layout of an index beyond the HBM end to end at golden size: built ON the device from signatures + counts with
    the slice lists going to pinned host memory, written back out (the lists come from that host copy) byte-identical to
    the reference-built .issl, scored, and shared by a node of two replicas that adopt the hot image and the ONE host
    buffer."""
    g = golden_uniform
    host = ca.IsslIndex.open(g.issl)
    n = host.header["n_sites"]
    data = g.issl.read_bytes()
    off_sites = 48 + 16 * host.header["n_scores"]
    sigs = np.frombuffer(data, dtype=np.uint64, count=n, offset=off_sites).copy()
    entries0 = np.frombuffer(data, dtype=np.uint64, count=n, offset=off_sites + 8 * n + 8 * 5 * 256)   # slice 0's lists
    occ = np.zeros(n, dtype=np.uint32)
    occ[(entries0 & np.uint64(0xFFFFFFFF)).astype(np.int64)] = (entries0 >> np.uint64(32)).astype(np.uint32)
    host.close()
    ix = ca.IsslIndex.build_on_device(sigs, occ, device=0, n_lines=int(occ.sum()), options={"compact": 1, "host_cold": 1})
    _check_layout(ix, "compact_cold")
    assert ix.cold()[1] >= 40 * n
    out = tmp_path / "roundtrip.issl"
    ix.write(out)
    assert out.read_bytes() == data
    guides = ca.encode_guides([s.encode() for s in g.guides])
    mit, cfd = ix.score(guides, 4, 75.0, "and")
    assert ca.format_scores(guides, mit, cfd, "and") == g.expected["and|75|4"]
    assert np.array_equal(ix.dump_hits(guides, 4, 0.0, "and"), g.hits(0))
    node = ca.IsslNode(ix, devices=[0, 0])
    try:
        mit, cfd = node.score(guides, 4, 75.0, "and")
        assert ca.format_scores(guides, mit, cfd, "and") == g.expected["and|75|4"]
    finally:
        node.close()
        ix.close()


def test_bare_compact_image_from_a_site_table_on_the_device(golden_uniform, tmp_path):
    """issl_index_build_from_device_sites + the image without slice lists: signatures and counts handed over as device
    tensors, bucket sizes counted on the device, nothing pinned; written back out (the lists are made again on the device,
    isslCreateIndex.cpp:218-234) byte-identical to the reference-built .issl; scores and hit lists -- whose list positions
    are recomputed from the stream (hit_terms) -- equal to the reference's; the image travels in a tensor like any other."""
    import torch
    g = golden_uniform
    host = ca.IsslIndex.open(g.issl)
    n = host.header["n_sites"]
    data = g.issl.read_bytes()
    off_sites = 48 + 16 * host.header["n_scores"]
    sigs = np.frombuffer(data, dtype=np.uint64, count=n, offset=off_sites).copy()
    entries0 = np.frombuffer(data, dtype=np.uint64, count=n, offset=off_sites + 8 * n + 8 * 5 * 256)   # slice 0's lists
    occ = np.zeros(n, dtype=np.uint32)
    occ[(entries0 & np.uint64(0xFFFFFFFF)).astype(np.int64)] = (entries0 >> np.uint64(32)).astype(np.uint32)
    assert np.array_equal(host.bucket_sizes(), ca.IsslIndex.build_from_sites(sigs, occ).bucket_sizes())
    sizes = host.bucket_sizes()
    host.close()
    d_sigs = torch.from_numpy(sigs.view(np.int64)).cuda()
    d_occ = torch.from_numpy(occ.view(np.int32)).cuda()
    guides = ca.encode_guides([s.encode() for s in g.guides])
    for options, bare in (({"keep_lists": 0}, 1), (None, 0)):
        ix = ca.IsslIndex.build_from_device_sites(d_sigs, d_occ, int(occ.sum()), device=0, options=options)
        try:
            assert ix.get_option("lists_absent") == bare and ix.get_option("is_sorted") == 1 and ix.cold() == (None, 0)
            assert np.array_equal(ix.bucket_sizes(), sizes)
            out = tmp_path / "roundtrip.issl"
            ix.write(out)
            assert out.read_bytes() == data
            for prune in (0, 1):
                ix.set_option("prune", prune)
                mit, cfd = ix.score(guides, 4, 75.0, "and")
                assert ca.format_scores(guides, mit, cfd, "and") == g.expected["and|75|4"]
                assert np.array_equal(ix.dump_hits(guides, 4, 0.0, "and"), g.hits(0))
            p, nbytes = ix.image()
            raw = torch.empty(nbytes + 256, dtype=torch.uint8, device="cuda:0")
            off = (-raw.data_ptr()) % 256
            twin = ca.IsslIndex.attach_tensor(ix.copy_image_to_tensor(raw[off:off + nbytes]))   # what a broadcast delivers
            mit, cfd = twin.score(guides, 4, 75.0, "and")
            assert ca.format_scores(guides, mit, cfd, "and") == g.expected["and|75|4"]
            assert np.array_equal(twin.dump_hits(guides, 4, 0.0, "and"), g.hits(0))
            twin.close()
        finally:
            ix.close()
    # a signature with bits above the 40 of a 20-mer would spill into the count the sorted layouts keep beside it: refused
    bad = sigs.copy()
    bad[n // 2] |= np.uint64(1) << np.uint64(41)
    with pytest.raises(ca.IsslError) as e:
        ca.IsslIndex.build_from_device_sites(torch.from_numpy(bad.view(np.int64)).cuda(), d_occ, int(occ.sum()), device=0)
    assert "bits above" in str(e.value)


def test_layout_sizes(golden_uniform):
    """HBM bytes per layout, in the order an upload tries them (issl_index_device_bytes needs no device)."""
    size = {}
    for layout in LAYOUTS:
        ix = _open(golden_uniform.issl, layout)
        size[layout] = ix.device_bytes()
        ix.close()
    auto = ca.IsslIndex.open(golden_uniform.issl)
    assert auto.device_bytes() == size["sorted"]
    auto.close()
    n = 8200  # about the sites of the set: the sections differ by whole multiples of it
    assert size["sorted"] > size["compact"] + 50 * n > size["compact_cold"] + 80 * n
    assert size["compact_bare"] == size["compact_cold"]
    assert size["list_esig"] > size["list"] + 30 * n > size["host_cold"] + 60 * n


def test_counts_that_differ_between_a_sites_lists_keep_the_list_order():
    """tests/golden/mixedocc: the five list entries of a site carry different occurrence counts; the reference takes the
    count of the entry it meets first (:348).  The sorted layouts keep ONE count per site, so the upload must notice and
    fall back to a list-order image; asking for a sorted layout explicitly fails."""
    g = Golden("mixedocc")
    sigs = ca.encode_guides([s.encode() for s in g.guides])
    for layout in (None, "list_esig", "list", "host_cold"):
        ix = ca.IsslIndex.open(g.issl) if layout is None else _open(g.issl, layout)
        ix.upload(0)
        assert ix.get_option("is_sorted") == 0
        for key, want in g.expected.items():
            method, thr, dist = key.split("|")
            mit, cfd = ix.score(sigs, int(dist), float(thr), method)
            assert ca.format_scores(sigs, mit, cfd, method) == want, (layout, key)
        for thr in g.hit_thresholds():
            assert np.array_equal(ix.dump_hits(sigs, 4, float(thr), "and"), g.hits(thr)), (layout, thr)
        ix.close()
    for layout in SORTED:
        ix = _open(g.issl, layout)
        with pytest.raises(ca.IsslError) as e:
            ix.upload(0)
        assert "sorted layout" in str(e.value)
        ix.close()
    # keep_lists=0 alone asks for a sorted image too (only such an image can do without its lists): the same refusal,
    # not an upload that starts over for ever
    ix = ca.IsslIndex.open(g.issl).set_option("keep_lists", 0)
    with pytest.raises(ca.IsslError) as e:
        ix.upload(0)
    assert "sorted layout" in str(e.value)
    ix.close()


def _sections(data):
    hdr = np.frombuffer(bytes(data[:48]), dtype=np.uint64)
    n, n_slices, n_scores = int(hdr[0]), int(hdr[4]), int(hdr[5])
    sizes_at = 48 + 16 * n_scores + 8 * n
    sizes = np.frombuffer(bytes(data[sizes_at:sizes_at + 8 * n_slices * 256]), dtype=np.uint64)
    starts = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    return n, n_slices, sizes, starts, sizes_at + 8 * n_slices * 256


def test_lists_that_do_not_ascend_by_site_id_keep_the_list_order(golden_uniform, tmp_path):
    """Every builder appends site ids in ascending order (isslCreateIndex.cpp:218-234), but the format does not ask for
    it and the reference scores a bucket in whatever order its list has (:344).  An index whose lists were shuffled inside
    their buckets is valid; the sorted layouts (scoring order = site id) cannot hold it: list-order image, results equal to
    the oracle's on the same file."""
    data = bytearray(golden_uniform.issl.read_bytes())
    n, n_slices, sizes, starts, entries_at = _sections(data)
    rng = np.random.default_rng(5)
    ent = np.frombuffer(bytes(data[entries_at:entries_at + 8 * n * n_slices]), dtype=np.uint64).copy()
    for b in range(n_slices * 256):
        if sizes[b] > 1:
            seg = ent[starts[b]:starts[b + 1]]
            ent[starts[b]:starts[b + 1]] = seg[rng.permutation(len(seg))]
    data[entries_at:entries_at + 8 * n * n_slices] = ent.tobytes()
    path = tmp_path / "shuffled.issl"
    path.write_bytes(bytes(data))
    oracle = ou.OracleIndex(path)
    guides = ca.encode_guides(golden_uniform.guides)
    for layout in (None, "list", "host_cold"):
        ix = ca.IsslIndex.open(path) if layout is None else _open(path, layout)
        ix.upload(0)
        assert ix.get_option("is_sorted") == 0
        for thr in (0.0, 75.0):
            omit, ocfd, ohits = oracle.score(guides, 4, thr, "and", want_hits=True)
            mit, cfd = ix.score(guides, 4, thr, "and")
            assert np.array_equal(mit.view(np.uint64), omit.view(np.uint64)) and np.array_equal(cfd.view(np.uint64), ocfd.view(np.uint64))
            assert np.array_equal(ix.dump_hits(guides, 4, thr, "and"), ohits)
        ix.close()
    oracle.close()
    ix = _open(path, "sorted")
    with pytest.raises(ca.IsslError):
        ix.upload(0)
    ix.close()


@pytest.mark.parametrize("layout", [None] + list(LAYOUTS))
def test_a_site_in_a_foreign_bucket_or_listed_twice_is_a_format_error(golden_uniform, layout):
    """A slice list that holds a site whose signature does not select it, or the same site twice: no builder writes
    either.  The reference would still run (it compares whole signatures and keeps a seen-bitmap, :376,:385-390), but
    the scan compares only the 16 positions outside the slice and the first-matching-slice rule stands in for the bitmap
    on the premise that every slice lists every site once, in its own bucket -- so the upload refuses the file in EVERY
    layout (ISSL_E_FORMAT) instead of scoring it differently from the reference in some."""
    base = golden_uniform.issl.read_bytes()
    n, n_slices, sizes, starts, entries_at = _sections(base)
    full = [b for b in range(256) if sizes[b] > 1]
    # (a) swap the first entries of two buckets of slice 0; (b) overwrite an entry with its neighbour (a duplicate)
    swapped = bytearray(base)
    a, b = entries_at + 8 * int(starts[full[0]]), entries_at + 8 * int(starts[full[-1]])
    swapped[a:a + 8], swapped[b:b + 8] = swapped[b:b + 8], swapped[a:a + 8]
    twice = bytearray(base)
    at = entries_at + 8 * int(starts[256 + full[3]])
    twice[at + 8:at + 16] = twice[at:at + 8]
    for data in (swapped, twice):
        ix = ca.IsslIndex.from_bytes(bytes(data)) if layout is None else _open(bytes(data), layout)
        with pytest.raises(ca.IsslError) as e:
            ix.upload(0)
        assert "Error reading index" in str(e.value)
        ix.close()
    intact = ca.IsslIndex.open(golden_uniform.issl).upload(0)
    assert intact.get_option("is_sorted") == 1
    intact.close()


@pytest.mark.parametrize("name", ["uniform", "clustered", "bigocc"])
@pytest.mark.parametrize("layout", [None, "compact", "compact_bare", "compact_cold", "list_esig", "host_cold"])
def test_upload_through_the_pinned_ring(name, layout):
    """A file-mapped index reaches the device through a ring of pinned chunks filled by pread (FileUploader: the link's rate
    instead of a fifth of it), its sections queued one behind the other and waited for one by one: the slice lists land beside the
    kernels that order / pack the slice before.  Sections below 64
    MiB take the plain copy, so the golden indexes never see the ring by themselves: here they do -- slots of 64 KiB (and of 4
    KiB: a chunk then ends inside every few hundred entries), no minimum size, the 9.7 MB index in ~150 (~2400) chunks that end
    inside sections -- and every reference stdout and hit list must come out as through the plain copy."""
    g = Golden(name)
    sigs = ca.encode_guides([s.encode() for s in g.guides])
    for chunk_kib, readers in ((64, 8), (4, 3)):
        ix = ca.IsslIndex.open(g.issl)
        if layout:
            for key, value in LAYOUTS[layout].items():
                ix.set_option(key, value)
        ix.set_option("upload_chunk_kib", chunk_kib).set_option("upload_ring_min_kib", 0).set_option("upload_threads", readers)
        ix.upload(0)
        try:
            if layout:
                _check_layout(ix, layout)
            for key, want in g.expected.items():
                method, thr, dist = key.split("|")
                mit, cfd = ix.score(sigs, int(dist), float(thr), method)
                assert ca.format_scores(sigs, mit, cfd, method) == want, (chunk_kib, key)
            for thr in g.hit_thresholds():
                assert np.array_equal(ix.dump_hits(sigs, 4, float(thr), "and"), g.hits(thr)), (chunk_kib, thr)
        finally:
            ix.close()
