"""BASELINE configs[4] at its real size: a 3 G-line index on ONE MI355X (a module of its own, behind test_scale.py: the
point needs the whole HBM, and that module's shared index is released when it ends)."""
import json
import os

import pytest

from test_scale import ScalePoint, _score_and_check, _hit_lists_match

BEYOND_HBM_LINES = int(os.environ.get("ISSL_BEYOND_HBM_LINES", 3_000_000_000))


@pytest.mark.gpu
def test_index_beyond_the_hbm_at_scale(tmp_path):
    """BASELINE configs[4] itself: a 3 G-line index (2.92 G distinct sites; 144 GB as an .issl, 444 GB in the default
    layout) on ONE MI355X.  The layout such an index gets -- compact sorted image without slice lists: scan stream, site
    ids per stream position, site table and counts = 52 B/site, 152 GB, nothing in host memory -- forced; the site table
    is drawn straight into device memory (the host never holds anything of the index's size) and the index built from
    there.  Pruned scan against the scan of whole buckets on all 100 000 guides, a sample and its hit lists against the
    oracle (the sample's neighbourhoods by brute force over the device-resident table).  Like every scale point it runs
    at its size or fails (skipped on a GPU that is no MI355X); the size is printed behind the test summary."""
    sp = ScalePoint(BEYOND_HBM_LINES, int(os.environ.get("ISSL_SCALE_GUIDES", 100_000)), options={"keep_lists": 0},
                    n_check=int(os.environ.get("ISSL_SCALE_CHECK", 32)), what="configs[4] (compact image without slice lists)",
                    device_synth=True)
    try:
        assert sp.ix.get_option("is_compact") == 1 and sp.ix.get_option("lists_absent") == 1
        assert sp.ix.cold() == (None, 0) and sp.ix.device_bytes() < 60 * len(sp.sigs) + (64 << 20)
        # image + temporaries at the high-water mark of the construction: 52 + 8.5 B per site (asserted in ScalePoint against the
        # image's own size): what makes the format's 2^32 - 1 sites (258 GB) fit the 288 GB of one MI355X
        assert sp.build_peak_bytes <= 61 * len(sp.sigs) + (1 << 30), (sp.build_peak_bytes, len(sp.sigs))
        print(f"construction of the {len(sp.sigs)}-site image: {sp.build_peak_bytes / 1e9:.1f} GB of HBM at its high-water mark "
              f"= {sp.build_peak_bytes / len(sp.sigs):.1f} B per site (image {sp.ix.device_bytes() / 1e9:.1f} GB)", flush=True)
        summary = _score_and_check(sp, tmp_path, "tests/test_scale_beyond_hbm.py::test_index_beyond_the_hbm_at_scale")
        _hit_lists_match(sp, tmp_path)
        if os.environ.get("ISSL_BEYOND_HBM_JSON"):
            json.dump(summary, open(os.environ["ISSL_BEYOND_HBM_JSON"], "w"), indent=1)
    finally:
        sp.ix.close()


