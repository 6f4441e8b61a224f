import json
import os
import pathlib
import subprocess
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
GOLD = ROOT / "tests" / "golden"


SCALE_NOTES = []  # tests/test_scale.py: the sizes its points actually ran at, printed behind the test summary


def pytest_terminal_summary(terminalreporter):
    for note in SCALE_NOTES:
        terminalreporter.write_line("scale point ran: " + note)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_first():
    """On a GPU box let torch create its HIP context BEFORE libissl_hip.so makes its first HIP call: both share
    one runtime (crackling_amd/_lib.py), and torch's lazy CUDA init was seen to hang when it ran after a long
    series of direct HIP calls in the same process.  bench.py follows the same order."""
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
        torch.zeros(1, device="cuda:0")
        torch.cuda.synchronize()


def _ensure_built():
    """Product library + oracle (checker) must exist; build them if the tree is fresh.  Runs when conftest is imported,
    i.e. before the test modules (which import crackling_amd at their top) are collected."""
    if not (ROOT / "crackling_amd" / "libissl_hip.so").exists() or not (ROOT / "bin" / "isslScoreOfftargets").exists():
        subprocess.run(["make", "-C", str(ROOT)], check=True, capture_output=True)
    if not (ROOT / "oracle" / "_build" / "liboracle.so").exists():
        subprocess.run(["make", "-C", str(ROOT / "oracle"), "all"], check=True, capture_output=True)


_ensure_built()


GOLDEN_SETS = ["uniform", "clustered", "edge"]


class Golden:
    def __init__(self, name):
        self.name = name
        self.dir = GOLD / name
        self.issl = self.dir / "index.issl"
        self.sites_txt = self.dir / "sites.txt"
        if not self.issl.exists():
            # sets whose index is too large to commit (slice widths 4 and 2: 7 / 17 MB of score table) store the SHA-256
            # of the reference-built file; the index is rebuilt here by this repository's host builder (CPU code) and
            # must have that digest before anything is tested against it
            import hashlib
            import tempfile
            import crackling_amd as ca
            width = int(name.replace("width", ""))
            cache = pathlib.Path(tempfile.gettempdir()) / f"issl_golden_{os.getuid()}"
            cache.mkdir(exist_ok=True)
            self.issl = cache / f"{name}.issl"
            want = (self.dir / "index.sha256").read_text().strip()
            if not self.issl.exists() or hashlib.sha256(self.issl.read_bytes()).hexdigest() != want:
                ix = ca.IsslIndex.build_from_text(self.sites_txt.read_bytes(), slice_width=width)
                ix.write(self.issl)
                ix.close()
            assert hashlib.sha256(self.issl.read_bytes()).hexdigest() == want, f"{name}: builder bytes differ from the reference's"
        self.guides_txt = self.dir / "guides.txt"
        self.expected = json.loads((self.dir / "expected.json").read_text())
        self.guides = self.guides_txt.read_text().splitlines()

    def hits(self, thr):
        import numpy as np
        p = self.dir / f"hits_and_{thr}.tsv"
        rows = [list(map(int, l.split("\t"))) for l in p.read_text().splitlines()]
        return np.array(rows, dtype=np.uint32).reshape(-1, 6)

    def hit_thresholds(self):
        return sorted(int(p.stem.split("_")[-1]) for p in self.dir.glob("hits_and_*.tsv"))


@pytest.fixture(scope="session", params=GOLDEN_SETS)
def golden(request):
    return Golden(request.param)


@pytest.fixture(scope="session")
def golden_uniform():
    return Golden("uniform")


@pytest.fixture(scope="session")
def golden_oddtable():
    """The clustered index with a rewritten local-MIT table: duplicate masks (first pair wins), masks with odd bits,
    patterns missing from the table -- outputs of the compiled reference (oracle/make_golden_oddtable.py)."""
    return Golden("oddtable")


# Golden sets that exercise branches no builder-made index reaches (oracle/make_golden_extra.py; all of them outputs
# of the compiled reference): occurrence counts at the saturation points of the image's 8- and 24-bit copies, counts
# that differ between the five lists of a site, a score table with negative, NaN and +inf values under guides with
# thousands of hits.
EXTRA_SETS = ["bigocc", "mixedocc", "signedtable"]


@pytest.fixture(scope="session", params=EXTRA_SETS)
def golden_extra(request):
    return Golden(request.param)


# Slice widths 4 and 2 (10 / 20 slices): reference outputs of oracle/make_golden_widths.py; the index itself is rebuilt
# by the host builder and checked against the digest of the reference-built file.
@pytest.fixture(scope="session", params=["width4", "width2"])
def golden_width(request):
    return Golden(request.param)
