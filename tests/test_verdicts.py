"""Caller-side thresholding fused into the scorer: issl_verdicts / ISSL_VERDICTS.

Pinned by tests/golden/verdicts/: the verdicts the reference caller's OWN loop (Crackling.py:780-835, lifted out of the
reference source at run time by oracle/make_golden_verdicts.py) produced on the reference scorer's stdout and on
borderline score sets.  Both the product (issl_verdicts) and the restatement used elsewhere in this file
(oracle/caller_thresholds.py) are checked against them."""
import json
import os
import pathlib
import subprocess
import sys

import numpy as np
import pytest

import crackling_amd as ca

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
from caller_thresholds import caller_verdicts  # noqa: E402

METHOD_STRINGS = ["and", "or", "avg", "mit", "cfd", "AND", " avg", "Mit", "xyz", ""]


def _check(sigs, mit, cfd, thr, method):
    text = ca.format_scores(sigs, mit, cfd, method)          # the scorer's stdout for these scores
    seqs = ca.decode_guides(sigs)
    want = caller_verdicts(text, seqs, thr, method)
    got = ca.verdicts(mit, cfd, thr, method)
    assert set(np.unique(got)) <= {0, 1, 255}
    if not want:
        assert (got == 255).all(), method
        return 0
    assert [int(v) for v in got] == [want[s] for s in seqs], (thr, method)
    return int((got == 0).sum())


GOLD = ROOT / "tests" / "golden"
CHAR = {1: "1", 0: "0", 255: "-"}


def test_golden_verdicts_of_the_reference_caller_on_reference_stdout():
    """tests/golden/verdicts/reference_stdout.json: every stdout of the compiled reference scorer, fed to the lifted
    caller loop under the configured method string and three other spellings of it."""
    rec = json.loads((GOLD / "verdicts" / "reference_stdout.json").read_text())
    assert rec["source_lines"] == [780, 835]
    expected = {name: json.loads((GOLD / name / "expected.json").read_text()) for name in ("uniform", "clustered", "edge")}
    spellings = set()
    for case in rec["cases"]:
        text = expected[case["golden"]][case["key"]]
        thr, method = case["threshold"], case["config_method"]
        seqs = [line.split("\t")[0] for line in text.splitlines()]
        # the restatement, from the text
        want = caller_verdicts(text, seqs, thr, method)
        assert "".join(CHAR[want[s]] if s in want else "-" for s in seqs) == case["verdicts"], (case["golden"], case["key"], method)
        # the product, from the scores.  issl_verdicts takes ONE method string -- the one the scorer was started with --
        # so it is comparable where the caller's configured string is that string (a sequence that occurs twice is keyed
        # once by the caller: its last line wins, which parse_scorer_output mirrors)
        printed = case["key"].split("|")[0]
        if method == printed:
            parsed = ca.parse_scorer_output(text)
            got = ca.verdicts([parsed[s]["mit"] for s in seqs], [parsed[s]["cfd"] for s in seqs], float(thr), method)
            assert "".join(CHAR[int(v)] for v in got) == case["verdicts"], (case["golden"], case["key"], method)
        spellings.add(method == printed)
    assert spellings == {True, False} and len(rec["cases"]) >= 700


def test_golden_verdicts_of_the_reference_caller_on_borderline_scores():
    """tests/golden/verdicts/borderline.json: score pairs whose 6-decimal text lands on either side of the threshold;
    scorer and caller configured with the same string (as Crackling does), incl. strings the scorer does not know."""
    rec = json.loads((GOLD / "verdicts" / "borderline.json").read_text())
    checked = 0
    for case in rec["cases"]:
        thr = case["threshold"]
        mit = np.array([float.fromhex(h) for h in case["mit_hex"]])
        cfd = np.array([float.fromhex(h) for h in case["cfd_hex"]])
        sigs = ca.encode_guides(case["seqs"])
        for method, want in case["verdicts_by_config_method"].items():
            got = ca.verdicts(mit, cfd, float(thr), method)
            assert "".join(CHAR[int(v)] for v in got) == want, (thr, method)
            text = ca.format_scores(sigs, mit, cfd, method)
            rest = caller_verdicts(text, case["seqs"], thr, method)
            assert "".join(CHAR[rest[s]] if s in rest else "-" for s in case["seqs"]) == want, (thr, method)
            checked += len(want)
    assert checked >= 10_000


def test_verdicts_follow_the_caller_on_random_and_borderline_scores():
    rng = np.random.default_rng(5)
    n = 4000
    sigs = np.unique(rng.integers(0, 1 << 40, size=n, dtype=np.uint64))    # distinct: the caller keys by sequence
    n = len(sigs)
    for thr in (75.0, 0.0, 50.0, 99.9999995, 75.0000004):
        mit = rng.uniform(0, 100, n)
        cfd = rng.uniform(0, 100, n)
        # scores whose 6-decimal text lands on the other side of the threshold than the double itself
        edge = thr + rng.choice([-6e-7, -5e-7, -4e-7, -1e-7, 0.0, 1e-7, 4e-7, 5e-7, 6e-7, 1.1e-6, -1.1e-6], n // 2)
        mit[: n // 2] = edge
        cfd[n // 4: n // 4 + n // 2] = np.roll(edge, 7)
        mit[-5:] = [100.0, 0.0, thr, np.nextafter(thr, 0), np.nextafter(thr, 200)]
        rejected = [_check(sigs, mit, cfd, thr, m) for m in METHOD_STRINGS]
        assert thr == 0.0 or rejected[0] > 0
    with pytest.raises(ValueError):
        ca.verdicts(np.zeros(3), np.zeros(4), 75, "and")


def test_verdicts_on_reference_stdout(golden):
    """Scores as the compiled reference printed them (tests/golden/*/expected.json)."""
    seen = 0
    for key, text in golden.expected.items():
        method, thr, _ = key.split("|")
        parsed = ca.parse_scorer_output(text)
        seqs = list(parsed)
        want = caller_verdicts(text, seqs, thr, method)
        got = ca.verdicts([parsed[s]["mit"] for s in seqs], [parsed[s]["cfd"] for s in seqs], float(thr), method)
        if method == "xyz":
            assert not want and (got == 255).all()
            continue
        assert [int(v) for v in got] == [want[s] for s in seqs], key
        seen += 1
    assert seen >= 5


def _verdict_file(path):
    return dict(line.split("\t") for line in pathlib.Path(path).read_text().splitlines())


@pytest.mark.gpu
def test_cli_writes_verdict_file_without_touching_stdout(golden, tmp_path):
    exe = str(ROOT / "bin" / "isslScoreOfftargets")
    for key in ("and|75|4", "or|75|4", "avg|75|4", "mit|75|4", "cfd|50|4", "xyz|0|4"):
        if key not in golden.expected:
            continue
        method, thr, dist = key.split("|")
        vf = tmp_path / f"verdicts_{method}.tsv"
        r = subprocess.run([exe, str(golden.issl), str(golden.guides_txt), dist, thr, method], capture_output=True,
                           env=dict(os.environ, ISSL_VERDICTS=str(vf)))
        assert r.returncode == 0, r.stderr.decode()
        assert r.stdout.decode() == golden.expected[key], key
        seqs = [line.split("\t")[0] for line in golden.expected[key].splitlines()]
        want = caller_verdicts(golden.expected[key], seqs, thr, method)
        assert _verdict_file(vf) == {s: str(v) for s, v in want.items()}, key
    r = subprocess.run([exe, str(golden.issl), str(golden.guides_txt), "4", "75", "and"], capture_output=True,
                       env=dict(os.environ, ISSL_VERDICTS=str(tmp_path / "no_such_dir" / "v.tsv")))
    assert r.returncode == 1 and r.stdout == b"" and b"cannot write verdict file" in r.stderr


@pytest.mark.gpu
def test_resident_server_writes_verdict_file(golden_uniform, tmp_path):
    import time
    exe = str(ROOT / "bin" / "isslScoreOfftargets")
    sock = str(tmp_path / "v.sock")
    server = subprocess.Popen([exe, "--serve", sock], stderr=subprocess.PIPE)
    try:
        for _ in range(100):
            if os.path.exists(sock):
                break
            time.sleep(0.05)
        assert os.path.exists(sock)
        key = "and|75|4"
        r = subprocess.run([exe, str(golden_uniform.issl), str(golden_uniform.guides_txt), "4", "75", "and"],
                           capture_output=True, cwd=str(tmp_path),
                           env=dict(os.environ, ISSL_SERVER=sock, ISSL_VERDICTS="relative_verdicts.tsv"))
        assert r.returncode == 0, r.stderr.decode()
        assert r.stdout.decode() == golden_uniform.expected[key]
        seqs = [line.split("\t")[0] for line in golden_uniform.expected[key].splitlines()]
        want = caller_verdicts(golden_uniform.expected[key], seqs, 75, "and")
        assert _verdict_file(tmp_path / "relative_verdicts.tsv") == {s: str(v) for s, v in want.items()}
    finally:
        subprocess.run([exe, "--stop", sock], capture_output=True)
        server.wait(timeout=30)
