"""CPU: host logic of the product library and its C ABI (no compute calls: there is no GPU here)."""
import ctypes as C
import os
import pathlib
import re
import subprocess

import numpy as np
import pytest

import crackling_amd as ca
from crackling_amd import _lib
import oracle_util as ou
from synth import random_sites, random_guides, sigs_to_text, text_order_key

ROOT = pathlib.Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    header = (ROOT / "include" / "issl_hip.h").read_text()
    declared = set(re.findall(r"\b(issl_[a-z_0-9]+)\s*\(", header))
    assert len(declared) >= 24
    lib = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/issl_hip.h but not exported"
    assert declared == set(_lib.EXPORTS)
    assert lib.issl_abi_version() == 6


def test_builder_bytes_match_reference_golden(golden, tmp_path):
    ix = ca.IsslIndex.build_from_text(golden.sites_txt.read_bytes())
    out = tmp_path / "x.issl"
    ix.write(out)
    assert out.read_bytes() == golden.issl.read_bytes()


def test_builder_cli_matches_reference_golden(golden, tmp_path):
    out = tmp_path / "cli.issl"
    subprocess.run([str(ROOT / "bin" / "isslCreateIndex"), str(golden.sites_txt), "20", "8", str(out)], check=True,
                   capture_output=True)
    assert out.read_bytes() == golden.issl.read_bytes()


def test_builder_from_sites_equals_from_text_and_oracle():
    sigs, occ = random_sites(50000, seed=5)
    text = sigs_to_text(sigs, occ)
    a = ca.IsslIndex.build_from_sites(sigs, occ)
    b = ca.IsslIndex.build_from_text(text)
    assert a.header == b.header
    assert a.header["n_lines"] == 50000 and a.header["n_sites"] == len(sigs)
    assert np.array_equal(a.bucket_sizes(), b.bucket_sizes())
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        pa, pb = os.path.join(tmp, "a"), os.path.join(tmp, "b")
        a.write(pa); b.write(pb)
        da = open(pa, "rb").read()
        assert da == open(pb, "rb").read()
        assert da == ou.build_issl(text)


def test_builder_other_slice_widths_match_oracle():
    sigs, occ = random_sites(3000, seed=6)
    text = sigs_to_text(sigs, occ)
    import tempfile
    for w in (4, 5):
        ix = ca.IsslIndex.build_from_text(text, slice_width=w)
        with tempfile.TemporaryDirectory() as tmp:
            p = os.path.join(tmp, "a")
            ix.write(p)
            assert open(p, "rb").read() == ou.build_issl(text, slice_width=w), w


def test_builder_rejects_bad_geometry():
    sigs, occ = random_sites(100, seed=7)
    with pytest.raises(ca.IsslError):
        ca.IsslIndex.build_from_sites(sigs, occ, slice_width=9)  # reference truncates slice values to 8 bits
    with pytest.raises(ca.IsslError):
        ca.IsslIndex.build_from_sites(sigs, occ, slice_width=1)


def test_encode_decode_roundtrip_and_non_acgt():
    seqs = ["ACGTACGTACGTACGTACGT", "TTTTTTTTTTTTTTTTTTTT", "ACGTNCGTACGTACGTACGT", "acgtacgtacgtacgtacgt"]
    sig = ca.encode_guides(seqs)
    assert np.array_equal(sig, ou.encode(seqs))
    dec = ca.decode_guides(sig)
    assert dec[0] == seqs[0] and dec[1] == seqs[1]
    assert dec[2] == "ACGTACGTACGTACGTACGT" and dec[3] == "A" * 20  # other bytes pack as 'A' (:99-102)
    assert int(sig[1]) == (1 << 40) - 1


def test_open_validates_like_the_reference_or_stricter(golden_uniform, tmp_path):
    data = golden_uniform.issl.read_bytes()
    ix = ca.IsslIndex.from_bytes(data)
    hd = ix.header
    assert hd["seq_len"] == 20 and hd["slice_width"] == 8 and hd["n_slices"] == 5 and hd["n_scores"] == 6195
    assert int(ix.bucket_sizes().sum()) == 5 * hd["n_sites"]
    for cut, msg in [(20, "header invalid"), (48 + 16 * 100, "scores"), (48 + 16 * 6195 + 80, "off-target"),
                     (48 + 16 * 6195 + 8 * hd["n_sites"] + 64, "slice list sizes"), (len(data) - 8, "slice contents")]:
        with pytest.raises(ca.IsslError) as e:
            ca.IsslIndex.from_bytes(data[:cut])
        assert msg in str(e.value), (cut, str(e.value))
    with pytest.raises(ca.IsslError):
        ca.IsslIndex.open(tmp_path / "does-not-exist.issl")
    bad = bytearray(data)
    bad[8:16] = (99).to_bytes(8, "little")  # seq_len 99
    with pytest.raises(ca.IsslError):
        ca.IsslIndex.from_bytes(bytes(bad))


def test_query_file_rules(tmp_path):
    lib = _lib.lib
    p = tmp_path / "q.txt"
    p.write_text("ACGTACGTACGTACGTACGT\nTTTTTTTTTTTTTTTTTTTT\n")
    out = C.c_void_p(); n = C.c_size_t()
    assert lib.issl_read_query_file(os.fsencode(p), 20, C.byref(out), C.byref(n)) == 0
    assert n.value == 2
    arr = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint64)), shape=(2,)).copy()
    lib.issl_free(out)
    assert np.array_equal(arr, ca.encode_guides(["ACGTACGTACGTACGTACGT", "T" * 20]))
    p.write_text("ACGTACGTACGTACGTACGT\r\n")  # CRLF: not a multiple of 21 (:277-282)
    assert lib.issl_read_query_file(os.fsencode(p), 20, C.byref(out), C.byref(n)) != 0
    assert "multiple of the expected line length" in lib.issl_last_error().decode()
    p.write_text("")  # empty: "Failed to read in query file." (:290-293)
    assert lib.issl_read_query_file(os.fsencode(p), 20, C.byref(out), C.byref(n)) != 0


def test_count_candidates_matches_bucket_arithmetic(golden_uniform):
    ix = ca.IsslIndex.open(golden_uniform.issl)
    sigs = ca.encode_guides([g for g in golden_uniform.guides])
    sizes = ix.bucket_sizes().reshape(5, 256)
    want = sum(int(sizes[s, (int(g) >> (8 * s)) & 0xFF]) for g in sigs for s in range(5))
    assert ix.count_candidates(sigs) == want


def test_scoring_without_device_fails_loudly(golden_uniform):
    """No silent CPU fallback: on a box without a GPU every compute entry point reports an error."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ix = ca.IsslIndex.open(golden_uniform.issl)
    with pytest.raises(ca.IsslError) as e:
        ix.upload(0)
    assert e.value.code == -5
    with pytest.raises(ca.IsslError) as e:
        ix.score(golden_uniform.guides[:3])
    assert e.value.code == -7  # no device image
    r = subprocess.run([str(ROOT / "bin" / "isslScoreOfftargets"), str(golden_uniform.issl),
                        str(golden_uniform.guides_txt), "4", "75", "and"], capture_output=True)
    assert r.returncode == 1 and r.stdout == b"" and b"no HIP device" in r.stderr


def test_cli_argument_and_file_errors(golden_uniform, tmp_path):
    exe = str(ROOT / "bin" / "isslScoreOfftargets")
    r = subprocess.run([exe, str(golden_uniform.issl), str(golden_uniform.guides_txt), "4"], capture_output=True)
    assert r.returncode == 1 and b"Usage" in r.stderr and r.stdout == b""
    r = subprocess.run([exe, str(tmp_path / "nope.issl"), str(golden_uniform.guides_txt), "4", "75", "and"],
                       capture_output=True)
    assert r.returncode == 1 and r.stdout == b""
    bad = tmp_path / "bad.txt"
    bad.write_text("ACGT\n")
    r = subprocess.run([exe, str(golden_uniform.issl), str(bad), "4", "75", "and"], capture_output=True)
    assert r.returncode == 1 and r.stdout == b"" and b"multiple of the expected line length" in r.stderr


def test_scan_word_packing_host_model():
    """Host model of the device packing (issl_kernels.hip scan_word): the fold of two scan words counts exactly
    the mismatches outside the bucket's own slice."""
    rng = np.random.default_rng(3)

    def gather_even16(x):
        x &= 0x55555555
        x = (x | (x >> 1)) & 0x33333333
        x = (x | (x >> 2)) & 0x0F0F0F0F
        x = (x | (x >> 4)) & 0x00FF00FF
        x = (x | (x >> 8)) & 0x0000FFFF
        return x

    def scan_word(sig, s):
        sh = 8 * s
        rem = ((sig & ((1 << sh) - 1)) | ((sig >> (sh + 8)) << sh)) & 0xFFFFFFFF
        return gather_even16(rem) | (gather_even16(rem >> 1) << 16)

    for _ in range(2000):
        a = int(rng.integers(0, 1 << 40)); b = int(rng.integers(0, 1 << 40)); s = int(rng.integers(0, 5))
        b = (b & ~(0xFF << (8 * s))) | (a & (0xFF << (8 * s)))  # same bucket
        x = a ^ b
        mm = ((x & 0xAAAAAAAAAAAAAAAA) >> 1) | (x & 0x5555555555555555)
        y = scan_word(a, s) ^ scan_word(b, s)
        assert bin((y | (y >> 16)) & 0xFFFF).count("1") == bin(mm).count("1")


def test_image_layouts_and_their_sizes(golden_uniform):
    """The layouts an upload tries, in order, and what each keeps in HBM per site (issl_index_device_bytes needs no
    device): the sections that differ between them (at scale: sorted 152 B per site, compact 92, compact with the slice
    lists in host memory 52; list order 108 / 68, with site table and lists in host memory 25)."""
    n = 8000
    sizes = {}
    for name, opts in {"sorted": {"sorted_layout": 1, "compact": 0, "host_cold": 0}, "compact": {"compact": 1, "host_cold": 0},
                       "compact_cold": {"compact": 1, "host_cold": 1}, "list_esig": {"sorted_layout": 0, "inline_sigs": 1, "host_cold": 0},
                       "list": {"sorted_layout": 0, "inline_sigs": 0, "host_cold": 0}, "host_cold": {"sorted_layout": 0, "host_cold": 1}}.items():
        ix = ca.IsslIndex.open(golden_uniform.issl)
        for k, v in opts.items():
            ix.set_option(k, v)
        sizes[name] = ix.device_bytes()
        ix.close()
    auto = ca.IsslIndex.open(golden_uniform.issl)
    assert auto.device_bytes() == sizes["sorted"]          # what an upload tries first
    auto.close()
    ref = ca.IsslIndex.open(golden_uniform.issl)
    tiles = int(((ref.bucket_sizes() + 2047) // 2048).sum())     # every bucket is padded to whole tiles of 2048 candidates
    ref.close()
    slack = 16 * 256                                             # (sections are 256-byte aligned)
    assert abs((sizes["sorted"] - sizes["compact"]) - 12 * 2048 * tiles) <= slack         # 16-byte records vs 4-byte ids per stream position
    assert abs((sizes["compact"] - sizes["compact_cold"]) - 40 * n) <= slack               # the slice lists
    assert abs((sizes["list_esig"] - sizes["list"]) - 40 * n) <= slack                     # the in-list signatures
    assert abs((sizes["list"] - sizes["host_cold"]) - (48 - 5) * n) <= slack               # site table + lists out, one byte per entry in
    assert abs((sizes["compact_cold"] - sizes["host_cold"]) - (4 * 2048 * tiles + 12 * n - 5 * n)) <= slack + 4 * 1280 * 257 + 256
    bad = ca.IsslIndex.open(golden_uniform.issl)
    bad.set_option("sorted_layout", 0).set_option("compact", 1)
    bad.set_option("inline_sigs", 1)
    with pytest.raises(ca.IsslError):
        ca.IsslIndex.open(golden_uniform.issl).set_option("compact", 2)
    bad.close()


def _libc_f(values):
    """glibc's own printf("%f") of every value (the reference prints its scores with it, isslScoreOfftargets.cpp:517-525)."""
    libc = C.CDLL(None)
    libc.snprintf.restype = C.c_int
    buf = C.create_string_buffer(512)
    out = []
    for v in values:
        k = libc.snprintf(buf, C.c_size_t(512), b"%f", C.c_double(float(v)))
        out.append(buf.raw[:k])
    return out


def test_own_percent_f_is_glibcs_digit_for_digit(golden):
    """issl_format_scores prints "%f" with a formatter of its own (a million lines through printf are as long as their
    scoring).  It must print what glibc prints: the exact binary value rounded to six decimals, ties to even."""
    rng = np.random.default_rng(11)
    ties = np.arange(1, 4001, 2, dtype=np.float64) / 128.0          # k/128 with k odd: x.xxxxxx5 exactly -- the ties
    near = np.concatenate([ties * (1 + 2.0 ** -52), ties * (1 - 2.0 ** -53), np.nextafter(ties, 0), np.nextafter(ties, 1e9)])
    scores = 10000.0 / (100.0 + np.concatenate([rng.exponential(30.0, 200000), rng.exponential(1e6, 20000), [0.0]]))  # :505-506
    special = np.array([0.0, -0.0, 100.0, 99.9999995, 99.99999949999999, 5e-7, 4.9999999999999996e-07, 5.000000000000001e-07,
                        1.5e-6, 2.5e-6, 2.4999999999999998e-06, 1e-300, 5e-324, 2.0 ** 39, 2.0 ** 40 - 0.5, 2.0 ** 40, 1e22, 1.7976931348623157e308,
                        np.inf, -np.inf, np.nan, -1.0, -1e-9, 0.1, 0.3, 1 / 3, 123456.7890125, 0.0078125, 0.0234375, 1e15 + 0.3])
    bits = rng.integers(0, 2 ** 63, size=100000, dtype=np.int64).view(np.float64)   # every exponent, sign bit clear
    vals = np.concatenate([ties, near, scores, special, bits, -bits[:2000]])
    sigs = rng.integers(0, 1 << 40, size=len(vals), dtype=np.uint64)
    want_f = _libc_f(vals)
    for i in rng.integers(0, len(vals), size=3000):   # Python's % is correctly rounded too: a second witness
        if np.isfinite(vals[i]):
            assert want_f[i] == (b"%f" % vals[i]), vals[i]
    seqs = [s.encode() for s in ca.decode_guides(sigs)]
    for threads in (1, 0, 5):
        got = ca.format_scores_native(sigs, vals, vals[::-1].copy(), "and", threads=threads)
        want = b"".join(s + b"\t" + m + b"\t" + c + b"\n" for s, m, c in zip(seqs, want_f, want_f[::-1]))
        assert got == want, threads
    # the method decides which columns are printed (:517-525), as in the reference's golden stdout
    small = slice(0, 50)
    for method, cols in (("mit", (1, 0)), ("cfd", (0, 1)), ("or", (1, 1)), ("avg", (1, 1)), ("bogus", (0, 0))):
        got = ca.format_scores_native(sigs[small], vals[small], vals[small], method)
        want = b"".join(s + b"\t" + (m if cols[0] else b"-1") + b"\t" + (m if cols[1] else b"-1") + b"\n" for s, m in zip(seqs[small], want_f[small]))
        assert got == want, method
    assert ca.format_scores_native(sigs[:0], vals[:0], vals[:0], "and") == b""
    # and the Python restatement the parity tests compare stdout with says the same
    assert ca.format_scores_native(sigs[:5000], scores[:5000], scores[5000:10000], "and").decode() == ca.format_scores(sigs[:5000], scores[:5000], scores[5000:10000], "and")


def test_query_file_reader_threads_and_rules(tmp_path):
    """issl_read_query_file (isslScoreOfftargets.cpp:275-305): lines of seq_len + 1 bytes, any other byte packs as 'A'; large
    files are read by several threads -- same guides, same order."""
    rng = np.random.default_rng(12)
    sigs = rng.integers(0, 1 << 40, size=300_001, dtype=np.uint64)
    text = "".join(s + "\n" for s in ca.decode_guides(sigs))
    path = tmp_path / "q.txt"
    path.write_text(text)
    out, n = C.c_void_p(), C.c_size_t()
    _lib.check(_lib.lib.issl_read_query_file(os.fsencode(path), 20, C.byref(out), C.byref(n)))
    got = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint64)), shape=(n.value,)).copy()
    _lib.lib.issl_free(out)
    assert np.array_equal(got, sigs)
    for bad, code in ((text[:-3], "multiple of the expected line length"), ("", "Failed to read in query file")):
        path.write_text(bad)
        assert _lib.lib.issl_read_query_file(os.fsencode(path), 20, C.byref(out), C.byref(n)) != 0
        assert code in _lib.lib.issl_last_error().decode()
    assert _lib.lib.issl_read_query_file(os.fsencode(tmp_path / "absent"), 20, C.byref(out), C.byref(n)) != 0


def test_cli_without_its_library_fails_loudly(golden, tmp_path):
    """bin/isslScoreOfftargets is host code that loads libissl_hip.so on demand (never when a resident server answers).  Away from
    the library -- a copy of the executable alone -- it says what it tried and exits 1 with nothing on stdout; ISSL_LIBRARY
    points it at the library again (and then it fails for want of a GPU, here)."""
    import shutil
    exe = tmp_path / "isslScoreOfftargets"
    shutil.copy(ROOT / "bin" / "isslScoreOfftargets", exe)
    args = [str(exe), str(golden.issl), str(golden.guides_txt), "4", "75", "and"]
    env = {k: v for k, v in os.environ.items() if k not in ("LD_LIBRARY_PATH", "ISSL_LIBRARY", "ISSL_SERVER")}
    r = subprocess.run(args, capture_output=True, env=env, cwd=str(tmp_path))
    assert r.returncode == 1 and r.stdout == b"" and b"cannot load libissl_hip.so" in r.stderr and b"ISSL_LIBRARY" in r.stderr
    r = subprocess.run(args, capture_output=True, env=dict(env, ISSL_LIBRARY=_lib.LIB_PATH), cwd=str(tmp_path))
    assert r.stdout == b"" and b"cannot load" not in r.stderr
    if not _has_gpu():
        assert r.returncode == 1 and b"no HIP device" in r.stderr
    # usage errors need no library at all
    r = subprocess.run([str(exe), str(golden.issl)], capture_output=True, env=env)
    assert r.returncode == 1 and b"Usage:" in r.stderr


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        return False


def test_every_documented_option_can_be_set_and_read_back(golden_uniform):
    """The options `include/issl_hip.h` lists for issl_index_set_option (their environment names in brackets there): each
    takes a value of its range and gives it back through issl_index_get_option; a value outside the range is refused and
    leaves the option as it was.  No device needed: options live on the handle."""
    import re
    header = (ROOT / "include" / "issl_hip.h").read_text()
    named = {n for n in re.findall(r"\b([a-z][a-z_0-9]+)\b", header) if ("ISSL_" + n.upper()) in header}  # name and environment name both there
    # (name, a value inside its range, a value outside it or None)
    cases = {"prune": (1, 2), "lanes": (3, 4), "hit_slots": (2, 3), "lean_tail": (0, 2), "small_bin": (0, 2), "scan_events": (1, 3),
             "scan_threads": (768, 63), "scan_blocks": (512, None), "upload_threads": (3, 33), "upload_chunk_kib": (64, 3),
             "upload_ring_min_kib": (0, None), "fine_items": (16, None), "expect_guides": (1000, None), "scan_generic": (1, 2),
             "stage_timing": (1, 2), "tail_shapes": (0, None), "item_guides": (64, None), "raw_chunks": (1000, None)}
    missing = sorted(n for n in ("small_bin", "scan_events", "upload_threads", "fine_items", "expect_guides", "lean_tail", "lanes", "hit_slots")
                     if n not in named)
    assert not missing, f"options this round added or changed are not in the header's list: {missing}"
    ix = ca.IsslIndex.open(golden_uniform.issl)
    try:
        for name, (good, bad) in cases.items():
            before = ix.get_option(name)
            ix.set_option(name, good)
            assert ix.get_option(name) == good, name
            if bad is not None:
                with pytest.raises(ca.IsslError):
                    ix.set_option(name, bad)
                assert ix.get_option(name) == good, name
            ix.set_option(name, before)
    finally:
        ix.close()
