"""Off-target site extraction: oracle vs the golden vectors of the reference Python (CPU), GPU
implementation vs goldens and oracle (GPU)."""
import ctypes as C
import pathlib
import subprocess

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
GOLD = ROOT / "tests" / "golden" / "extract"
SETS = ["multi", "repeat"]


def oracle_extract(blobs):
    so = ROOT / "oracle" / "_build" / "libextract_oracle.so"
    if not so.exists():
        subprocess.run(["make", "-C", str(ROOT / "oracle"), "all"], check=True, capture_output=True)
    lib = C.CDLL(str(so))
    lib.oracle_extract.restype = C.c_void_p
    lib.oracle_extract.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.POINTER(C.c_size_t)]
    lib.oracle_extract_free.argtypes = [C.c_void_p]
    files = (C.c_char_p * len(blobs))(*blobs)
    lens = (C.c_size_t * len(blobs))(*[len(b) for b in blobs])
    n = C.c_size_t()
    p = lib.oracle_extract(files, lens, len(blobs), C.byref(n))
    out = C.string_at(p, n.value)
    lib.oracle_extract_free(p)
    return out


def random_fasta(seed, n_records, max_len, p_n=0.003, lower=0.2, width=70):
    rng = np.random.default_rng(seed)
    out = []
    for r in range(n_records):
        n = int(rng.integers(0, max_len))
        s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].copy()
        s[rng.random(n) < p_n] = ord("N")
        m = rng.random(n) < lower
        s[m] = s[m] + 32  # lower case
        seq = s.tobytes().decode()
        out.append(f">rec{r} something\n")
        out += [seq[i:i + width] + "\n" for i in range(0, len(seq), width)]
    return "".join(out).encode()


@pytest.mark.parametrize("name", SETS)
def test_oracle_matches_reference_python_golden(name):
    got = oracle_extract([(GOLD / f"{name}.fa").read_bytes()])
    assert got == (GOLD / f"{name}.sites.txt").read_bytes()


def test_oracle_known_sites():
    # one forward site (N20 + NGG) and its text; one reverse-pattern match (first 20 chars reverse-complemented)
    out = oracle_extract([b">a\nACGTACGTACGTACGTACGTAGG\n>b\nCCAACGTACGTACGTACGTACGTT\n"]).decode().split()
    assert "ACGTACGTACGTACGTACGT" in out
    rc = str.maketrans("ACGT", "TGCA")
    assert "CCAACGTACGTACGTACGTA".translate(rc)[::-1] in out
    # a T in the first position is not a forward site (pattern starts with [ACG])
    assert oracle_extract([b">a\nTCGTACGTACGTACGTACGTAGG\n"]) == b""


def test_extraction_without_device_fails_loudly(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import crackling_amd as ca
    with pytest.raises(ca.IsslError) as e:
        ca.extract_offtargets([(GOLD / "multi.fa").read_bytes()])
    assert e.value.code == -5 and "no CPU fallback" in str(e.value)
    r = subprocess.run([str(ROOT / "bin" / "extractOfftargets"), str(tmp_path / "o.txt"), str(GOLD / "multi.fa")],
                       capture_output=True)
    assert r.returncode == 1 and b"no HIP device" in r.stderr and not (tmp_path / "o.txt").exists()
    r = subprocess.run([str(ROOT / "bin" / "extractOfftargets"), str(tmp_path / "o.txt")], capture_output=True)
    assert r.returncode == 2 and b"usage" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("name", SETS)
def test_gpu_extraction_matches_reference_golden(name):
    import crackling_amd as ca
    got = ca.extract_offtargets([(GOLD / f"{name}.fa").read_bytes()])
    assert got == (GOLD / f"{name}.sites.txt").read_bytes()


@pytest.mark.gpu
def test_gpu_extraction_matches_oracle_on_random_genomes(tmp_path):
    import crackling_amd as ca
    for seed, recs, mx in [(1, 5, 20000), (2, 1, 300000), (3, 40, 3000), (4, 3, 30)]:
        blob = random_fasta(seed, recs, mx)
        assert ca.extract_offtargets([blob]) == oracle_extract([blob]), seed
    blobs = [random_fasta(10 + i, 3, 50000) for i in range(3)]
    want = oracle_extract(blobs)
    assert ca.extract_offtargets(blobs) == want
    lines = want.split(b"\n")[:-1]
    assert lines == sorted(lines) and len(lines) > 1000
    # executable: output file + the builder consumes it
    paths = []
    for i, b in enumerate(blobs):
        p = tmp_path / f"g{i}.fa"; p.write_bytes(b); paths.append(str(p))
    out = tmp_path / "sites.txt"
    r = subprocess.run([str(ROOT / "bin" / "extractOfftargets"), str(out)] + paths + ["--threads", "4"], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    assert out.read_bytes() == want
    r = subprocess.run([str(ROOT / "bin" / "extractOfftargets"), str(tmp_path / "o2.txt"), str(tmp_path / "g0.fa")], capture_output=True)
    assert r.returncode == 0 and (tmp_path / "o2.txt").read_bytes() == oracle_extract([blobs[0]])
    issl = tmp_path / "x.issl"
    subprocess.run([str(ROOT / "bin" / "isslCreateIndex"), str(out), "20", "8", str(issl)], check=True, capture_output=True)
    ix = ca.IsslIndex.open(issl)
    assert ix.header["n_lines"] == len(lines) and ix.header["n_sites"] == len(set(lines))
    ix.close()


@pytest.mark.gpu
def test_gpu_extraction_of_inputs_parsed_in_several_pieces():
    """Inputs above 4 MB are parsed by several host threads, cut at line starts and joined with the sequential rule for
    record separators: wrapped and unwrapped records, CRLF line ends, blank lines, padded lines, many short records (so
    that pieces start with headers) -- same bytes as the oracle's single pass."""
    import crackling_amd as ca
    rng = np.random.default_rng(77)
    parts = [random_fasta(21, 6, 1_500_000, width=60)]                        # ~4.5 MB, wrapped
    one = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=5_000_000)].tobytes()
    parts.append(b">unwrapped one line\n" + one + b"\n")                    # a 5 MB line
    parts.append(random_fasta(22, 4000, 1200, width=50).replace(b"\n", b"\r\n"))  # CRLF, thousands of headers
    parts.append(b"\n\n  >padded header\n   acgtacgtacgtacgtacgtagg   \n\n>x\n" + random_fasta(23, 3, 900_000))
    blob = b"".join(parts)
    assert len(blob) > 12_000_000
    want = oracle_extract([blob])
    assert ca.extract_offtargets([blob]) == want
    assert ca.extract_offtargets([blob[:6_000_000], blob[6_000_000:]]) == oracle_extract([blob[:6_000_000], blob[6_000_000:]])
    assert want.count(b"\n") > 1_000_000


@pytest.mark.gpu
def test_gpu_extraction_empty_and_tiny_inputs():
    import crackling_amd as ca
    assert ca.extract_offtargets([b""]) == b""
    assert ca.extract_offtargets([b">x\nACGT\n"]) == b""
    assert ca.extract_offtargets([b">x\nACGTACGTACGTACGTACGTAGG"]) == b"ACGTACGTACGTACGTACGT\n"
