"""CPU: the host-side .issl parser, the index builder, the guide codec and the scorer's text code (own "%f", threaded query
reader) under AddressSanitizer + UBSan (tools/host_sanitize.cpp: every truncation of a golden index, 4000 random field / word /
bit corruptions, the builder's round trip, 70 000 lines formatted against printf's on one and four threads).  GPU sanitizers are not available on the pool; this is the part of the product that reads untrusted bytes."""
import pathlib
import subprocess

ROOT = pathlib.Path(__file__).resolve().parent.parent


def test_parser_and_builder_are_clean_under_asan_and_ubsan(golden, tmp_path):
    exe = tmp_path / "host_sanitize"
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                            str(ROOT / "tools" / "host_sanitize.cpp"), str(ROOT / "crackling_amd" / "csrc" / "issl_host.cpp"),
                            str(ROOT / "crackling_amd" / "csrc" / "issl_text.cpp"),
                            "-lpthread", "-o", str(exe)], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    index = tmp_path / "index.issl"          # the harness writes a temporary file next to the index
    index.write_bytes(golden.issl.read_bytes())
    run = subprocess.run([str(exe), str(index), str(golden.sites_txt)], capture_output=True, text=True,
                         env={"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert run.returncode == 0, run.stdout + run.stderr
    assert run.stdout.startswith("ok:") and "ERROR" not in run.stderr, run.stdout + run.stderr
