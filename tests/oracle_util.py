"""ctypes access to the CPU oracle (oracle/issl_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import pathlib
import subprocess

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
SO = ROOT / "oracle" / "_build" / "liboracle.so"


def _load():
    if not SO.exists():
        subprocess.run(["make", "-C", str(ROOT / "oracle"), "all"], check=True, capture_output=True)
    lib = C.CDLL(str(SO))
    lib.oracle_index_load.restype = C.c_void_p
    lib.oracle_index_load.argtypes = [C.c_char_p]
    lib.oracle_index_free.argtypes = [C.c_void_p]
    lib.oracle_score.restype = C.c_int
    lib.oracle_score.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_double, C.c_int, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.oracle_method_from_string.restype = C.c_int
    lib.oracle_method_from_string.argtypes = [C.c_char_p]
    lib.oracle_encode.restype = C.c_uint64
    lib.oracle_encode.argtypes = [C.c_char_p, C.c_uint64]
    lib.oracle_build_issl.restype = C.c_void_p
    lib.oracle_build_issl.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.oracle_free.argtypes = [C.c_void_p]
    lib.oracle_local_mit.restype = C.c_double
    lib.oracle_local_mit.argtypes = [C.c_uint64, C.c_uint64]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


class OracleIndex:
    def __init__(self, path):
        self.h = lib().oracle_index_load(os.fsencode(str(path)))
        if not self.h:
            raise RuntimeError(f"oracle could not load {path}")

    def score(self, sigs, max_dist=4, threshold=75.0, method="and", threads=0, want_hits=False, hit_cap=1 << 22):
        sigs = np.ascontiguousarray(sigs, dtype=np.uint64)
        n = len(sigs)
        mit = np.empty(n, dtype=np.float64)
        cfd = np.empty(n, dtype=np.float64)
        m = lib().oracle_method_from_string(method.encode())
        nh = C.c_uint64(0)
        hits = np.empty((hit_cap if want_hits else 0, 6), dtype=np.uint32)
        rc = lib().oracle_score(self.h, sigs.ctypes.data, n, max_dist, float(threshold), m, threads,
                                mit.ctypes.data, cfd.ctypes.data, hits.ctypes.data if want_hits else None,
                                hit_cap if want_hits else 0, C.byref(nh))
        assert rc == 0
        if want_hits:
            assert nh.value <= hit_cap, "oracle hit buffer too small"
            return mit, cfd, hits[: nh.value].copy()
        return mit, cfd

    def close(self):
        if self.h:
            lib().oracle_index_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def encode(seqs, seq_len=20):
    return np.array([lib().oracle_encode(s.encode() if isinstance(s, str) else s, seq_len) for s in seqs], dtype=np.uint64)


def build_issl(text, seq_len=20, slice_width=8):
    if isinstance(text, str):
        text = text.encode()
    n = C.c_uint64()
    p = lib().oracle_build_issl(text, len(text) // (seq_len + 1), seq_len, slice_width, C.byref(n))
    data = C.string_at(p, n.value)
    lib().oracle_free(p)
    return data


def format_tsv(seqs, mit, cfd, method):
    want_mit = method in ("mit", "and", "or", "avg")
    want_cfd = method in ("cfd", "and", "or", "avg")
    out = []
    for s, m, c in zip(seqs, mit, cfd):
        out.append(f"{s}\t{('%f' % m) if want_mit else '-1'}\t{('%f' % c) if want_cfd else '-1'}\n")
    return "".join(out)
