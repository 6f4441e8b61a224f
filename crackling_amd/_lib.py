"""ctypes binding of libissl_hip.so (C ABI declared in include/issl_hip.h)."""
import ctypes as C
import os

# ISSL_HIP_LIBRARY: another build of the same library (A/B of kernel variants, tools/ablate.py); default: the in-tree one
LIB_PATH = os.environ.get("ISSL_HIP_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libissl_hip.so")


class IsslError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"[{code}] {message}")
        self.code = code
        self.message = message


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make` (or __graft_entry__.build()). "
        "crackling_amd has no CPU fallback for the ISSL scorer."
    )



def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64 (same
    SONAME as /opt/rocm's).  If libissl_hip.so pulled in /opt/rocm's copy first, a later `import torch` would bring
    a second runtime into the process and that one finds no GPU.  So when torch is installed, its libamdhip64 is
    loaded first (without importing torch); libissl_hip.so's DT_NEEDED libamdhip64.so.7 then binds to it by SONAME.
    The standalone executables in bin/ do not go through here and use /opt/rocm's runtime."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
            return cand
    return None


HIP_RUNTIME = _preload_hip_runtime()
lib = C.CDLL(LIB_PATH)


class Header(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_sites", "seq_len", "n_lines", "slice_width", "n_slices", "n_scores")]


class Hit(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("guide", "slice", "pos", "id", "dist", "occ")]


class Stats(C.Structure):
    _fields_ = [
        ("n_guides", C.c_uint64), ("candidates", C.c_uint64), ("hits", C.c_uint64), ("scan_tiles", C.c_uint64),
        ("ms_bin", C.c_double), ("ms_scan", C.c_double), ("ms_verify", C.c_double), ("ms_group", C.c_double),
        ("ms_replay", C.c_double), ("ms_total", C.c_double), ("scan_launches", C.c_uint64), ("raw_records", C.c_uint64), ("n_batches", C.c_uint64),
        ("planned_comparisons", C.c_uint64), ("reference_comparisons", C.c_uint64), ("pruned", C.c_uint64),
        ("ms_scan_events", C.c_double),
    ]


class Span(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_char)), ("len", C.c_size_t)]


class NodeInfo(C.Structure):
    _fields_ = [("n_devices", C.c_int), ("used_rccl", C.c_int), ("ms_upload", C.c_double),
                ("ms_broadcast", C.c_double), ("ms_last_score", C.c_double)]


_P = C.c_void_p
_u64p = C.POINTER(C.c_uint64)
_f64p = C.POINTER(C.c_double)

_protos = {
    "issl_last_error": (C.c_char_p, []),
    "issl_abi_version": (C.c_int, []),
    "issl_index_open": (C.c_int, [C.c_char_p, C.POINTER(_P)]),
    "issl_index_from_memory": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "issl_index_build_from_text": (C.c_int, [C.c_char_p, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(_P)]),
    "issl_index_build_from_sites": (C.c_int, [_P, _P, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(_P)]),
    "issl_index_build_on_device": (C.c_int, [_P, _P, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(_P)]),
    "issl_index_build_on_device_opt": (C.c_int, [_P, _P, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_char_p, C.POINTER(_P)]),
    "issl_index_build_from_device_sites": (C.c_int, [_P, _P, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_char_p, C.POINTER(_P)]),
    "issl_index_write": (C.c_int, [_P, C.c_char_p]),
    "issl_index_header": (C.c_int, [_P, C.POINTER(Header)]),
    "issl_index_bucket_sizes": (C.c_int, [_P, _P, C.c_size_t]),
    "issl_index_close": (C.c_int, [_P]),
    "issl_index_device_bytes": (C.c_int, [_P, C.POINTER(C.c_size_t)]),
    "issl_device_memory": (C.c_int, [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "issl_index_upload": (C.c_int, [_P, C.c_int]),
    "issl_index_upload_into": (C.c_int, [_P, C.c_int, _P, C.c_size_t]),
    "issl_index_attach_image": (C.c_int, [C.c_int, _P, C.c_size_t, C.POINTER(_P)]),
    "issl_index_image": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "issl_index_copy_image_to": (C.c_int, [_P, _P, C.c_size_t]),
    "issl_index_cold": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "issl_index_attach_image_cold": (C.c_int, [C.c_int, _P, C.c_size_t, _P, C.c_size_t, C.POINTER(_P)]),
    "issl_index_set_option": (C.c_int, [_P, C.c_char_p, C.c_char_p]),
    "issl_index_get_option": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_longlong)]),
    "issl_encode_guides": (C.c_int, [C.c_char_p, C.c_size_t, C.c_size_t, C.c_size_t, _P]),
    "issl_decode_guide": (C.c_int, [C.c_uint64, C.c_size_t, C.c_char_p]),
    "issl_read_query_file": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "issl_free": (None, [_P]),
    "issl_format_scores": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.POINTER(Span)), C.POINTER(C.c_size_t)]),
    "issl_free_spans": (None, [C.POINTER(Span), C.c_size_t]),
    "issl_method_from_string": (C.c_int, [C.c_char_p]),
    "issl_score": (C.c_int, [_P, _P, C.c_size_t, C.c_int, C.c_double, C.c_int, _P, _P]),
    "issl_score_device": (C.c_int, [_P, _P, C.c_size_t, C.c_int, C.c_double, C.c_int, _P, _P, _P]),
    "issl_score_device_async": (C.c_int, [_P, _P, C.c_size_t, C.c_int, C.c_double, C.c_int, _P, _P, _P]),
    "issl_score_wait": (C.c_int, [_P, _P]),
    "issl_score_finish": (C.c_int, [_P, _P]),
    "issl_dump_hits": (C.c_int, [_P, _P, C.c_size_t, C.c_int, C.c_double, C.c_int, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "issl_last_stats": (C.c_int, [_P, C.POINTER(Stats)]),
    "issl_count_candidates": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_uint64)]),
    "issl_verdicts": (C.c_int, [_P, _P, C.c_size_t, C.c_double, C.c_char_p, _P]),
    "issl_extract_from_memory": (C.c_int, [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_int,
                                           C.POINTER(_P), C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)]),
    "issl_extract_offtargets": (C.c_int, [C.POINTER(C.c_char_p), C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_uint64)]),
    "issl_node_create": (C.c_int, [_P, C.POINTER(C.c_int), C.c_int, C.POINTER(_P)]),
    "issl_node_score": (C.c_int, [_P, _P, C.c_size_t, C.c_int, C.c_double, C.c_int, _P, _P]),
    "issl_node_get_info": (C.c_int, [_P, C.POINTER(NodeInfo)]),
    "issl_node_shard_times": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]),
    "issl_node_close": (C.c_int, [_P]),
}
for _name, (_res, _args) in _protos.items():
    _fn = getattr(lib, _name)  # AttributeError here = library/header mismatch: fail loudly
    _fn.restype = _res
    _fn.argtypes = _args

EXPORTS = tuple(_protos)


def check(rc):
    if rc != 0:
        raise IsslError(rc, lib.issl_last_error().decode(errors="replace"))
