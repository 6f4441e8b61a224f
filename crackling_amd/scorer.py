"""Host-side mirror of the reference interface for the ISSL scoring step.

`IsslIndex` wraps an index handle of the C ABI; `run_scorer_binary` / `parse_scorer_output` restate
what Crackling's driver does around the scorer process (src/crackling/Crackling.py:747-786): write
`seq[0:20] + "\\n"` per guide, run `<binary> <issl> <query> <maxDist> <threshold> <method> > out`,
split each stdout line on tabs and keep lines with exactly three fields.
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

from . import _lib
from ._lib import lib, check, IsslError

METHODS = {"unknown": 0, "mit": 1, "cfd": 2, "and": 3, "or": 4, "avg": 5}


def _method_code(method):
    if isinstance(method, str):
        return lib.issl_method_from_string(method.encode())
    return int(method)


def encode_guides(seqs, seq_len=20):
    """2-bit pack guides (isslScoreOfftargets.cpp:63-71). seqs: iterable of str/bytes of length seq_len."""
    seqs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
    for s in seqs:
        if len(s) != seq_len:
            raise ValueError(f"guide of length {len(s)}, expected {seq_len}")
    out = np.empty(len(seqs), dtype=np.uint64)
    if seqs:
        check(lib.issl_encode_guides(b"".join(seqs), len(seqs), seq_len, seq_len, out.ctypes.data))
    return out


def decode_guides(sigs, seq_len=20):
    buf = C.create_string_buffer(seq_len + 1)
    out = []
    for s in np.asarray(sigs, dtype=np.uint64):
        check(lib.issl_decode_guide(int(s), seq_len, buf))
        out.append(buf.value.decode())
    return out


def format_scores(sigs, mit, cfd, method, seq_len=20):
    """The scorer's stdout (isslScoreOfftargets.cpp:514-527) for already computed scores."""
    code = _method_code(method)
    want_mit = code in (1, 3, 4, 5)
    want_cfd = code in (2, 3, 4, 5)
    lines = []
    for seq, m, c in zip(decode_guides(sigs, seq_len), mit, cfd):
        lines.append(f"{seq}\t{('%f' % m) if want_mit else '-1'}\t{('%f' % c) if want_cfd else '-1'}\n")
    return "".join(lines)


def format_scores_native(sigs, mit, cfd, method, seq_len=20, threads=0):
    """The same text from the library's own formatter (issl_format_scores: what bin/isslScoreOfftargets prints), as bytes."""
    sigs = np.ascontiguousarray(sigs, dtype=np.uint64)
    mit = np.ascontiguousarray(mit, dtype=np.float64)
    cfd = np.ascontiguousarray(cfd, dtype=np.float64)
    assert len(sigs) == len(mit) == len(cfd)
    from ._lib import Span
    spans, n_spans = C.POINTER(Span)(), C.c_size_t()
    check(lib.issl_format_scores(sigs.ctypes.data, mit.ctypes.data, cfd.ctypes.data, len(sigs), seq_len, _method_code(method),
                                 threads, C.byref(spans), C.byref(n_spans)))
    try:
        return b"".join(C.string_at(spans[i].data, spans[i].len) for i in range(n_spans.value))
    finally:
        lib.issl_free_spans(spans, n_spans)


class IsslIndex:
    """An ISSL index: host arrays (.issl sections) and, after upload(), its HBM image."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle) if not isinstance(handle, C.c_void_p) else handle
        self._keep = None  # torch tensor that backs the device image, if any

    # -- construction ------------------------------------------------------------------------
    @classmethod
    def open(cls, path):
        h = C.c_void_p()
        check(lib.issl_index_open(os.fsencode(path), C.byref(h)))
        return cls(h)

    @classmethod
    def from_bytes(cls, data):
        h = C.c_void_p()
        buf = (C.c_char * len(data)).from_buffer_copy(data)
        check(lib.issl_index_from_memory(C.addressof(buf), len(data), C.byref(h)))
        return cls(h)

    @classmethod
    def build_from_text(cls, text, seq_len=20, slice_width=8):
        """isslCreateIndex counterpart; text = sorted sites, one per line."""
        if isinstance(text, str):
            text = text.encode()
        if len(text) % (seq_len + 1):
            raise ValueError("site list is not a multiple of the line length")
        h = C.c_void_p()
        check(lib.issl_index_build_from_text(text, len(text) // (seq_len + 1), seq_len, slice_width, C.byref(h)))
        return cls(h)

    @classmethod
    def build_from_sites(cls, sigs, occ, n_lines=None, seq_len=20, slice_width=8):
        sigs = np.ascontiguousarray(sigs, dtype=np.uint64)
        occ = np.ascontiguousarray(occ, dtype=np.uint32)
        if n_lines is None:
            n_lines = int(occ.sum(dtype=np.uint64))
        h = C.c_void_p()
        check(lib.issl_index_build_from_sites(sigs.ctypes.data, occ.ctypes.data, len(sigs), n_lines, seq_len,
                                              slice_width, C.byref(h)))
        return cls(h)

    @classmethod
    def build_on_device(cls, sigs, occ, device=0, n_lines=None, seq_len=20, slice_width=8, options=None):
        """Like build_from_sites + upload, but the slice lists are built on the GPU (no 48 B/site host arrays).
        options: layout options of the image, e.g. {"compact": 1, "host_cold": 1}."""
        sigs = np.ascontiguousarray(sigs, dtype=np.uint64)
        occ = np.ascontiguousarray(occ, dtype=np.uint32)
        if n_lines is None:
            n_lines = int(occ.sum(dtype=np.uint64))
        h = C.c_void_p()
        opts = ",".join(f"{k}={v}" for k, v in (options or {}).items()).encode() or None
        check(lib.issl_index_build_on_device_opt(sigs.ctypes.data, occ.ctypes.data, len(sigs), n_lines, seq_len,
                                                 slice_width, device, opts, C.byref(h)))
        return cls(h)

    @classmethod
    def build_from_device_sites(cls, d_sigs, d_occ, n_lines, device=0, seq_len=20, slice_width=8, options=None):
        """The same for a site table that already sits in the memory of `device`: d_sigs (int64/uint64 view of the packed
        signatures, text order, distinct) and d_occ (int32/uint32 counts) are torch CUDA tensors; nothing of the size of the
        index touches host memory.  The tensors may be freed afterwards."""
        assert d_sigs.is_cuda and d_occ.is_cuda and d_sigs.element_size() == 8 and d_occ.element_size() == 4
        assert d_sigs.is_contiguous() and d_occ.is_contiguous() and d_sigs.numel() == d_occ.numel()
        h = C.c_void_p()
        opts = ",".join(f"{k}={v}" for k, v in (options or {}).items()).encode() or None
        check(lib.issl_index_build_from_device_sites(d_sigs.data_ptr(), d_occ.data_ptr(), d_sigs.numel(), int(n_lines), seq_len,
                                                     slice_width, device, opts, C.byref(h)))
        return cls(h)

    @classmethod
    def attach_tensor(cls, tensor):
        """Adopt an HBM image that arrived in a torch uint8 CUDA tensor (e.g. by RCCL broadcast)."""
        h = C.c_void_p()
        check(lib.issl_index_attach_image(tensor.device.index or 0, tensor.data_ptr(), tensor.numel(), C.byref(h)))
        ix = cls(h)
        ix._keep = tensor
        return ix

    # -- properties ---------------------------------------------------------------------------
    @property
    def header(self):
        hd = _lib.Header()
        check(lib.issl_index_header(self._h, C.byref(hd)))
        return {n: int(getattr(hd, n)) for n, _ in hd._fields_}

    def bucket_sizes(self):
        hd = self.header
        n = hd["n_slices"] << hd["slice_width"]
        out = np.empty(n, dtype=np.uint64)
        check(lib.issl_index_bucket_sizes(self._h, out.ctypes.data, n))
        return out

    def write(self, path):
        check(lib.issl_index_write(self._h, os.fsencode(path)))

    def device_bytes(self):
        n = C.c_size_t()
        check(lib.issl_index_device_bytes(self._h, C.byref(n)))
        return n.value

    # -- device -------------------------------------------------------------------------------
    def upload(self, device=0):
        check(lib.issl_index_upload(self._h, device))
        return self

    def upload_into_tensor(self, tensor):
        """Build the HBM image inside a caller-owned torch uint8 CUDA tensor."""
        check(lib.issl_index_upload_into(self._h, tensor.device.index or 0, tensor.data_ptr(), tensor.numel()))
        self._keep = tensor
        return self

    def set_option(self, key, value):
        """Tuning knob of this handle (include/issl_hip.h: issl_index_set_option); no batches may be in flight."""
        check(lib.issl_index_set_option(self._h, str(key).encode(), str(value).encode()))
        return self

    def get_option(self, key):
        v = C.c_longlong()
        check(lib.issl_index_get_option(self._h, str(key).encode(), C.byref(v)))
        return v.value

    def cold(self):
        """(host pointer, bytes) of the pinned host buffer holding the cold sections, (None, 0) when all is in HBM."""
        p = C.c_void_p()
        n = C.c_size_t()
        check(lib.issl_index_cold(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def copy_image_to_tensor(self, tensor):
        """Device-to-device copy of the HBM image into a torch uint8 CUDA tensor (256-byte aligned data pointer)."""
        check(lib.issl_index_copy_image_to(self._h, tensor.data_ptr(), tensor.numel()))
        return tensor

    def has_device_image(self):
        return self.get_option("cold_on_host") >= 0

    def image(self):
        p = C.c_void_p()
        n = C.c_size_t()
        check(lib.issl_index_image(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    # -- scoring ------------------------------------------------------------------------------
    def _sigs(self, guides):
        if isinstance(guides, np.ndarray) and guides.dtype == np.uint64:
            return np.ascontiguousarray(guides)
        return encode_guides(guides, self.header["seq_len"])

    def score(self, guides, max_dist=4, threshold=75.0, method="and"):
        """-> (mit, cfd) float64 arrays: 10000/(100+sum) per guide (isslScoreOfftargets.cpp:505-506)."""
        sigs = self._sigs(guides)
        mit = np.empty(len(sigs), dtype=np.float64)
        cfd = np.empty(len(sigs), dtype=np.float64)
        check(lib.issl_score(self._h, sigs.ctypes.data, len(sigs), int(max_dist), float(threshold),
                             _method_code(method), mit.ctypes.data, cfd.ctypes.data))
        return mit, cfd

    def score_device(self, d_guides, d_mit, d_cfd, max_dist=4, threshold=75.0, method="and", stream=None):
        """Guides (int64/uint64 view of packed signatures) and outputs are torch CUDA tensors."""
        check(lib.issl_score_device(self._h, d_guides.data_ptr(), d_guides.numel(), int(max_dist), float(threshold),
                                    _method_code(method), d_mit.data_ptr(), d_cfd.data_ptr(),
                                    C.c_void_p(stream) if stream else None))

    def score_device_async(self, d_guides, d_mit, d_cfd, max_dist=4, threshold=75.0, method="and", stream=None):
        """Enqueue only; call finish() before trusting the outputs."""
        check(lib.issl_score_device_async(self._h, d_guides.data_ptr(), d_guides.numel(), int(max_dist),
                                          float(threshold), _method_code(method), d_mit.data_ptr(), d_cfd.data_ptr(),
                                          C.c_void_p(stream) if stream else None))

    def wait(self, stream):
        """Make `stream` wait for every batch enqueued so far (no host synchronisation)."""
        check(lib.issl_score_wait(self._h, C.c_void_p(stream) if stream else None))

    def finish(self, stream=None):
        """Synchronise the enqueued batches.  Returns False when they must be enqueued again (scratch space grew)."""
        rc = lib.issl_score_finish(self._h, C.c_void_p(stream) if stream else None)
        if rc == -8:
            return False
        check(rc)
        return True

    def dump_hits(self, guides, max_dist=4, threshold=0.0, method="and"):
        """Scored off-targets in the reference's scoring order: array of (guide, slice, pos, id, dist, occ)."""
        sigs = self._sigs(guides)
        n = C.c_size_t()
        check(lib.issl_dump_hits(self._h, sigs.ctypes.data, len(sigs), int(max_dist), float(threshold),
                                 _method_code(method), None, 0, C.byref(n)))
        out = np.empty((n.value, 6), dtype=np.uint32)
        if n.value:
            check(lib.issl_dump_hits(self._h, sigs.ctypes.data, len(sigs), int(max_dist), float(threshold),
                                     _method_code(method), out.ctypes.data, n.value, C.byref(n)))
        return out

    def stats(self):
        st = _lib.Stats()
        check(lib.issl_last_stats(self._h, C.byref(st)))
        return {n: getattr(st, n) for n, _ in st._fields_}

    def count_candidates(self, guides):
        sigs = self._sigs(guides)
        out = C.c_uint64()
        check(lib.issl_count_candidates(self._h, sigs.ctypes.data, len(sigs), C.byref(out)))
        return out.value

    def close(self):
        if self._h:
            lib.issl_index_close(self._h)
            self._h = None
            self._keep = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class IsslNode:
    """Several GPUs of this node in one process: the index image is uploaded on devices[0], broadcast (RCCL over
    xGMI, peer copies as fallback) and every batch is a queue of chunks, one host thread per device taking the next
    chunk when it has finished its last (scores land in input order)."""

    def __init__(self, index, devices=None):
        self._index = index  # keep the host arrays / root image alive
        h = C.c_void_p()
        if devices is None:
            check(lib.issl_node_create(index._h, None, 0, C.byref(h)))
        else:
            arr = (C.c_int * len(devices))(*devices)
            check(lib.issl_node_create(index._h, arr, len(devices), C.byref(h)))
        self._h = h

    def score(self, guides, max_dist=4, threshold=75.0, method="and"):
        sigs = self._index._sigs(guides)
        mit = np.empty(len(sigs), dtype=np.float64)
        cfd = np.empty(len(sigs), dtype=np.float64)
        check(lib.issl_node_score(self._h, sigs.ctypes.data, len(sigs), int(max_dist), float(threshold),
                                  _method_code(method), mit.ctypes.data, cfd.ctypes.data))
        return mit, cfd

    def shard_times(self):
        """Per device, for the last score(): (milliseconds spent scoring, guides scored)."""
        n = self.info()["n_devices"]
        ms = (C.c_double * n)()
        g = (C.c_uint64 * n)()
        check(lib.issl_node_shard_times(self._h, ms, g, n))
        return list(ms), list(g)

    def info(self):
        inf = _lib.NodeInfo()
        check(lib.issl_node_get_info(self._h, C.byref(inf)))
        return {n: getattr(inf, n) for n, _ in inf._fields_}

    def close(self):
        if self._h:
            lib.issl_node_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def extract_offtargets(fasta_blobs, device=0):
    """Sorted site list (bytes, one 20-mer per line) of FASTA / multi-FASTA contents -- the GPU counterpart of
    crackling/utils/extractOfftargets.py.  fasta_blobs: iterable of bytes."""
    blobs = [b if isinstance(b, bytes) else b.encode() for b in fasta_blobs]
    files = (C.c_char_p * len(blobs))(*blobs)
    lens = (C.c_size_t * len(blobs))(*[len(b) for b in blobs])
    out = C.c_void_p()
    n = C.c_size_t()
    sites = C.c_uint64()
    check(lib.issl_extract_from_memory(files, lens, len(blobs), device, C.byref(out), C.byref(n), C.byref(sites)))
    data = C.string_at(out, n.value) if n.value else b""
    lib.issl_free(out)
    return data


def run_scorer_binary(binary, issl_path, guides23, max_dist, threshold, method, workdir=None, env=None):
    """What Crackling.py:747-778 does for one page of guides: returns the scorer's stdout text.

    guides23: iterable of target strings; only the first 20 characters are written (:751)."""
    with tempfile.TemporaryDirectory(dir=workdir) as tmp:
        inp = os.path.join(tmp, "offtargetscore.input")
        outp = os.path.join(tmp, "offtargetscore.output")
        with open(inp, "w") as fh:
            for g in guides23:
                fh.write(g[0:20] + "\n")
        cmd = f"{binary} {issl_path} {inp} {max_dist} {threshold} {method} > {outp}"
        subprocess.run(cmd, shell=True, check=True, env=env)  # Helpers.py:39-42
        with open(outp) as fh:
            return fh.read()


def parse_scorer_output(text):
    """Crackling.py:780-786: {sequence: {"mit": float, "cfd": float}} from 3-field lines."""
    out = {}
    for line in text.splitlines():
        parts = line.strip().split("\t")
        if len(parts) == 3:
            out[parts[0]] = {"mit": float(parts[1]), "cfd": float(parts[2])}
    return out


def verdicts(mit, cfd, threshold, method):
    """Crackling.py:780-835 on arrays: 1 = accepted, 0 = rejected, 255 = no rule matches the method name.

    `method` is the configured string (the scorer matches it exactly, the caller lower-cases it)."""
    mit = np.ascontiguousarray(mit, dtype=np.float64)
    cfd = np.ascontiguousarray(cfd, dtype=np.float64)
    if mit.shape != cfd.shape or mit.ndim != 1:
        raise ValueError("mit and cfd must be 1-D arrays of the same length")
    out = np.empty(len(mit), dtype=np.uint8)
    check(lib.issl_verdicts(mit.ctypes.data, cfd.ctypes.data, len(mit), float(threshold), str(method).encode(),
                            out.ctypes.data))
    return out
