"""Multi-GPU orchestration of the scoring step: one process per GPU, `torch.distributed`.

Guides are independent (isslScoreOfftargets.cpp:316-509 touches only the read-only index), so a batch is cut into
per-rank shards with no data-path collective.  The only communication is
  * start-up: the HBM index image is produced by rank 0 and broadcast (RCCL over xGMI on GPUs),
  * end: a gather of 16 B per guide (MIT, CFD) to rank 0, which prints in input order (:514-527).
The reference splits its guide loop statically over OpenMP threads (:316); Crackling emits guides in genome order, so
contiguous eighths would put a repeat-dense region on one GPU.  Shards are therefore INTERLEAVED chunks: chunk k (of
`chunk` guides) belongs to rank k mod world, which spreads every region over all ranks (SURVEY 8e "finer chunks").
The functions take the process-group functions from `torch.distributed`, so the same code runs on the `gloo` backend
in the CPU tests (tests/test_sharding_gloo.py) and on `nccl` (= RCCL) on the GPU node; bench.py uses them as they are.
"""
import numpy as np

DEFAULT_CHUNK = 4096


def shard_bounds(n, world, rank):
    """Contiguous shard [lo, hi) of rank `rank`: sizes differ by at most one, order preserved."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_indices(n, world, rank, chunk=DEFAULT_CHUNK):
    """Indices (ascending) of the guides rank `rank` scores: chunks of `chunk` guides dealt round-robin.

    chunk=None gives the contiguous split of shard_bounds()."""
    if chunk is None:
        lo, hi = shard_bounds(n, world, rank)
        return np.arange(lo, hi, dtype=np.int64)
    idx = np.arange(n, dtype=np.int64)
    return idx[(idx // chunk) % world == rank]


def shard_size(n, world, rank, chunk=DEFAULT_CHUNK):
    if chunk is None:
        lo, hi = shard_bounds(n, world, rank)
        return hi - lo
    full, rest = divmod(n, chunk)            # `full` whole chunks, then one of `rest` guides
    mine = full // world + (1 if rank < full % world else 0)
    return mine * chunk + (rest if full % world == rank else 0)


def _staged(dist, torch, device):
    """True when collectives on `device` tensors have to go through host memory: the `gloo` backend moves CPU tensors
    (its CUDA support covers only some collectives); `nccl` (= RCCL) moves device tensors as they are."""
    return getattr(device, "type", str(device)) != "cpu" and str(dist.get_backend()).lower() == "gloo"


def broadcast_image(dist, torch, index, device, src=0, index_cls=None, piece_bytes=1 << 30, stats=None):
    """Rank `src` holds an IsslIndex (host arrays, or an image already on `device`); every rank ends up with an
    IsslIndex attached to its own copy of the image.  Returns (index, seconds spent in the broadcast proper).
    The index passed in on rank `src` is CONSUMED when it already has a device image: that image is copied into the
    tensor the ranks share the layout of and the library-owned copy is closed; use the returned index from then on.

    The image travels as ONE uint8 tensor: rank `src` builds it straight into the tensor (host arrays) or copies its
    device-built image there; the others receive and attach (issl_index_attach_image).  The broadcast itself goes in
    pieces of `piece_bytes` (1 GiB: a 300 M-site image is 61 GB, and no collective has to take a count beyond 2^31).
    `index_cls` (default IsslIndex) provides attach_tensor(); the CPU tests pass a host-memory stand-in with the same
    methods.  `stats` (a dict, optional) receives `image_bytes` and, on a GPU, `hbm_used_peak_bytes`: what the device holds
    at the high-water mark of this function on this rank -- on rank `src` with a device-built index that is the library's
    image AND the tensor it is copied into (2 x 44.5 GB at 300 M sites), everywhere else the tensor alone."""
    import time
    if index_cls is None:
        from .scorer import IsslIndex as index_cls
    rank = dist.get_rank()
    staged = _staged(dist, torch, device)
    nbytes = torch.zeros(1, dtype=torch.int64, device="cpu" if staged else device)
    if rank == src:
        # an image with sections in pinned host memory cannot be adopted by another process (issl_index_attach_image):
        # say so BEFORE the source index is consumed (-1 = every rank raises, nobody hangs in the broadcast)
        cold = index.cold()[1] if index.has_device_image() and hasattr(index, "cold") else 0
        nbytes[0] = -1 if cold else index.device_bytes()
    dist.broadcast(nbytes, src)
    n = int(nbytes.item())
    if n < 0:
        raise RuntimeError("broadcast_image: the image on the source rank keeps its slice lists in pinned host memory "
                           "(host_cold=1); upload it with keep_lists=0 / without host_cold so that it is self-contained")
    raw = torch.empty(n + 256, dtype=torch.uint8, device=device)
    off = (-raw.data_ptr()) % 256
    image = raw[off:off + n]
    attach = True

    def note_peak():
        if stats is not None:
            stats["image_bytes"] = n
            if getattr(device, "type", str(device)) != "cpu" and torch.cuda.is_available():
                free_b, total_b = torch.cuda.mem_get_info(device)
                stats["hbm_used_peak_bytes"] = max(stats.get("hbm_used_peak_bytes", 0), int(total_b - free_b))
    note_peak()
    if rank == src:
        if index.has_device_image():           # built on the device: move the image into the tensor, then attach like
            index.copy_image_to_tensor(image)  # everybody else (the library-owned copy is released)
            note_peak()
            index.close()
        else:                                  # host arrays: build the image straight into the tensor
            index.upload_into_tensor(image)
            attach = False
    _sync(torch, device)
    t0 = time.perf_counter()
    for at in range(0, n, piece_bytes):
        piece = image[at:at + piece_bytes]
        if staged:  # (test set-ups only: two ranks sharing one GPU over gloo)
            host = piece.cpu()
            dist.broadcast(host, src)
            if rank != src:
                piece.copy_(host)
        else:
            dist.broadcast(piece, src)
    _sync(torch, device)
    seconds = time.perf_counter() - t0
    return (index_cls.attach_tensor(image) if attach else index), seconds


def _sync(torch, device):
    if getattr(device, "type", str(device)) != "cpu" and torch.cuda.is_available():
        torch.cuda.synchronize()


class ShardLayout:
    """Who scores which guide of an n-guide batch, and how the gathered pieces go back into input order.
    Built once per batch shape (the index tensors live on `device`), reused for every gather."""

    def __init__(self, torch, n, world, chunk=DEFAULT_CHUNK, device="cpu"):
        self.n, self.world, self.chunk = n, world, chunk
        self.sizes = [shard_size(n, world, r, chunk) for r in range(world)]
        self.width = max(self.sizes) if self.sizes else 0
        self.indices = [shard_indices(n, world, r, chunk) for r in range(world)]
        self.index_tensors = [torch.from_numpy(i).to(device) for i in self.indices]


def score_sharded(dist, torch, score_fn, guides, device="cpu", dst=0, chunk=DEFAULT_CHUNK):
    """Score `guides` (uint64 signatures, identical on every rank) shard-wise and gather on `dst`.

    score_fn(sigs) -> (mit, cfd) float64 arrays for the local shard.  Returns (mit, cfd) numpy arrays in input order
    on rank `dst`, (None, None) elsewhere."""
    world, rank = dist.get_world_size(), dist.get_rank()
    layout = ShardLayout(torch, len(guides), world, chunk, device)
    mit, cfd = score_fn(guides[layout.indices[rank]])
    out_m, out_c = gather_scores(dist, torch, layout, mit, cfd, device=device, dst=dst)
    if out_m is None:
        return None, None
    return out_m.cpu().numpy(), out_c.cpu().numpy()


def gather_scores(dist, torch, layout, mit, cfd, device="cpu", dst=0):
    """Gather the per-rank score arrays (any leading shape, last axis = this rank's shard; numpy or torch) to `dst` and
    put them back in input order THERE, on `device`: returns two float64 tensors of shape (..., n) on `dst`,
    (None, None) elsewhere.  One collective: dist.gather of a (2, ..., width) tensor per rank."""
    world, rank = dist.get_world_size(), dist.get_rank()
    mit_t = mit if isinstance(mit, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(mit))
    cfd_t = cfd if isinstance(cfd, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(cfd))
    lead = tuple(mit_t.shape[:-1])
    local = torch.zeros((2,) + lead + (layout.width,), dtype=torch.float64, device=device)
    local[0, ..., :layout.sizes[rank]] = mit_t.to(device)
    local[1, ..., :layout.sizes[rank]] = cfd_t.to(device)
    if _staged(dist, torch, device):
        local = local.cpu()
    parts = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local, parts, dst=dst)
    if rank != dst:
        return None, None
    parts = [p.to(device) for p in parts]
    out = torch.empty((2,) + lead + (layout.n,), dtype=torch.float64, device=device)
    for r in range(world):
        out[..., layout.index_tensors[r]] = parts[r][..., :layout.sizes[r]]
    return out[0], out[1]
