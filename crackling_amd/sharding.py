"""Multi-GPU orchestration of the scoring step: one process per GPU, `torch.distributed`.

Guides are independent (isslScoreOfftargets.cpp:316-509 touches only the read-only index), so the batch
is cut into `world` contiguous shards with no data-path collective.  The only communication is
  * start-up: the HBM index image is produced by rank 0 and broadcast (RCCL over xGMI on GPUs),
  * end: a gather of 16 B per guide (MIT, CFD) to rank 0, which prints in input order (:514-527).
The functions take the process group functions from `torch.distributed`, so the same code runs on the
`gloo` backend in the CPU tests (tests/test_sharding_gloo.py) and on `nccl` (= RCCL) on the GPU node.
"""
import numpy as np


def shard_bounds(n, world, rank):
    """Contiguous shard [lo, hi) of rank `rank`: sizes differ by at most one, order preserved."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def broadcast_image(dist, torch, index, device, src=0):
    """Rank `src` holds an uploaded IsslIndex; every other rank receives the image and attaches to it.

    Returns the IsslIndex usable on this rank."""
    from .scorer import IsslIndex
    rank = dist.get_rank()
    nbytes = torch.zeros(1, dtype=torch.int64, device=device)
    if rank == src:
        nbytes[0] = index.device_bytes()
    dist.broadcast(nbytes, src)
    n = int(nbytes.item())
    raw = torch.empty(n + 256, dtype=torch.uint8, device=device)
    off = (-raw.data_ptr()) % 256
    image = raw[off:off + n]
    if rank == src:
        index.upload_into_tensor(image)
    dist.broadcast(image, src)
    return index if rank == src else IsslIndex.attach_tensor(image)


def score_sharded(dist, torch, score_fn, guides, device="cpu", dst=0):
    """Score `guides` (uint64 signatures, identical on every rank) shard-wise and gather on `dst`.

    score_fn(sigs) -> (mit, cfd) float64 arrays for the local shard.  Returns (mit, cfd) in input order on
    rank `dst`, (None, None) elsewhere."""
    world, rank = dist.get_world_size(), dist.get_rank()
    n = len(guides)
    lo, hi = shard_bounds(n, world, rank)
    mit, cfd = score_fn(guides[lo:hi])
    width = max(shard_bounds(n, world, r)[1] - shard_bounds(n, world, r)[0] for r in range(world))
    local = torch.zeros(2, width, dtype=torch.float64, device=device)
    if hi > lo:
        local[0, :hi - lo] = torch.from_numpy(np.ascontiguousarray(mit)).to(device)
        local[1, :hi - lo] = torch.from_numpy(np.ascontiguousarray(cfd)).to(device)
    parts = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local, parts, dst=dst)
    if rank != dst:
        return None, None
    out_m = np.empty(n, dtype=np.float64)
    out_c = np.empty(n, dtype=np.float64)
    for r in range(world):
        a, b = shard_bounds(n, world, r)
        part = parts[r].cpu().numpy()
        out_m[a:b] = part[0, :b - a]
        out_c[a:b] = part[1, :b - a]
    return out_m, out_c
