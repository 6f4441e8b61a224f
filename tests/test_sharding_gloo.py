"""CPU, world_size 2 and 3, gloo: the N>1 orchestration bench.py and multi-GPU callers run on `nccl`.

Under test is crackling_amd/sharding.py -- every collective call site of bench.py: the broadcast of the image size and
of the image, the interleaved guide shards, the one gather of the scores and their return to input order.  The HIP
path needs a GPU, so the per-rank scorer here is the oracle (the checker) behind a host-memory stand-in for IsslIndex:
its "image" is the .issl bytes in a CPU tensor, attach_tensor() loads them into the oracle."""
import os
import socket
import sys
import pathlib

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = pathlib.Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class HostImageIndex:
    """IsslIndex stand-in with a host-memory image: the methods sharding.broadcast_image() uses."""

    def __init__(self, issl_bytes=None, oracle=None, device_built=False):
        self.bytes, self.oracle, self.device_built, self.closed = issl_bytes, oracle, device_built, False

    def device_bytes(self):
        return len(self.bytes)

    def has_device_image(self):
        return self.device_built

    def _fill(self, tensor):
        tensor[:len(self.bytes)] = torch.frombuffer(bytearray(self.bytes), dtype=torch.uint8)

    def upload_into_tensor(self, tensor):      # index with host arrays: the image is built straight into the tensor
        self._fill(tensor)
        self.oracle = self._load(self.bytes)

    def copy_image_to_tensor(self, tensor):    # index built on the device: its image is copied into the tensor
        self._fill(tensor)

    def close(self):
        self.closed = True

    @staticmethod
    def _load(data):
        import tempfile
        import oracle_util as ou
        with tempfile.NamedTemporaryFile(suffix=".issl") as f:
            f.write(data); f.flush()
            return ou.OracleIndex(f.name)

    @classmethod
    def attach_tensor(cls, tensor):
        return cls(oracle=cls._load(tensor.numpy().tobytes()))

    def score(self, g):
        return self.oracle.score(g, 4, 75.0, "and", threads=1)


def _worker(rank, world, port, issl, guides_path, out_path, device_built, chunk):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from crackling_amd import sharding
    guides = np.load(guides_path)
    # only rank 0 has the index; everybody else receives the image (same calls as bench.py)
    index = HostImageIndex(open(issl, "rb").read(), device_built=device_built) if rank == 0 else None
    # (pieces of 1000 bytes: the golden image goes in several hundred broadcasts, like a 61 GB one in 1 GiB pieces)
    index, seconds = sharding.broadcast_image(dist, torch, index, torch.device("cpu"), index_cls=HostImageIndex,
                                              piece_bytes=1000 if chunk == 16 else 1 << 30)
    assert seconds >= 0 and index.oracle is not None and not index.closed
    # one batch through score_sharded ...
    mit, cfd = sharding.score_sharded(dist, torch, index.score, guides, chunk=chunk)
    # ... and bench.py's shape: K steps of the local shard, ONE gather of (steps, shard) arrays
    layout = sharding.ShardLayout(torch, len(guides), world, chunk, "cpu")
    m1, c1 = index.score(guides[layout.indices[rank]])
    steps_m, steps_c = torch.from_numpy(np.stack([m1, m1])), torch.from_numpy(np.stack([c1, c1]))
    gm, gc = sharding.gather_scores(dist, torch, layout, steps_m, steps_c, device="cpu")
    if rank == 0:
        assert gm.shape == (2, len(guides))
        np.savez(out_path, mit=mit, cfd=cfd, gm=gm.numpy(), gc=gc.numpy())
    else:
        assert mit is None and gm is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,device_built,chunk", [(2, False, 16), (3, True, 16), (2, True, None), (3, False, 4096)])
def test_broadcast_shard_gather_equal_single_process(golden_uniform, tmp_path, world, device_built, chunk):
    import oracle_util as ou
    sigs = ou.encode(golden_uniform.guides)[:101]  # odd count: ragged shards, a last chunk of 5
    gp = tmp_path / "g.npy"; np.save(gp, sigs)
    out = tmp_path / "out.npz"
    mp.spawn(_worker, args=(world, _free_port(), str(golden_uniform.issl), str(gp), str(out), device_built, chunk),
             nprocs=world, join=True)
    got = np.load(out)
    ix = ou.OracleIndex(golden_uniform.issl)
    mit, cfd = ix.score(sigs, 4, 75.0, "and")
    assert np.array_equal(got["mit"], mit) and np.array_equal(got["cfd"], cfd)
    assert np.array_equal(got["gm"], np.stack([mit, mit])) and np.array_equal(got["gc"], np.stack([cfd, cfd]))


def test_shard_bounds_cover_and_order():
    from crackling_amd.sharding import shard_bounds
    for n in (0, 1, 7, 8, 100, 1001):
        for w in (1, 2, 3, 8):
            cuts = [shard_bounds(n, w, r) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_interleaved_shards_partition_the_batch():
    from crackling_amd.sharding import shard_indices, shard_size
    for n in (0, 1, 5, 4096, 4097, 100_001):
        for w in (1, 2, 3, 8):
            for chunk in (None, 1, 7, 4096):
                parts = [shard_indices(n, w, r, chunk) for r in range(w)]
                assert [len(p) for p in parts] == [shard_size(n, w, r, chunk) for r in range(w)]
                allidx = np.concatenate(parts) if parts else np.empty(0, dtype=np.int64)
                assert np.array_equal(np.sort(allidx), np.arange(n))
                assert all((np.diff(p) > 0).all() for p in parts)       # input order kept inside a shard
                if chunk is not None and n >= w * chunk * 4:             # a region of the batch is spread over all ranks
                    region = np.arange(n // 2, n // 2 + w * chunk)
                    assert all(np.intersect1d(p, region).size > 0 for p in parts)


def _worker_layout(rank, world, port, n, chunk, steps, out_path):
    """The orchestration alone at full size: a scorer that is a function of the guide (so that a guide returned to the
    wrong place shows), bench.py's calls -- guide broadcast, interleaved shards, K steps, ONE gather, input order."""
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from crackling_amd import sharding
    if rank == 0:
        g_all = torch.from_numpy(np.random.default_rng(31).integers(0, 1 << 40, size=n, dtype=np.int64))
    else:
        g_all = torch.empty(n, dtype=torch.int64)
    dist.broadcast(g_all, 0)                                   # bench.py: the batch goes to every rank
    layout = sharding.ShardLayout(torch, n, world, chunk, "cpu")
    mine = g_all[layout.index_tensors[rank]].contiguous()      # ... and every rank takes its interleaved shard
    assert mine.numel() == layout.sizes[rank] == sharding.shard_size(n, world, rank, chunk)
    score = lambda g, k: ((g % 1000003).to(torch.float64) + k, (g >> 20).to(torch.float64) * 0.25 - k)   # noqa: E731
    out_m = torch.stack([score(mine, k)[0] for k in range(steps)]) if mine.numel() else torch.empty(steps, 0, dtype=torch.float64)
    out_c = torch.stack([score(mine, k)[1] for k in range(steps)]) if mine.numel() else torch.empty(steps, 0, dtype=torch.float64)
    gm, gc = sharding.gather_scores(dist, torch, layout, out_m, out_c, device="cpu")
    sizes = torch.tensor([float(mine.numel())], dtype=torch.float64)
    allsizes = [torch.empty_like(sizes) for _ in range(world)] if rank == 0 else None
    dist.gather(sizes, allsizes, dst=0)                         # (bench.py's per-rank record goes the same way)
    if rank == 0:
        assert gm.shape == (steps, n) and gc.shape == (steps, n)
        for k in range(steps):
            wm, wc = score(g_all, k)
            assert torch.equal(gm[k], wm) and torch.equal(gc[k], wc), k
        np.save(out_path, np.array([int(s.item()) for s in allsizes]))
    else:
        assert gm is None and gc is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,chunk", [(1_000_000, 4096), (3 * 4096 + 10, 4096), (8 * 4096, 4096), (5, 4096)])
def test_eight_ranks_with_configs3_layout(tmp_path, n, chunk):
    """BASELINE configs[3]'s shape on eight ranks (gloo): 1 M guides in interleaved chunks of 4096 -- 244 whole chunks and a
    tail of 576 that lands on rank 4 --, a batch that leaves ranks 4..7 without a single guide, one that divides evenly
    and one smaller than a chunk: shard sizes, one gather, input order."""
    from crackling_amd.sharding import shard_size
    world, out = 8, tmp_path / "sizes.npy"
    mp.spawn(_worker_layout, args=(world, _free_port(), n, chunk, 3, str(out)), nprocs=world, join=True)
    sizes = np.load(out)
    assert sizes.sum() == n and sizes.tolist() == [shard_size(n, world, r, chunk) for r in range(world)]
    if n == 1_000_000:
        assert sizes.tolist() == [31 * 4096] * 4 + [30 * 4096 + 576] + [30 * 4096] * 3
    if n == 3 * 4096 + 10:
        assert sizes.tolist() == [4096, 4096, 4096, 10, 0, 0, 0, 0]


def test_eight_ranks_oracle_scored_with_empty_shards(golden_uniform, tmp_path):
    """World 8 through the whole path -- image broadcast in pieces, attach, shards, gather -- with the oracle as scorer: 101
    guides in chunks of 16 leave rank 7 an empty shard."""
    import oracle_util as ou
    world = 8
    sigs = ou.encode(golden_uniform.guides)[:101]
    gp = tmp_path / "g.npy"; np.save(gp, sigs)
    out = tmp_path / "out.npz"
    mp.spawn(_worker, args=(world, _free_port(), str(golden_uniform.issl), str(gp), str(out), True, 16), nprocs=world, join=True)
    got = np.load(out)
    ix = ou.OracleIndex(golden_uniform.issl)
    mit, cfd = ix.score(sigs, 4, 75.0, "and")
    assert np.array_equal(got["mit"], mit) and np.array_equal(got["cfd"], cfd)
    assert np.array_equal(got["gm"], np.stack([mit, mit])) and np.array_equal(got["gc"], np.stack([cfd, cfd]))
