"""CPU, world_size 2, gloo: the N>1 orchestration (contiguous guide shards, gather in input order).
The per-rank scorer is the oracle here (the checker) -- the HIP path needs a GPU; what is under test is the
sharding/gather logic that bench.py and multi-GPU callers use with index.score on `nccl`."""
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


def _worker(rank, world, port, issl, guides_path, out_path):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_util as ou
    from crackling_amd.sharding import score_sharded
    guides = np.load(guides_path)
    ix = ou.OracleIndex(issl)
    mit, cfd = score_sharded(dist, torch, lambda g: ix.score(g, 4, 75.0, "and", threads=1), guides)
    if rank == 0:
        np.savez(out_path, mit=mit, cfd=cfd)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_scores_equal_single_process(golden_uniform, tmp_path, world):
    import oracle_util as ou
    sigs = ou.encode(golden_uniform.guides)[:101]  # odd count: ragged shards
    gp = tmp_path / "g.npy"; np.save(gp, sigs)
    out = tmp_path / "out.npz"
    mp.spawn(_worker, args=(world, _free_port(), str(golden_uniform.issl), str(gp), str(out)), nprocs=world, join=True)
    got = np.load(out)
    ix = ou.OracleIndex(golden_uniform.issl)
    mit, cfd = ix.score(sigs, 4, 75.0, "and")
    assert np.array_equal(got["mit"], mit) and np.array_equal(got["cfd"], cfd)


def test_shard_bounds_cover_and_order():
    from crackling_amd.sharding import shard_bounds
    for n in (0, 1, 7, 8, 100, 1001):
        for w in (1, 2, 3, 8):
            cuts = [shard_bounds(n, w, r) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1
