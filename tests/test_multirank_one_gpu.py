"""The N > 1 code path of bench.py rehearsed on the ONE GPU of the test box: `python -m torch.distributed.run
--nproc-per-node 2 bench.py --gpus 2` with both ranks on cuda:0 and the collectives over gloo (RCCL refuses two ranks
on one device) -- the same script, the same crackling_amd/sharding.py calls (image broadcast in pieces, attach,
interleaved shards, one gather, input order) the driver's 8-GPU run goes through; the gathered scores must equal the
single-process result bit for bit.  (The RCCL transport itself is exercised with one rank by
test_gpu_parity.py::test_node_sharding_and_rccl_broadcast and by `torch.distributed.run --nproc-per-node 1`.)"""
import json
import os
import pathlib
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_two_ranks_on_one_gpu_through_bench(tmp_path):
    args = ["--steps", "2", "--warmup", "1", "--sites", "3000000", "--guides", "30000", "--chunk", "1024",
            "--no-cpu-baseline", "--no-extras", "--spinup-ms", "0"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", *args, "--dump-scores", str(tmp_path / "one.npz")],
                         capture_output=True, text=True, env=env, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    env2 = dict(env, ISSL_BENCH_DEVICE="0", ISSL_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", "2", *args,
                          "--dump-scores", str(tmp_path / "two.npz")], capture_output=True, text=True, env=env2, timeout=600)
    assert two.returncode == 0, two.stderr[-3000:]
    line = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and len(line["per_rank"]) == 2
    assert sum(r["guides"] for r in line["per_rank"]) == 30000 and line["setup_s"]["broadcast_s"] > 0
    a, b = np.load(tmp_path / "one.npz"), np.load(tmp_path / "two.npz")
    assert np.array_equal(a["guides"], b["guides"])
    assert np.array_equal(a["mit"].view(np.uint64), b["mit"].view(np.uint64))
    assert np.array_equal(a["cfd"].view(np.uint64), b["cfd"].view(np.uint64))
    assert (a["mit"] < 100).any()


def test_one_rank_over_rccl_equals_the_plain_run(tmp_path):
    """Every RCCL call site of bench.py's N > 1 path -- init_process_group with a device id, the image broadcast of device
    tensors in pieces, attach, the guide broadcast, the gather of device tensors, barrier, max-reduce -- executed with the
    `nccl` backend itself and world size 1 (what the one GPU of the box allows), no ISSL_BENCH_BACKEND: the line must carry
    the collectives' diagnostics (per-rank broadcast time, image bytes, RCCL version) and the scores must equal the plain
    single-process run bit for bit."""
    args = ["--steps", "2", "--warmup", "1", "--sites", "3000000", "--guides", "30000", "--chunk", "1024",
            "--no-cpu-baseline", "--no-extras", "--spinup-ms", "0"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("ISSL_BENCH_BACKEND", None); env.pop("ISSL_BENCH_DEVICE", None)
    one = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", *args, "--dump-scores", str(tmp_path / "plain.npz")],
                         capture_output=True, text=True, env=env, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    plain = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert plain["collectives"] is None and plain["per_rank"] is None
    rccl = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                           "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", "1", *args,
                           "--dump-scores", str(tmp_path / "rccl.npz")], capture_output=True, text=True, env=dict(env, MASTER_ADDR="127.0.0.1"), timeout=600)
    assert rccl.returncode == 0, rccl.stderr[-3000:]
    line = json.loads([l for l in rccl.stdout.splitlines() if l.startswith("{")][-1])
    col = line["collectives"]
    assert col["backend"] == "nccl" and col["world"] == 1 and col["rccl_version"] and col["image_bytes"] == line["config"]["image_bytes"]
    assert line["setup_s"]["broadcast_s"] > 0 and col["broadcast_GBps"] > 0
    assert len(line["per_rank"]) == 1 and line["per_rank"][0]["guides"] == 30000 and line["per_rank"][0]["image_bytes"] == col["image_bytes"]
    assert line["per_rank"][0]["broadcast_s"] == pytest.approx(line["setup_s"]["broadcast_s"])
    assert "entering the image broadcast" in rccl.stderr and "image attached" in rccl.stderr
    a, b = np.load(tmp_path / "plain.npz"), np.load(tmp_path / "rccl.npz")
    assert np.array_equal(a["guides"], b["guides"])
    assert np.array_equal(a["mit"].view(np.uint64), b["mit"].view(np.uint64))
    assert np.array_equal(a["cfd"].view(np.uint64), b["cfd"].view(np.uint64))
