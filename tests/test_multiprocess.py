"""N>1 path on CPU: world_size-2 (and 3, uneven shards) gloo ranks, launched exactly like the driver launches bench.py
(python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT
from gymnasium_amd import distributed as gd


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(cmd_for_port, **kw):
    """Launch with a fresh rendezvous port; a port another (parallel) test grabbed between _free_port() and torchrun's bind is retried."""
    for _ in range(4):
        p = subprocess.run(cmd_for_port(_free_port()), capture_output=True, text=True, **kw)
        if p.returncode == 0 or not any(m in p.stderr for m in ("Address already in use", "EADDRINUSE", "address already in use")):
            break
    return p


def test_shard_range_partitions_exactly():
    for total in (1, 7, 8, 65536, 262144):
        for world in (1, 2, 3, 8):
            if world > total:
                continue
            blocks = [gd.shard_range(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [h - l for l, h in blocks]
            assert max(sizes) - min(sizes) <= 1
    assert gd.shard_range(262144, 3, 8) == (98304, 131072)  # configs[4]: 8 x 32768
    with pytest.raises(ValueError):
        gd.shard_range(8, 2, 2)


@pytest.mark.parametrize("env_id,world,total", [("CartPole-v1", 2, 64), ("Pendulum-v1", 2, 32), ("CartPole-v1", 3, 50)])
def test_sharded_ranks_reproduce_single_process(tmp_path, env_id, world, total):
    out = tmp_path / "result.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = lambda port: [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",  # noqa: E731
                        "--master-port", str(port), os.path.join(ROOT, "tests", "_mp_worker.py"), env_id, str(total), "60", str(out)]
    p = _torchrun(cmd, env=env, cwd=ROOT, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    res = json.load(open(out))
    assert res["world"] == world
    assert res["ok_traj"], "shard trajectories differ from the single-process batch"
    assert res["ok_stats"], "all-reduced statistics differ from the single-process totals"
    assert res["elapsed_max"] == float(world)  # MAX over ranks of (1 + rank)
    assert res["env_steps"] > 0
