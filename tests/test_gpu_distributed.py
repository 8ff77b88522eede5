"""-m gpu: the RCCL leg of the multi-GPU path on the one GPU a test box has.

The N > 1 layout (one process per GPU, env-sharded, ONE metric all-reduce: gymnasium_amd/distributed.py) is covered by world_size-2/3
gloo tests on CPU (tests/test_multiprocess.py).  What those cannot touch is the `nccl` (= RCCL on ROCm) backend itself; a
world_size-1 process group at least initialises RCCL on the device, runs the same all_reduce calls bench.py issues, and checks that
the engine's stream and RCCL's coexist.  No scaling curve is measured here.

The group lives in a CHILD process, like a bench.py rank does: initialising and destroying an RCCL communicator inside the long-lived
pytest process was measured (round 2) to make a later, unrelated HIP call of the same process abort.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, socket, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
import gymnasium_amd
from gymnasium_amd import distributed as gd

with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{{port}}", rank=0, world_size=1, device_id=dev)
env = gymnasium_amd.make_vec("CartPole-v1", num_envs=4096, device=0, output="torch")
env.reset(seed=0); env.action_space.seed(0); env.rollout(32)
st = env.statistics()
# the two collectives of bench.py (SUM of the counters, MAX of the elapsed time), issued for real on RCCL
t = torch.tensor([float(st[k]) for k in gd.STAT_KEYS], dtype=torch.float64, device=dev); before = t.clone()
dist.all_reduce(t, op=dist.ReduceOp.SUM)
e = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(e, op=dist.ReduceOp.MAX)
dist.barrier(); torch.cuda.synchronize()
assert torch.equal(t, before) and float(e[0]) == 1.25
red = gd.reduce_statistics(st, elapsed_s=0.5, device=dev)
assert red["env_steps"] == st["env_steps"] and red["elapsed_s"] == 0.5
assert st["env_steps"] + st["reset_steps"] == 4096 * 32
# the rank / device census of the bench line, with its collectives forced at world size 1: all-reduce of ones + all_gather_object on RCCL
cen = gd.census(0, 0, device=dev, force_collective=True)
assert cen["ranks"] == 1 and cen["distinct_devices"] == 1 and cen["devices"][0]["rank"] == 0 and len(cen["devices"][0]["uuid"]) > 8, cen
env.close()
dist.destroy_process_group()
print("RCCL_OK")
"""


def test_rccl_world_size_one_metric_reduction():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
