#!/usr/bin/env python3
"""What the stand-alone wrapper passes cost next to a MuJoCo-kind step: Ant-v5, 65 536 envs, device tensors in / out,
ClipReward(NormalizeReward(NormalizeObservation(env))) against the bare env.  python scripts/r03/wrapped_mujoco_step.py [env_id] [num_envs]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import gymnasium_amd  # noqa: E402
from gymnasium_amd.wrappers import vector as W  # noqa: E402

env_id = sys.argv[1] if len(sys.argv) > 1 else "Ant-v5"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536


def timed(env, steps=12):
    env.reset(seed=0)
    act = torch.zeros((n,) + env.single_action_space.shape, dtype=torch.float32, device="cuda")
    for _ in range(3):
        env.step(act)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        env.step(act)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3


bare = gymnasium_amd.make_vec(env_id, num_envs=n, output="torch")
a = timed(bare)
bare.close()
env = gymnasium_amd.make_vec(env_id, num_envs=n, output="torch")
wrapped = W.ClipReward(W.NormalizeReward(W.NormalizeObservation(env)), -10.0, 10.0)
b = timed(wrapped)
wrapped.close()
print(f"{env_id} x{n}: bare step {a:.3f} ms, with NormalizeObservation + NormalizeReward + ClipReward (stand-alone passes) {b:.3f} ms: +{b - a:.3f} ms = +{100 * (b - a) / a:.2f} %")
