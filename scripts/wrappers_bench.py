"""CartPole-v1 x N through ClipReward(NormalizeReward(NormalizeObservation(env))): wall time per step with the wrappers fused into the step
kernel (mi_set_step_epilogue) and as stand-alone passes.  `--mode fused|standalone --output torch|numpy`; run under
`rocprofv3 --kernel-trace --stats` to see the launches per step (profiles/r02_wrappers_*.txt)."""
import argparse
import json
import time

import numpy as np

import gymnasium_amd
from gymnasium_amd import wrappers as gw

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="fused")
ap.add_argument("--output", default="torch")
ap.add_argument("--num-envs", type=int, default=65536)
ap.add_argument("--steps", type=int, default=300)
a = ap.parse_args()
import torch

env = gymnasium_amd.make_vec("CartPole-v1", num_envs=a.num_envs, output=a.output)
if a.mode != "fused":
    env.FUSES_WRAPPERS = False
w = gw.ClipReward(gw.NormalizeReward(gw.NormalizeObservation(env)), -5.0, 5.0)
w.reset(seed=0)
env.action_space.seed(0)
acts = [env.action_space.sample() for _ in range(8)]
if a.output == "torch":
    acts = [torch.from_numpy(x).cuda() for x in acts]
for t in range(20):
    w.step(acts[t % 8])
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in range(a.steps):
    w.step(acts[t % 8])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print(json.dumps({"mode": a.mode, "output": a.output, "num_envs": a.num_envs, "us_per_wrapped_step": dt * 1e6, "env_steps_per_s": a.num_envs / dt}))
