"""Where the per-launch step() goes at 65 536 sub-environments: in-graph time per step (host out of the loop) for a few configurations, next to
the box's dispatch floor (scripts/r04/dispatch_floor.py).  usage: python scripts/r04/step_latency.py [num_envs]"""
import sys
import time

sys.path.insert(0, ".")
import torch

import gymnasium_amd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536


def in_graph(env_id, steps=32, reps=60, **kw):
    env = gymnasium_amd.make_vec(env_id, num_envs=N, device=0, output="torch", copy=False, **kw)
    env.reset(seed=0)
    a = torch.randint(0, 2, (N,), device="cuda") if env._discrete else (torch.rand((N, env._engine.act_dim), device="cuda") * 0.8 - 0.4)
    for _ in range(30):
        env.step(a)
    g = env.capture_steps(actions=a, steps=steps)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (reps * steps) * 1e6
    env.close()
    return dt


for env_id, kw in [("CartPole-v1", {}), ("CartPole-v1", {"fast_math": True}), ("CartPole-v1", {"autoreset_mode": "SameStep"}),
                   ("CartPole-v1", {"max_episode_steps": 1000000, "fast_math": True}), ("MountainCar-v0", {}), ("Pendulum-v1", {}), ("Acrobot-v1", {}),
                   ("Taxi-v3", {}), ("Hopper-v5", {})]:
    try:
        print("%-26s %-44s %6.2f us per step (in a 32-step graph)" % (env_id, kw, in_graph(env_id, **kw)))
    except Exception as e:  # noqa: BLE001
        print(env_id, kw, "failed:", repr(e)[:200])
