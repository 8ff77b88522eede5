"""The two-role rollout kernel (rollout_duo_kernel) against the one-role kernel it replaces: identical trajectories, env-steps/s of both.
usage: python scripts/r04/duo_ab.py [num_envs] [env ids...]"""
import os
import sys
import time

sys.path.insert(0, ".")
import torch

import gymnasium_amd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
IDS = sys.argv[2:] or ["CartPole-v1", "Pendulum-v1", "Acrobot-v1", "MountainCar-v0", "MountainCarContinuous-v0"]
T = 128


def run(env_id, duo, launches=60, seed=3):
    os.environ["MI355ENV_ROLLOUT_DUO"] = "1" if duo else "0"
    env = gymnasium_amd.make_vec(env_id, num_envs=N, device=0, output="torch")
    env.reset(seed=seed)
    env.action_space.seed(seed)
    first = env.rollout(T)
    first = {k: v.clone() for k, v in first.items()}
    for _ in range(5):
        env.rollout(T)
    torch.cuda.synchronize()
    env.reset_statistics()
    t0 = time.perf_counter()
    for _ in range(launches):
        env.rollout(T)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = env.statistics()
    last = {k: v.clone() for k, v in env.rollout(T).items()}
    state = [torch.as_tensor(x).clone() for x in env.get_state()]
    rng = env.get_rng_state().copy()
    env.close()
    return first, last, state, rng, st, st["env_steps"] / dt


for env_id in IDS:
    a = run(env_id, False)
    b = run(env_id, True)
    same = all(torch.equal(a[0][k], b[0][k]) for k in a[0]) and all(torch.equal(a[1][k], b[1][k]) for k in a[1])
    same_state = all(torch.equal(x, y) for x, y in zip(a[2], b[2])) and (a[3] == b[3]).all()
    stats_equal = all(a[4][k] == b[4][k] for k in ("env_steps", "reset_steps", "episodes", "length_sum")) and abs(a[4]["return_sum"] - b[4]["return_sum"]) <= 1e-9 * abs(a[4]["return_sum"])
    print("%-26s one role %.4g  two roles %.4g env-steps/s (x%.3f)  trajectories %s  state+rng %s  totals %s" % (
        env_id, a[5], b[5], b[5] / a[5], "IDENTICAL" if same else "DIFFER", "identical" if same_state else "DIFFER", "equal" if stats_equal else "DIFFER"))
