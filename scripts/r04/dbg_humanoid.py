"""Which sub-environments differ between the cooperative and the one-lane Humanoid kernels, and is their wavefront partner resetting?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import gymnasium_amd

n, T, window = 64, 20, 5
os.environ["MI355ENV_MJ_SERIAL"] = "1"
ser = gymnasium_amd.make_vec("Humanoid-v5", num_envs=n)
del os.environ["MI355ENV_MJ_SERIAL"]
os.environ["MI355ENV_MJ_COOP"] = "1"
coop = gymnasium_amd.make_vec("Humanoid-v5", num_envs=n)
del os.environ["MI355ENV_MJ_COOP"]
o1, _ = ser.reset(seed=21)
o2, _ = coop.reset(seed=21)
ser.action_space.seed(4)
prev_done = np.zeros(n, bool)
for t in range(T):
    a = ser.action_space.sample()
    o1, r1, te1, tr1, _ = ser.step(a)
    o2, r2, te2, tr2, _ = coop.step(a)
    bad = np.flatnonzero(np.abs(o1 - o2).max(1) > 1e-6)
    print(f"t={t} resetting now: {np.flatnonzero(prev_done).tolist()} terminated now: {np.flatnonzero(te1).tolist()} (coop {np.flatnonzero(te2).tolist()}) bad envs: {bad.tolist()}")
    for i in bad:
        d = np.abs(o1[i] - o2[i])
        print(f"   env {i}: partner {i ^ 1} resetting={bool(prev_done[i ^ 1])} self resetting={bool(prev_done[i])} ncols {(d > 1e-6).sum()} max {d.max():.3e} first cols {np.flatnonzero(d > 1e-6)[:8].tolist()}")
    prev_done = te1 | tr1
    if (t + 1) % window == 0:
        coop.set_state(*ser.get_state())
