import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import gymnasium_amd
np.set_printoptions(precision=6, linewidth=200, suppress=True)
n, T, window = 64, 19, 5
os.environ["MI355ENV_MJ_SERIAL"] = "1"
ser = gymnasium_amd.make_vec("Humanoid-v5", num_envs=n)
del os.environ["MI355ENV_MJ_SERIAL"]
os.environ["MI355ENV_MJ_COOP"] = "1"
coop = gymnasium_amd.make_vec("Humanoid-v5", num_envs=n)
del os.environ["MI355ENV_MJ_COOP"]
ser.reset(seed=21), coop.reset(seed=21)
ser.action_space.seed(4)
for t in range(T):
    a = ser.action_space.sample()
    o1, r1, te1, tr1, i1 = ser.step(a)
    o2, r2, te2, tr2, i2 = coop.step(a)
    bad = np.flatnonzero(np.abs(o1 - o2).max(1) > 1e-6)
    if len(bad):
        i = bad[0]
        print("t", t, "env", i)
        print("ser  qvel", o1[i, 22:45]); print("coop qvel", o2[i, 22:45])
        print("ser  cvel b1..3", o1[i, 175:193]); print("coop cvel b1..3", o2[i, 175:193])
        s1, s2 = ser.get_state(), coop.get_state()
        print("state equal:", np.array_equal(s1[0][i], s2[0][i]), "max diff", np.abs(s1[0][i] - s2[0][i]).max(), "elapsed", s1[1][i], s2[1][i], "flags", s1[2][i], s2[2][i])
        print("state diff idx", np.flatnonzero(s1[0][i] != s2[0][i]))
        print("rng equal", np.array_equal(ser.get_rng_state()[i], coop.get_rng_state()[i]))
        print("reward", r1[i], r2[i], "info x", i1["x_position"][i], i2["x_position"][i])
        break
    if (t + 1) % window == 0:
        coop.set_state(*ser.get_state())
