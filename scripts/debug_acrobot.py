"""Where do the fused Acrobot rollout and stepping part ways?  (round-3 debugging aid; run on the GPU box)"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import gymnasium_amd

n, T = 64, 40
a = gymnasium_amd.make_vec("Acrobot-v1", num_envs=n, output="torch")
b = gymnasium_amd.make_vec("Acrobot-v1", num_envs=n, output="torch")
c = gymnasium_amd.make_vec("Acrobot-v1", num_envs=n, output="torch")
a.reset(seed=3), b.reset(seed=3), c.reset(seed=3)
a.action_space.seed(1)
out = a.rollout(T)
outc = c.rollout(T, actions=out["actions"])
sa = a.get_state()
for t in range(T):
    o, r, te, tr, _ = b.step(out["actions"][t])
    bad = (out["obs"][t] != o).any(dim=1).cpu().numpy()
    badc = (outc["obs"][t] != o).any(dim=1).cpu().numpy()
    if bad.any() or badc.any():
        i = int(np.flatnonzero(bad | badc)[0])
        print(f"t={t}: {int(bad.sum())} lanes of the sampled rollout, {int(badc.sum())} of the given-actions rollout differ from stepping; first lane {i}")
        print(" stepping obs ", o[i].cpu().numpy(), float(r[i]), bool(te[i]), bool(tr[i]))
        print(" rollout  obs ", out["obs"][t, i].cpu().numpy(), float(out["rewards"][t, i]), bool(out["terminations"][t, i]))
        print(" rolloutC obs ", outc["obs"][t, i].cpu().numpy(), float(outc["rewards"][t, i]), bool(outc["terminations"][t, i]))
        if t:
            print(" previous step: term", bool(out["terminations"][t - 1, i]), "trunc", bool(out["truncations"][t - 1, i]), "obs", out["obs"][t - 1, i].cpu().numpy())
        print(" diff (rollout - stepping)", (out["obs"][t, i] - o[i]).cpu().numpy())
        break
else:
    print("no difference in", T, "steps")
