import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import gymnasium_amd
n = 64
b = gymnasium_amd.make_vec("Acrobot-v1", num_envs=n, output="torch")
b.reset(seed=3)
ref = None
for T in (1, 2, 3, 4, 7, 8, 9, 16, 40):
    a = gymnasium_amd.make_vec("Acrobot-v1", num_envs=n, output="torch")
    a.reset(seed=3)
    a.action_space.seed(1)
    out = a.rollout(T)
    if ref is None:
        o, *_ = b.step(out["actions"][0])
        ref = o.clone()
    bad = (out["obs"][0] != ref).any(dim=1).cpu().numpy()
    # the same rollout again on a fresh env but with the actions given (other instantiation)
    c = gymnasium_amd.make_vec("Acrobot-v1", num_envs=n, output="torch")
    c.reset(seed=3)
    outc = c.rollout(T, actions=out["actions"])
    badc = (outc["obs"][0] != ref).any(dim=1).cpu().numpy()
    later = [(int((out["obs"][t] != outc["obs"][t]).any(dim=1).sum())) for t in range(T)]
    print(f"T={T}: sampled t=0 bad lanes {np.flatnonzero(bad).tolist()}, given-actions t=0 bad {np.flatnonzero(badc).tolist()}, sampled vs given per t: {later[:12]}")
    a.close(), c.close()
