import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import gymnasium_amd
n = 64
a = gymnasium_amd.make_vec("Acrobot-v1", num_envs=n, output="torch")
b = gymnasium_amd.make_vec("Acrobot-v1", num_envs=n, output="torch")
a.reset(seed=3), b.reset(seed=3)
a.action_space.seed(1)
s0 = a.get_state()[0].copy()
out = a.rollout(1)
o, r, te, tr, _ = b.step(out["actions"][0])
sa, sb = a.get_state()[0], b.get_state()[0]
bad = (out["obs"][0] != o).any(dim=1).cpu().numpy()
print("bad lanes", np.flatnonzero(bad), "actions", out["actions"][0].cpu().numpy()[bad])
print("state bits equal:", np.array_equal(sa.view(np.uint64), sb.view(np.uint64)), "max |state diff|", np.abs(sa - sb).max())
for i in np.flatnonzero(bad)[:4]:
    print(i, "s0", s0[i], "\n   fused ", sa[i], "\n   step  ", sb[i], "\n   sin(theta2) libm", np.sin(sb[i][1]), "f32", np.float32(np.sin(sb[i][1])), "fused obs", out["obs"][0, i].cpu().numpy()[3], "step obs", o[i].cpu().numpy()[3])
    print("   theta2", sb[i][1], "as f32 then sin:", np.float32(np.sin(np.float64(np.float32(sb[i][1])))), " sinf:", np.sin(np.float32(sb[i][1])))
