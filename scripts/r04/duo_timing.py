"""Run a few launches of the two-role rollout kernel built with -DMI_DUO_TIMING (scripts/build_variant.py timing classic.hip -DMI_DUO_TIMING):
per role, the cycles spent working and waiting at the phase barrier.  MI355ENV_LIBRARY must point at the variant."""
import sys

sys.path.insert(0, ".")
import torch

import gymnasium_amd

env_id = sys.argv[1] if len(sys.argv) > 1 else "CartPole-v1"
env = gymnasium_amd.make_vec(env_id, num_envs=65536, device=0, output="torch")
env.reset(seed=0)
env.action_space.seed(0)
for _ in range(3):
    env.rollout(128)
torch.cuda.synchronize()
