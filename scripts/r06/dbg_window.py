"""Debug: why does bench.mujoco_window_check disagree on the GPU?  (round 6, call E)"""
import sys, os, warnings
warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import bench
import gymnasium_amd
from oracle import oracle

env_id = sys.argv[1] if len(sys.argv) > 1 else "Ant-v5"
cfg = bench.Config(env_id, 4096, 4, 0, 0, None)
for _ in range(8):
    cfg.launch()
env, eng, N, T = cfg.env, cfg.eng, cfg.N, cfg.inner
idx = np.arange(0, N, 16)[:256]
state, elapsed, flags = env.get_state()
words = env.get_rng_state()
bufs = cfg.alloc_trajectory()
cfg.launch(bufs)
acts, obs, rew, te, tr = tuple(b.cpu().numpy() for b in bufs[0])
print("gpu: nan obs", int(np.isnan(obs).sum()), "nan rew", int(np.isnan(rew).sum()), "te", int(te.sum()), "tr", int(tr.sum()), "flags", np.bincount(flags, minlength=4), "elapsed max", int(elapsed.max()))
print("state nan", int(np.isnan(state).sum()), "state shape", state.shape, "acts dtype", acts.dtype, acts.shape)
c = gymnasium_amd.make_vec(env_id, num_envs=len(idx), _engine_factory=oracle.engine_factory)
c.reset(seed=0)
c._engine.seed(np.ascontiguousarray(words[idx]), None)
c.set_state(state[idx], elapsed[idx], flags[idx])
o2, r2 = np.zeros((T, len(idx), eng.obs_dim)), np.zeros((T, len(idx)))
te2, tr2 = np.zeros((T, len(idx)), np.bool_), np.zeros((T, len(idx)), np.bool_)
c._engine.rollout(T, np.ascontiguousarray(acts[:, idx]), None, o2, r2, te2, tr2)
print("oracle: nan obs", int(np.isnan(o2).sum()), "nan rew", int(np.isnan(r2).sum()), "te", int(te2.sum()), "tr", int(tr2.sum()))
d = np.abs(obs[:, idx] - o2)
print("max diff per step", [float(np.nanmax(d[t])) for t in range(T)])
bad = np.argwhere((te[:, idx] != te2) | (tr[:, idx] != tr2))
print("flag mismatches", len(bad), bad[:8].tolist())
if len(bad):
    t, j = bad[0]
    g = idx[j]
    print("robot", g, "flags", flags[g], "elapsed", elapsed[g], "gpu te/tr", te[:, g], tr[:, g], "oracle", te2[:, j], tr2[:, j])
    print("gpu rew", rew[:, g], "oracle rew", r2[:, j])
    print("gpu obs[:5]", obs[t, g, :5], "oracle", o2[t, j, :5])
st2 = c.get_state()
print("oracle state after == gpu state after?", float(np.nanmax(np.abs(st2[0] - env.get_state()[0][idx]))))
