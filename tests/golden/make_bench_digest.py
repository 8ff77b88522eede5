#!/usr/bin/env python3
"""The known answer of bench.py's first timed launch, computed BY THE REFERENCE ITSELF (build container only; ~4 minutes):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_bench_digest.py

BASELINE.json configs[1] at its exact shape: gymnasium.make_vec("CartPole-v1", num_envs=65536, vectorization_mode="sync") -- 65 536 scalar
CartPoleEnv objects behind TimeLimit in one SyncVectorEnv (vector/sync_vector_env.py:187-337) -- reset(seed=0), action_space.seed(0), then 128 x
step(action_space.sample()).  Writes bench_digest.json: the sha256 over the bytes of (actions int64 [T,N], observations float32 [T,N,4],
rewards float64 [T,N], terminated bool [T,N], truncated bool [T,N]) -- what bench.trajectory_digest() computes over the trajectory the rollout
kernel writes to HBM -- plus digests of strided slices for debugging.  tests/test_bench_digest.py requires the oracle to reproduce it on the CPU and
tests/test_gpu_bench_contract.py requires `output_sha256` of the bench line to equal it.

    python tests/golden/make_bench_digest.py Pendulum-v1      (then Acrobot-v1, MountainCarContinuous-v0; ONE AT A TIME: they share the output file)

adds BASELINE.json configs[2] at ITS exact shape (65 536 sub-environments, 128 steps, the same seeds) to bench_digest_configs2.json: the secondary lines of
the bench (scripts/bench_extras.py) report the digest of the same known-answer rollout."""
import hashlib
import json
import os
import sys
import time

REF = os.environ.get("GYM_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import numpy as np  # noqa: E402

import gymnasium as gym  # noqa: E402

assert int(np.__version__.split(".")[0]) >= 2
OUT = os.path.dirname(os.path.abspath(__file__))


def digest(arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).view(np.uint8).reshape(-1).data)
    return h.hexdigest()


def run_in_chunks(env_id, N, T, chunk):
    """The same rollout for ids whose 65 536 scalar envs do not fit in memory at once (Taxi: every env builds its own 500 x 6 transition dict, ~1 MB): the
    sub-environments are independent and sub-environment g is seeded 0 + g (sync_vector_env.py:207-208), so SyncVectorEnv over `chunk` of them at a time, reset
    with the seed LIST [lo, lo + 1, ...] and fed the columns [lo, lo + chunk) of the actions that the FULL batched space (the same `batch_space(single, N)` object
    SyncVectorEnv builds, seeded 0) samples, gives the rows the one big SyncVectorEnv would."""
    from gymnasium.vector.utils import batch_space

    probe = gym.make(env_id)
    space = batch_space(probe.action_space, N)
    probe.close()
    space.seed(0)
    actions = [space.sample() for _ in range(T)]
    cols = {k: [] for k in ("obs", "rew", "te", "tr")}
    t0 = time.time()
    for lo in range(0, N, chunk):
        env = gym.make_vec(env_id, num_envs=chunk, vectorization_mode="sync")
        env.reset(seed=list(range(lo, lo + chunk)))
        o_, r_, d_, u_ = [], [], [], []
        for t in range(T):
            o, r, d, u, _ = env.step(actions[t][lo:lo + chunk])
            if isinstance(o, tuple):
                o = np.stack([np.asarray(x, dtype=np.int64) for x in o], axis=-1)
            o_.append(o.copy()), r_.append(r.copy()), d_.append(d.copy()), u_.append(u.copy())
        env.close()
        cols["obs"].append(np.stack(o_)), cols["rew"].append(np.stack(r_)), cols["te"].append(np.stack(d_)), cols["tr"].append(np.stack(u_))
        print(f"sub-environments {lo}..{lo + chunk - 1} at {time.time() - t0:.0f} s", flush=True)
    return [a.copy() for a in actions], list(np.concatenate(cols["obs"], axis=1)), list(np.concatenate(cols["rew"], axis=1)), list(np.concatenate(cols["te"], axis=1)), \
        list(np.concatenate(cols["tr"], axis=1))


def main(env_id="CartPole-v1", N=65536, T=128, out_name="bench_digest.json", chunk=None):
    t0 = time.time()
    if chunk:
        acts, obs, rew, te, tr = run_in_chunks(env_id, N, T, chunk)
    else:
        env = gym.make_vec(env_id, num_envs=N, vectorization_mode="sync")
        print(f"{N} scalar envs built in {time.time() - t0:.0f} s", flush=True)
        env.reset(seed=0)
        env.action_space.seed(0)
        acts, obs, rew, te, tr = [], [], [], [], []
        for t in range(T):
            a = env.action_space.sample()
            o, r, d, u, _ = env.step(a)
            if isinstance(o, tuple):  # Blackjack: Tuple(Discrete, Discrete, Discrete) batches to a tuple of arrays; the engine's observation row is the three int64
                o = np.stack([np.asarray(x, dtype=np.int64) for x in o], axis=-1)
            acts.append(a.copy()), obs.append(o.copy()), rew.append(r.copy()), te.append(d.copy()), tr.append(u.copy())
            if t % 16 == 0:
                print(f"step {t} at {time.time() - t0:.0f} s", flush=True)
    traj = tuple(np.stack(x) for x in (acts, obs, rew, te, tr))
    assert traj[0].dtype in (np.int64, np.float32) and traj[1].dtype in (np.float32, np.int64) and traj[2].dtype == np.float64 and traj[3].dtype == np.bool_ and traj[4].dtype == np.bool_
    out = {"what": f"gymnasium {gym.__version__} make_vec({env_id!r}, {N}, 'sync'), reset(seed=0), action_space.seed(0), {T} x step(sample()); "
                   "sha256 over (actions, obs, rewards, terminated, truncated) bytes, time-major",
           "numpy": np.__version__, f"{env_id}:{N}:{T}:rank0": digest(traj),
           f"{env_id}:{N}:{T}:rank0:first_1024_envs": digest(tuple(np.ascontiguousarray(x[:, :1024]) for x in traj)),
           f"{env_id}:{N}:{T}:rank0:every_64th_env": digest(tuple(np.ascontiguousarray(x[:, ::64]) for x in traj)),
           "episodes_finished": int((traj[3] | traj[4]).sum()), "reward_sum": float(traj[2].sum())}
    path = os.path.join(OUT, out_name)
    if out_name != "bench_digest.json":  # BASELINE.json configs[2]: one shared file, one entry set per env id
        prev = json.load(open(path)) if os.path.exists(path) else {}
        prev.update({k: v for k, v in out.items() if k.startswith(env_id)})
        prev[env_id + ":what"], prev["numpy"] = out["what"], out["numpy"]
        prev[env_id + ":episodes_finished"], prev[env_id + ":reward_sum"] = out["episodes_finished"], out["reward_sum"]
        out = prev
    with open(path + f".{os.getpid()}.tmp", "w") as f:
        json.dump(out, f, indent=1)
    os.replace(path + f".{os.getpid()}.tmp", path)
    print(json.dumps(out, indent=1), f"\n{time.time() - t0:.0f} s")


if __name__ == "__main__":
    if len(sys.argv) > 1:  # python make_bench_digest.py Pendulum-v1 [chunk] -> bench_digest_configs2.json (BASELINE.json configs[2]: ~5-10 minutes per id, one at a time)
        main(sys.argv[1], out_name="bench_digest_configs2.json", chunk=int(sys.argv[2]) if len(sys.argv) > 2 else None)
    else:
        main()
