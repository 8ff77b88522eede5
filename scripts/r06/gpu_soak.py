#!/usr/bin/env python3
"""Randomised soak of the HIP engine against the oracle (round 6: round 5's soak plus the on-device policy -- step(None) and action_space.sample() drawn by the engine; run on the GPU box): random env id, autoreset mode, batch size (ragged), TimeLimit, seed,
NumPy / device-tensor I/O, stepping mixed with fused rollouts, partial resets and set_state round trips -- classic control and ToyText compared bit for bit,
for a wall-clock budget.  Prints the number of configurations and transitions compared; exits non-zero at the first difference (with the configuration)."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
import gymnasium_amd  # noqa: E402
from oracle import oracle  # noqa: E402

def same_info(x, y, ignore=("t",)):
    """Recursive equality of info dicts: same keys, same dtypes / shapes / values (object arrays element by element); the episode clock `t` is wall time."""
    if isinstance(x, dict) or isinstance(y, dict):
        if not (isinstance(x, dict) and isinstance(y, dict)) or set(x) != set(y):
            return False
        return all(k in ignore or same_info(x[k], y[k], ignore) for k in x)
    if hasattr(x, "cpu"):
        x = x.cpu().numpy()
    x, y = np.asarray(x), np.asarray(y)
    if x.dtype != y.dtype or x.shape != y.shape:
        return False
    if x.dtype == object:
        return all((a is None and b is None) or (a is not None and b is not None and np.array_equal(np.asarray(a.cpu() if hasattr(a, "cpu") else a), np.asarray(b)))
                   for a, b in zip(x.ravel(), y.ravel()))
    return bool(np.array_equal(x, y))


IDS = ["CartPole-v1", "Pendulum-v1", "Acrobot-v1", "MountainCar-v0", "MountainCarContinuous-v0", "FrozenLake-v1", "FrozenLake8x8-v1", "CliffWalking-v1", "Taxi-v4", "Blackjack-v1"]


def main(budget_s=150.0, seed=0):
    import torch

    rng = np.random.default_rng(seed)
    t0, configs, transitions, device_policy = time.time(), 0, 0, 0
    while time.time() - t0 < budget_s:
        env_id = IDS[rng.integers(len(IDS))]
        mode = ["NextStep", "SameStep", "Disabled"][rng.integers(3)]
        n = int(rng.choice([1, 2, 63, 64, 65, 255, 257, 1000, int(rng.integers(1, 5000))]))
        max_steps = int(rng.choice([1, 2, 3, 7, 20, 50]))
        use_torch = bool(rng.integers(2))
        stats = bool(rng.integers(2))
        kw = dict(num_envs=n, autoreset_mode=mode, max_episode_steps=max_steps, record_episode_statistics=stats)
        cfg = (env_id, mode, n, max_steps, use_torch, stats)
        gpu = gymnasium_amd.make_vec(env_id, device=0, output="torch" if use_torch else "numpy", **kw)
        cpu = gymnasium_amd.make_vec(env_id, _engine_factory=oracle.engine_factory, **kw)
        def host(x):  # device tensors (or Blackjack's tuple of them) -> NumPy
            if isinstance(x, tuple):
                return tuple(host(y) for y in x)
            return x.cpu().numpy() if hasattr(x, "cpu") else x

        def same(x, y):
            if isinstance(x, tuple) or isinstance(y, tuple):
                return isinstance(x, tuple) and isinstance(y, tuple) and len(x) == len(y) and all(same(a, b) for a, b in zip(x, y))
            return np.array_equal(x, y)

        s = int(rng.integers(0, 2**40))
        assert same(host(gpu.reset(seed=s)[0]), cpu.reset(seed=s)[0]), (cfg, "reset")
        cpu.action_space.seed(s % 2**32)
        gpu.action_space.seed(s % 2**32)
        for t in range(int(rng.integers(5, 60))):
            what = rng.integers(10)
            before = cpu.action_space.np_random.bit_generator.state
            if what == 0 and use_torch and mode != "Disabled":  # a fused rollout with the on-device policy == the same steps one by one on the oracle
                T = int(rng.integers(1, 9))
                gpu.action_space.np_random.bit_generator.state = cpu.action_space.np_random.bit_generator.state
                out = gpu.rollout(T)
                for k in range(T):
                    a = cpu.action_space.sample()
                    c = cpu.step(a)
                    assert np.array_equal(out["actions"][k].cpu().numpy().reshape(a.shape), a), (cfg, t, k, "policy")
                    for name, j in (("obs", 0), ("rewards", 1), ("terminations", 2), ("truncations", 3)):
                        ref_j = np.stack(c[j], axis=-1) if isinstance(c[j], tuple) else c[j]  # (the rollout's observation rows: Blackjack's three integers as columns)
                        assert np.array_equal(out[name][k].cpu().numpy(), ref_j), (cfg, t, k, name)
                    if mode == "Disabled":
                        break
                transitions += T * n
                continue
            a = cpu.action_space.sample()
            if what in (3, 4):  # the engine's own draw of the same batch: the GPU env's stream is put where the oracle env's was before `a`
                cpu.action_space.np_random.bit_generator.state = before
                gpu.action_space.np_random.bit_generator.state = before
                assert np.array_equal(cpu.action_space.sample(), a)
            if what == 3:  # step(None): the policy drawn inside the step kernel
                g, c = gpu.step(None), cpu.step(a)
                if use_torch:  # (with NumPy batches step(None) is the two calls it stands for and keeps no copy of the batch)
                    assert np.array_equal(host(gpu.last_sampled_actions).reshape(a.shape), a), (cfg, t, "step(None) policy")
                device_policy += n
            elif what == 4:  # action_space.sample() served from the engine's draw-ahead block
                ag = gpu.action_space.sample()
                assert np.array_equal(host(ag), a) and host(ag).dtype == a.dtype, (cfg, t, "sample()")
                g, c = gpu.step(torch.from_numpy(a).cuda() if use_torch else a), cpu.step(a)
                device_policy += n
            else:
                g, c = gpu.step(torch.from_numpy(a).cuda() if use_torch else a), cpu.step(a)
            for j in range(4):
                assert same(host(g[j]), c[j]), (cfg, t, j)
            if not use_torch:  # (device-resident infos have a STATIC key set by design -- `episode` every step with its mask, batched `final_obs`: hip_vector_env.py
                assert same_info(g[4], c[4]), (cfg, t, "infos", sorted(g[4]), sorted(c[4]))  # _build_infos_device; the NumPy dict is the reference's)
            transitions += n
            done = c[2] | c[3]
            if mode == "Disabled" and done.any():
                m = done.copy()
                assert same(host(gpu.reset(options={"reset_mask": m})[0]), cpu.reset(options={"reset_mask": m})[0]), (cfg, t, "masked reset")
            elif what == 1 and mode != "Disabled":  # a partial reset thrown in
                m = rng.random(n) < 0.3
                if m.any():
                    assert same(host(gpu.reset(options={"reset_mask": m})[0]), cpu.reset(options={"reset_mask": m})[0]), (cfg, t, "partial reset")
            elif what == 2:  # checkpoint round trip through the C ABI
                st = gpu.get_state()
                ref = cpu.get_state()
                assert all(np.array_equal(x, y) for x, y in zip(st, ref)), (cfg, t, "state")
                gpu.set_state(*st)
        assert np.array_equal(gpu.get_rng_state(), cpu.get_rng_state()), (cfg, "generators")
        gpu.close(), cpu.close()
        configs += 1
    print(f"soak ok: {configs} random configurations, {transitions} transitions compared bit for bit ({device_policy} of them with the policy drawn on the device) in {time.time() - t0:.0f} s")


def soak_shared(budget_s=60.0, seed=1):
    """rng="shared" (MI_CFG_SHARED_RNG): ragged batches across workgroup boundaries, short TimeLimits (bursts of simultaneous re-draws), custom bounds."""
    rng = np.random.default_rng(seed)
    t0, configs, transitions = time.time(), 0, 0
    while time.time() - t0 < budget_s:
        n = int(rng.choice([1, 2, 255, 256, 257, 511, 513, 1000, int(rng.integers(1, 20000))]))
        kw = dict(num_envs=n, rng="shared", max_episode_steps=int(rng.choice([1, 2, 5, 13, 40])), sutton_barto_reward=bool(rng.integers(2)))
        gpu = gymnasium_amd.make_vec("CartPole-v1", device=0, **kw)
        cpu = gymnasium_amd.make_vec("CartPole-v1", _engine_factory=oracle.engine_factory, **kw)
        s = int(rng.integers(0, 2**62))
        opts = None if rng.integers(2) else {"low": -float(rng.uniform(0, 0.2)), "high": float(rng.uniform(0, 0.2))}
        assert np.array_equal(gpu.reset(seed=s, options=opts)[0], cpu.reset(seed=s, options=opts)[0]), (kw, "reset")
        cpu.action_space.seed(s % 2**32)
        for t in range(int(rng.integers(3, 50))):
            a = cpu.action_space.sample()
            g, c = gpu.step(a), cpu.step(a)
            for j in range(4):
                assert np.array_equal(g[j], c[j]) and g[j].dtype == c[j].dtype, (kw, t, j)
            if rng.integers(15) == 0:
                assert np.array_equal(gpu.reset()[0], cpu.reset()[0]), (kw, t, "reset without seed")
        assert np.array_equal(gpu.get_rng_state()[0], cpu.get_rng_state()[0]), (kw, "generator")
        transitions += n * (t + 1)
        gpu.close(), cpu.close()
        configs += 1
    print(f"shared-generator soak ok: {configs} random configurations, {transitions} transitions compared bit for bit in {time.time() - t0:.0f} s")


def soak_mujoco(budget_s=60.0, seed=2, atol=2e-8):
    """The eleven MuJoCo kinds, teacher-forced from the oracle's state every step (contact dynamics amplify last-bit differences): observations, rewards and the
    physics state within `atol`, flags equal; random batch sizes, modes, TimeLimits and float32 / float64 action rows."""
    ids = ["HalfCheetah-v5", "Ant-v5", "Humanoid-v5", "HumanoidStandup-v5", "Hopper-v5", "Walker2d-v5", "InvertedPendulum-v5", "InvertedDoublePendulum-v5",
           "Reacher-v5", "Swimmer-v5", "Pusher-v5"]
    rng = np.random.default_rng(seed)
    t0, configs, transitions, worst = time.time(), 0, 0, 0.0
    while time.time() - t0 < budget_s:
        env_id = ids[rng.integers(len(ids))]
        n = int(rng.choice([1, 3, 15, 16, 17, 33, 100, int(rng.integers(1, 400))]))
        kw = dict(num_envs=n, autoreset_mode=["NextStep", "SameStep"][rng.integers(2)], max_episode_steps=int(rng.choice([3, 10, 40])))
        gpu = gymnasium_amd.make_vec(env_id, device=0, **kw)
        cpu = gymnasium_amd.make_vec(env_id, _engine_factory=oracle.engine_factory, **kw)
        s = int(rng.integers(0, 2**40))
        og, oc = gpu.reset(seed=s)[0], cpu.reset(seed=s)[0]
        assert np.abs(og - oc).max() <= 1e-9, (env_id, kw, "reset")
        cpu.action_space.seed(s % 2**32)
        for t in range(int(rng.integers(2, 14))):
            a = cpu.action_space.sample()
            if rng.integers(3) == 0:
                a = a.astype(np.float64) * 0.9
            gpu.set_state(*cpu.get_state())
            g, c = gpu.step(a), cpu.step(a)
            d = max(float(np.abs(g[0] - c[0]).max()), float(np.abs(g[1] - c[1]).max()))
            worst = max(worst, d)
            assert d <= atol and np.array_equal(g[2], c[2]) and np.array_equal(g[3], c[3]), (env_id, kw, t, d)
        transitions += n * (t + 1)
        gpu.close(), cpu.close()
        configs += 1
    print(f"MuJoCo soak ok: {configs} random configurations, {transitions} transitions within {atol:g} (worst {worst:.2e}) in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "shared":
        soak_shared(float(sys.argv[2]) if len(sys.argv) > 2 else 60.0)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "mujoco":
        soak_mujoco(float(sys.argv[2]) if len(sys.argv) > 2 else 60.0)
        sys.exit(0)
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 150.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
