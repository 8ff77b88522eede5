"""-m gpu: states no trajectory from reset() reaches, put there with set_state() -- CartPole's RARE lanes and Pendulum's far angles against the oracle, bit for bit.

CartPoleT::step (gymnasium_amd/csrc/envs_classic.h; cartpole.py:164-226) takes a short path when the pole angle is inside the short sincos routine's
range (|theta| < ~0.855) and the two divisions' operands are inside the three-FMA range; every other lane redoes its accelerations through the general
routines in an out-of-line function (`general_accel`).  Episodes end at |theta| > 0.2095, so no trajectory that starts from reset() ever reaches that
code: the states here are put there with set_state() -- angles up to 1e5 rad, angular velocities up to 1e8, the range's own boundary from both sides
-- and stepped by BOTH kernels that contain the code: step() (step_kernel) and rollout() (rollout_duo_kernel, whose first step then runs on them).

Pendulum's angle is never wrapped (pendulum.py:139-150 keeps th; only the cost normalises it), so the exact fmod by the rounded reciprocal and the
general sin / cos are checked far beyond the 80 rad an episode can reach -- up to 1e8, the edge of the range the restated libm routines cover
(|x| < 1.05e8, docs/classic_kernels.md: beyond it glibc switches to Payne-Hanek and the kernels defer to ocml; with angles of 1e12 this test
fails in the float64 reward of the second step, as that note says it would)."""
import numpy as np
import pytest

import gymnasium_amd
from wide_states import EDGE, wide_acrobot_states, wide_mountaincar_states, wide_pendulum_states, wide_states  # noqa: F401

pytestmark = pytest.mark.gpu


def pair(n, output="numpy", **kw):
    from oracle import oracle

    gpu = gymnasium_amd.make_vec("CartPole-v1", num_envs=n, device=0, output=output, **kw)
    cpu = gymnasium_amd.make_vec("CartPole-v1", num_envs=n, _engine_factory=oracle.engine_factory, **kw)
    first = gpu.reset(seed=5)[0]
    assert np.array_equal(first.cpu().numpy() if output == "torch" else first, cpu.reset(seed=5)[0])
    return gpu, cpu


@pytest.mark.parametrize("n", [1000, 65536])
def test_step_kernel_on_rare_lanes(n):
    gpu, cpu = pair(n, autoreset_mode="Disabled", max_episode_steps=10**6)
    zeros = np.zeros(n, dtype=np.int32)
    for seed in range(3):
        s = wide_states(n, seed)
        a = np.random.default_rng(100 + seed).integers(0, 2, n)
        for env in (gpu, cpu):
            env.reset(seed=seed)
            env.set_state(s, zeros, np.zeros(n, dtype=np.uint8))
        g, c = gpu.step(a), cpu.step(a)
        for j, name in enumerate(("obs", "rewards", "terminations", "truncations")):
            assert np.array_equal(g[j], c[j], equal_nan=True), (seed, name, s[np.flatnonzero((g[j] != c[j]).reshape(n, -1).any(axis=1))[:4]])
        sg, sc = gpu.get_state(), cpu.get_state()
        assert np.array_equal(sg[0].view(np.uint64), sc[0].view(np.uint64)), (seed, "state words")
        assert c[2].mean() > 0.5  # (most of these states are beyond the termination thresholds, as intended)
    gpu.close(), cpu.close()


@pytest.fixture(params=[1, 0], ids=["two_roles", "one_role"])
def rollout_kernel(request, monkeypatch):
    """Both kernels behind rollout(): rollout_duo_kernel (the default for this configuration) and the one-role rollout_kernel it replaced (engine.hip)."""
    monkeypatch.setenv("MI355ENV_ROLLOUT_DUO", str(request.param))
    return request.param


@pytest.mark.parametrize("n", [1000, 65536])
def test_rollout_kernel_on_rare_lanes(n, rollout_kernel):
    """rollout(T) right after set_state(): the rollout kernel's first step runs on the wide states (then NEXT_STEP resets the finished sub-environments)."""
    T = 8
    gpu, cpu = pair(n, output="torch")
    zeros = np.zeros(n, dtype=np.int32)
    for seed in range(2):
        s = wide_states(n, 50 + seed)
        for env in (gpu, cpu):
            env.reset(seed=seed)
            env.action_space.seed(9 + seed)
            env.set_state(s, zeros, np.zeros(n, dtype=np.uint8))
        out = gpu.rollout(T)
        host = lambda x: x.cpu().numpy()
        for k in range(T):
            a = cpu.action_space.sample()
            c = cpu.step(a)
            assert np.array_equal(host(out["actions"][k]).reshape(a.shape), a), (seed, k, "policy")
            for name, j in (("obs", 0), ("rewards", 1), ("terminations", 2), ("truncations", 3)):
                assert np.array_equal(host(out[name][k]), c[j], equal_nan=True), (seed, k, name)
        sg, sc = gpu.get_state(), cpu.get_state()
        assert all(np.array_equal(x, y) for x, y in zip(sg, sc)), (seed, "state after the rollout")
    gpu.close(), cpu.close()


@pytest.mark.parametrize("n", [1000, 65536])
def test_pendulum_wide_angles(n, rollout_kernel):
    from oracle import oracle

    gpu = gymnasium_amd.make_vec("Pendulum-v1", num_envs=n, device=0, output="torch")
    cpu = gymnasium_amd.make_vec("Pendulum-v1", num_envs=n, _engine_factory=oracle.engine_factory)
    zeros = np.zeros(n, dtype=np.int32)
    host = lambda x: x.cpu().numpy()
    for seed in range(2):
        s = wide_pendulum_states(n, seed)
        for env in (gpu, cpu):
            env.reset(seed=seed)
            env.action_space.seed(3 + seed)
            env.set_state(s, zeros, np.zeros(n, dtype=np.uint8))
        import torch

        a = cpu.action_space.sample()  # step(): step_kernel
        g, c = gpu.step(torch.from_numpy(a).cuda()), cpu.step(a)
        for j, name in enumerate(("obs", "rewards", "terminations", "truncations")):
            assert np.array_equal(host(g[j]), c[j]), (seed, "step", name, s[np.flatnonzero((host(g[j]) != c[j]).reshape(n, -1).any(axis=1))[:4]])
        assert np.array_equal(gpu.get_state()[0].view(np.uint64), cpu.get_state()[0].view(np.uint64)), (seed, "state words after step()")
        for env in (gpu, cpu):
            env.set_state(s, zeros, np.zeros(n, dtype=np.uint8))
        gpu.action_space.np_random.bit_generator.state = cpu.action_space.np_random.bit_generator.state
        T = 16  # rollout(): the two-role kernel
        out = gpu.rollout(T)
        for k in range(T):
            a = cpu.action_space.sample()
            c = cpu.step(a)
            assert np.array_equal(host(out["actions"][k]).reshape(a.shape), a), (seed, k, "policy")
            for name, j in (("obs", 0), ("rewards", 1), ("terminations", 2), ("truncations", 3)):
                assert np.array_equal(host(out[name][k]), c[j]), (seed, k, name)
        sg, sc = gpu.get_state(), cpu.get_state()
        assert all(np.array_equal(x, y) for x, y in zip(sg, sc)), (seed, "state after the rollout")
    gpu.close(), cpu.close()


@pytest.mark.parametrize("env_id,states,T", [("Acrobot-v1", wide_acrobot_states, 8), ("MountainCar-v0", wide_mountaincar_states, 16),
                                             ("MountainCarContinuous-v0", wide_mountaincar_states, 16)])
def test_other_classic_envs_on_wide_states(env_id, states, T):
    import torch
    from oracle import oracle

    n = 20000  # (not a multiple of the workgroup's 256 sub-environments)
    gpu = gymnasium_amd.make_vec(env_id, num_envs=n, device=0, output="torch")
    cpu = gymnasium_amd.make_vec(env_id, num_envs=n, _engine_factory=oracle.engine_factory)
    zeros, flags = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.uint8)
    host = lambda x: x.cpu().numpy()
    for seed in range(2):
        s = states(n, 20 + seed)
        for env in (gpu, cpu):
            env.reset(seed=seed)
            env.action_space.seed(3 + seed)
            env.set_state(s, zeros, flags)
        a = cpu.action_space.sample()
        g, c = gpu.step(torch.from_numpy(a).cuda()), cpu.step(a)
        for j, name in enumerate(("obs", "rewards", "terminations", "truncations")):
            assert np.array_equal(host(g[j]), c[j]), (seed, "step", name, s[np.flatnonzero((host(g[j]) != c[j]).reshape(n, -1).any(axis=1))[:4]])
        sg, sc = gpu.get_state(), cpu.get_state()
        assert np.array_equal(sg[0].view(np.uint64), sc[0].view(np.uint64)), (seed, "state words after step()", s[np.flatnonzero((sg[0] != sc[0]).any(axis=1))[:4]])
        for env in (gpu, cpu):
            env.reset(seed=seed)
            env.set_state(s, zeros, flags)
        gpu.action_space.np_random.bit_generator.state = cpu.action_space.np_random.bit_generator.state
        out = gpu.rollout(T)
        for k in range(T):
            a = cpu.action_space.sample()
            c = cpu.step(a)
            assert np.array_equal(host(out["actions"][k]).reshape(a.shape), a), (seed, k, "policy")
            for name, j in (("obs", 0), ("rewards", 1), ("terminations", 2), ("truncations", 3)):
                assert np.array_equal(host(out[name][k]), c[j]), (seed, k, name)
        sg, sc = gpu.get_state(), cpu.get_state()
        assert all(np.array_equal(x, y) for x, y in zip(sg, sc)), (seed, "state after the rollout")
    gpu.close(), cpu.close()
