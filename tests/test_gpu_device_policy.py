"""GPU: the on-device policy of the per-step path -- `action_space.sample()` served by the engine's action stream
(mi_action_sample) and `step(None)` (mi_step with actions == NULL: the step kernel draws the batch itself) against the NumPy sampler of the same
seeded space (spaces/multi_discrete.py:176-178, spaces/box.py:463-465) and against the oracle stepped with host-sampled actions."""
import numpy as np
import pytest

import gymnasium_amd
import policy_suite as ps

pytestmark = pytest.mark.gpu

SEVEN = ["CartPole-v1", "Pendulum-v1", "Acrobot-v1", "MountainCar-v0", "MountainCarContinuous-v0", "Taxi-v4", "Ant-v5"]


@pytest.mark.parametrize("env_id", SEVEN + ["Blackjack-v1", "Humanoid-v5"])
@pytest.mark.parametrize("out", [dict(), dict(output="torch", sample_output="torch")], ids=["numpy", "torch"])
def test_sample_equals_the_numpy_sampler_for_1000_steps(env_id, out):
    ps.check_sample_equals_numpy(env_id, None, n=129 if env_id.endswith("-v5") else 1000, steps=1000, **out)


@pytest.mark.parametrize("env_id", ["CartPole-v1", "Pendulum-v1", "Taxi-v4", "Ant-v5"])
@pytest.mark.parametrize("out", [dict(), dict(output="torch"), dict(output="torch", sample_output="torch")], ids=["numpy", "torch-numpy", "torch-torch"])
def test_one_stream_for_every_consumer(env_id, out):
    ps.check_one_stream(env_id, None, n=300, **out)


@pytest.mark.parametrize("env_id", SEVEN + ["Blackjack-v1", "FrozenLake-v1", "HalfCheetah-v5", "Hopper-v5"])
def test_step_none_equals_step_of_a_sample(env_id):
    ps.check_step_none_equals_step_sample(env_id, None, n=129 if env_id.endswith("-v5") else 1000, steps=1000 if not env_id.endswith("-v5") else 60)


@pytest.mark.parametrize("env_id", SEVEN)
def test_step_none_against_the_oracle_with_host_sampled_actions(env_id, oracle_factory):
    """1 000 steps (MuJoCo: 40) of `step(None)` on the GPU == the oracle stepped with the NumPy sampler's batches: the drawn actions bit for bit, and
    the trajectory bit for bit (classic control, ToyText) or within the MuJoCo kinds' stated 1e-8."""
    mj = env_id.endswith("-v5")
    n, steps = (129, 40) if mj else (1000, 1000)
    gpu = gymnasium_amd.make_vec(env_id, num_envs=n, device=0, output="torch", sample_output="torch")
    cpu = gymnasium_amd.make_vec(env_id, num_envs=n, _engine_factory=oracle_factory)
    og, _ = gpu.reset(seed=11)
    oc, _ = cpu.reset(seed=11)
    assert np.array_equal(ps._np(og), oc) if not mj else np.allclose(ps._np(og), oc, rtol=0, atol=1e-12)
    gpu.action_space.seed(2)
    ref = ps.reference_space(gpu, 2)
    for t in range(steps):
        g = gpu.step(None)
        act = ref.sample()
        assert np.array_equal(ps._np(gpu.last_sampled_actions).reshape(act.shape), act), (env_id, t)
        c = cpu.step(act)
        if mj:
            np.testing.assert_allclose(ps._np(g[0]), c[0], rtol=0, atol=1e-8, err_msg=f"{env_id} obs t={t}")
            np.testing.assert_allclose(ps._np(g[1]), c[1], rtol=0, atol=1e-8, err_msg=f"{env_id} reward t={t}")
            cpu.set_state(*gpu.get_state())  # (windowed comparison, like tests/test_gpu_mujoco.py: chaotic dynamics amplify 1e-10)
        else:
            assert np.array_equal(ps._np(g[0]), c[0]) and np.array_equal(ps._np(g[1]), c[1]), (env_id, t)
        assert np.array_equal(ps._np(g[2]), c[2]) and np.array_equal(ps._np(g[3]), c[3]), (env_id, t)
    gpu.close(), cpu.close()


def test_step_none_at_the_benchmark_shape_equals_the_fused_rollout():
    """65 536 CartPoles: 128 x step(None) == rollout(128) of a twin (whose first launch is the reference's known-answer digest,
    tests/test_gpu_parity.py::test_fused_rollout_reproduces_the_reference_digest_at_full_size)."""
    import torch

    n, T = 65536, 128
    a = gymnasium_amd.make_vec("CartPole-v1", num_envs=n, device=0, output="torch", sample_output="torch")
    b = gymnasium_amd.make_vec("CartPole-v1", num_envs=n, device=0, output="torch")
    a.reset(seed=0), b.reset(seed=0)
    a.action_space.seed(0), b.action_space.seed(0)
    out = b.rollout(T)
    for t in range(T):
        o, r, te, tr, _ = a.step(None)
        assert torch.equal(a.last_sampled_actions, out["actions"][t]), t
        assert torch.equal(o, out["obs"][t]) and torch.equal(r, out["rewards"][t]) and torch.equal(te, out["terminations"][t]) and torch.equal(tr, out["truncations"][t]), t
    assert a.statistics() == b.statistics()
    assert np.array_equal(ps._np(a.action_space.sample()), b.action_space.sample())
    a.close(), b.close()


@pytest.mark.parametrize("env_id", ["CartPole-v1", "Pendulum-v1", "Taxi-v4", "Ant-v5"])
def test_random_policy_inside_a_captured_graph(env_id):
    """capture_steps(policy="random"): the action stream's position lives on the device, so every replay draws the NEXT batches -- the replayed
    steps equal eager step(None) calls of a twin, and the space's stream continues after the last replay."""
    import torch

    if env_id == "Taxi-v4":
        pytest.skip("ToyText assembles its infos on the host: its step() cannot be captured")
    n, G, R = 512, 5, 7
    a = gymnasium_amd.make_vec(env_id, num_envs=n, device=0, output="torch", sample_output="torch")
    b = gymnasium_amd.make_vec(env_id, num_envs=n, device=0, output="torch", sample_output="torch")
    a.reset(seed=4), b.reset(seed=4)
    a.action_space.seed(6), b.action_space.seed(6)
    for _ in range(2):  # kernels load on first use
        a.step(None), b.step(None)
    g = a.capture_steps(policy="random", steps=G)
    for rep in range(R):
        last = g.replay()
        for _ in range(G):
            eager = b.step(None)
        torch.cuda.synchronize()
        for k in range(4):
            assert torch.equal(last[k], eager[k]), (env_id, rep, k)
        assert torch.equal(a.last_sampled_actions, b.last_sampled_actions), (env_id, rep)
    assert np.array_equal(ps._np(a.action_space.sample()), ps._np(b.action_space.sample()))
    a.close(), b.close()
