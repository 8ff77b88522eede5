"""-m gpu: rng="shared" (the reference's NumPy CartPoleVectorEnv semantics, MI_CFG_SHARED_RNG) on the HIP engine, through the C ABI: array_equal to the
trajectories recorded from the reference class, with NumPy batches, device tensors and the fused rollout entry point; and a large batch against the
oracle (many workgroups: the cross-workgroup scan that places every re-drawn sub-environment in the one stream)."""
import numpy as np
import pytest

import gymnasium_amd
from test_cartpole_shared_rng import check_seed_sequence_entry_point, check_shared_rng_segments

pytestmark = pytest.mark.gpu


def test_numpy_batches_equal_the_reference_vector_env():
    check_shared_rng_segments(lambda **kw: gymnasium_amd.make_vec("CartPole-v1", rng="shared", device=0, **kw))


def test_seed_sequence_entry_point():
    check_seed_sequence_entry_point(lambda **kw: gymnasium_amd.make_vec("CartPole-v1", rng="shared", device=0, **kw))


def test_device_tensors_equal_the_reference_vector_env():
    import torch

    class TorchActions:
        """the golden actions are NumPy: hand them over as device tensors"""

        def __init__(self, env):
            self.env = env

        def __getattr__(self, k):
            return getattr(self.env, k)

        def step(self, a):
            return self.env.step(torch.from_numpy(np.ascontiguousarray(a)).cuda())

    check_shared_rng_segments(lambda **kw: TorchActions(gymnasium_amd.make_vec("CartPole-v1", rng="shared", device=0, output="torch", **kw)))


def test_fused_rollout_equals_the_reference_vector_env():
    check_shared_rng_segments(lambda **kw: gymnasium_amd.make_vec("CartPole-v1", rng="shared", device=0, output="torch", **kw), via_rollout=True)


@pytest.mark.parametrize("fast_math", [False])
def test_many_workgroups_against_the_oracle(oracle_factory, fast_math):
    n, T = 20000, 96  # 79 workgroups; max_episode_steps = 12 makes bursts of simultaneous truncations on top of the random terminations
    gpu = gymnasium_amd.make_vec("CartPole-v1", num_envs=n, rng="shared", device=0, max_episode_steps=12)
    cpu = gymnasium_amd.make_vec("CartPole-v1", num_envs=n, rng="shared", max_episode_steps=12, _engine_factory=oracle_factory)
    assert np.array_equal(gpu.reset(seed=5)[0], cpu.reset(seed=5)[0])
    gpu.action_space.seed(1)
    for t in range(T):
        a = gpu.action_space.sample()
        g, c = gpu.step(a), cpu.step(a)
        for k in range(4):
            assert np.array_equal(g[k], c[k]), (t, k)
    assert np.array_equal(gpu.get_rng_state()[0], cpu.get_rng_state()[0])
    sg, sc = gpu.statistics(), cpu.statistics()
    assert sg == sc, (sg, sc)
    # on-device policy through the fused entry point == the same steps taken one by one
    gpu.close(), cpu.close()
    gpu = gymnasium_amd.make_vec("CartPole-v1", num_envs=4096, rng="shared", device=0, output="torch")
    cpu = gymnasium_amd.make_vec("CartPole-v1", num_envs=4096, rng="shared", _engine_factory=oracle_factory)
    gpu.reset(seed=9), cpu.reset(seed=9)
    gpu.action_space.seed(3), cpu.action_space.seed(3)
    out = gpu.rollout(40)
    for t in range(40):
        a = cpu.action_space.sample()
        o, r, te, tr, _ = cpu.step(a)
        assert np.array_equal(out["actions"][t].cpu().numpy(), a) and np.array_equal(out["obs"][t].cpu().numpy(), o), t
        assert np.array_equal(out["terminations"][t].cpu().numpy(), te) and np.array_equal(out["rewards"][t].cpu().numpy(), r.astype(np.float64)), t
    gpu.close(), cpu.close()


def test_set_state_recounts_the_pending_resets(oracle_factory):
    """mi_set_state may change which sub-environments are pending an autoreset: the per-workgroup counts the next step's scan reads must follow the new flags."""
    n = 1000
    gpu = gymnasium_amd.make_vec("CartPole-v1", num_envs=n, rng="shared", device=0)
    cpu = gymnasium_amd.make_vec("CartPole-v1", num_envs=n, rng="shared", _engine_factory=oracle_factory)
    gpu.reset(seed=4), cpu.reset(seed=4)
    gpu.action_space.seed(0)
    for _ in range(3):
        a = gpu.action_space.sample()
        gpu.step(a), cpu.step(a)
    st, el, fl = cpu.get_state()
    fl = fl.copy()
    fl[::7] |= 1  # MI_FLAG_NEEDS_RESET on every seventh sub-environment
    fl[5::11] &= 0xFE
    gpu.set_state(st, el, fl), cpu.set_state(st, el, fl)
    for t in range(6):
        a = gpu.action_space.sample()
        g, c = gpu.step(a), cpu.step(a)
        for k in range(4):
            assert np.array_equal(g[k], c[k]), (t, k)
    assert np.array_equal(gpu.get_rng_state()[0], cpu.get_rng_state()[0])
    gpu.close(), cpu.close()


def test_a_batch_with_an_invalid_action_is_refused_whole():
    """cartpole.py:424-426: `assert self.action_space.contains(action)` before anything is touched.  A device caller's bad batch steps no
    sub-environment and consumes no draw: after the error has been raised the env continues exactly like a twin that never saw the batch."""
    import torch

    n = 3000
    a = gymnasium_amd.make_vec("CartPole-v1", num_envs=n, rng="shared", device=0, output="torch", max_episode_steps=9)
    b = gymnasium_amd.make_vec("CartPole-v1", num_envs=n, rng="shared", device=0, output="torch", max_episode_steps=9)
    a.reset(seed=2), b.reset(seed=2)
    gen = torch.Generator(device="cpu").manual_seed(0)
    acts = [torch.randint(0, 2, (n,), generator=gen).cuda() for _ in range(40)]
    for t in range(20):
        a.step(acts[t]), b.step(acts[t])
    bad = acts[20].clone()
    bad[1234] = 7
    a.step(bad)  # enqueued; the error word is raised by the next call that looks
    with pytest.raises(AssertionError, match="action"):
        a.step(acts[20])
        a.synchronize()
    for t in range(20, 40):
        ra, rb = a.step(acts[t]), b.step(acts[t])
        for k in range(4):
            assert torch.equal(ra[k], rb[k]), (t, k)
    assert np.array_equal(a.get_rng_state()[0], b.get_rng_state()[0]) and a.statistics() == b.statistics()
    a.close(), b.close()
