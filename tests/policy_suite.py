"""Backend-agnostic checks of the on-device policy (gymnasium_amd/vector/device_policy.py, mi_action_sample / mi_step with actions == NULL).

The known answer is the reference's own sampler: ``batch_space(single_action_space, n)`` seeded like the env's space, drawn with NumPy
(gymnasium/spaces/multi_discrete.py:176-178, box.py:463-465; under the real gymnasium these ARE gymnasium's classes, on the GPU box the mirror,
which tests/test_oracle_golden.py pins on action_samples.npz).  ``factory``: the oracle's engine factory (CPU) or None (the HIP engine).
"""
import copy
import pickle

import numpy as np

import gymnasium_amd
from gymnasium_amd.gym_api import batch_space

IDS = ["CartPole-v1", "Pendulum-v1", "Acrobot-v1", "MountainCar-v0", "MountainCarContinuous-v0", "Taxi-v4", "FrozenLake-v1", "Blackjack-v1",
       "Ant-v5", "HalfCheetah-v5", "Humanoid-v5"]


def _make(env_id, n, factory, **kw):
    extra = {} if factory is None else {"_engine_factory": factory}
    return gymnasium_amd.make_vec(env_id, num_envs=n, **extra, **kw)


def _np(x):
    return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)


def _same(x, y):
    if isinstance(x, (tuple, list)):
        return len(x) == len(y) and all(_same(p, q) for p, q in zip(x, y))
    return np.array_equal(_np(x), _np(y))


def reference_space(env, seed):
    ref = batch_space(env.single_action_space, env.num_envs)
    ref.seed(seed)
    return ref


def check_sample_equals_numpy(env_id, factory, n=300, steps=1000, **kw):
    """``steps`` successive action_space.sample() == the NumPy sampler of the same seeded space, across several refills of the draw-ahead block."""
    env = _make(env_id, n, factory, **kw)
    env.action_space._hip_ring_steps = min(env.action_space._hip_ring_steps, 96)  # (several refills within `steps`)
    env.action_space.seed(123)
    ref = reference_space(env, 123)
    for t in range(steps):
        a, b = env.action_space.sample(), ref.sample()
        assert _np(a).dtype == b.dtype and _np(a).shape == b.shape, (env_id, t, _np(a).dtype, b.dtype, _np(a).shape, b.shape)
        assert np.array_equal(_np(a), b), f"{env_id}: sample {t} differs from the NumPy sampler"
        if kw.get("sample_output") == "torch":
            assert hasattr(a, "data_ptr")
    # the NumPy generator of the space is where the stream is: the next host draw continues it
    assert env.action_space.np_random.random() == ref.np_random.random()
    env.close()


def check_one_stream(env_id, factory, n=64, **kw):
    """sample(), np_random draws, rollout(), step(None), seed(), pickling and close() consume ONE stream in call order, like the reference's space."""
    torch_out = kw.get("output") == "torch"
    env = _make(env_id, n, factory, **kw)
    env.reset(seed=1)
    env.action_space.seed(7)
    ref = reference_space(env, 7)
    for _ in range(3):
        assert np.array_equal(_np(env.action_space.sample()), ref.sample())
    # a host draw in between (returns what was drawn ahead, moves the generator to the stream's position)
    assert np.array_equal(env.action_space.np_random.random(5), ref.np_random.random(5))
    assert np.array_equal(_np(env.action_space.sample()), ref.sample())
    if torch_out:
        out = env.rollout(6)  # the on-device policy of the fused rollout draws from the same stream
        for t in range(6):
            assert np.array_equal(_np(out["actions"][t]).reshape(ref.shape), ref.sample()), f"rollout step {t}"
        assert np.array_equal(_np(env.action_space.sample()), ref.sample())
        for _ in range(4):  # step(None): the step kernel draws its own batch
            env.step(None)
            assert np.array_equal(_np(env.last_sampled_actions), ref.sample())
        assert np.array_equal(_np(env.action_space.sample()), ref.sample())
    # copies detach at the current position and continue on the host
    dup = copy.deepcopy(env.action_space)
    pk = pickle.loads(pickle.dumps(env.action_space))
    want = ref.sample()
    assert np.array_equal(dup.sample(), want) and np.array_equal(pk.sample(), want)
    assert np.array_equal(_np(env.action_space.sample()), want)
    # re-seeding starts a new stream
    env.action_space.seed(99)
    ref.seed(99)
    assert np.array_equal(_np(env.action_space.sample()), ref.sample())
    space = env.action_space
    env.close()  # the space outlives the env with its generator at the stream's position
    assert np.array_equal(space.sample(), ref.sample())


def check_step_none_equals_step_sample(env_id, factory, n=256, steps=120, tol=0.0, **kw):
    """step(None) -- the policy drawn inside the step kernel -- == step(action_space.sample()) on a twin, observation for observation."""
    import torch

    a = _make(env_id, n, factory, output="torch", sample_output="torch", **kw)
    b = _make(env_id, n, factory, output="torch", sample_output="torch", **kw)
    oa, _ = a.reset(seed=3)
    ob, _ = b.reset(seed=3)
    assert _same(oa, ob)
    a.action_space.seed(5), b.action_space.seed(5)
    for t in range(steps):
        ra = a.step(None)
        act = b.action_space.sample()
        rb = b.step(act)
        assert torch.equal(a.last_sampled_actions.reshape(act.shape), act), (env_id, t)
        for k in range(4):
            if tol and k < 2:
                np.testing.assert_allclose(_np(ra[k]), _np(rb[k]), rtol=tol, atol=tol, err_msg=f"{env_id} t={t} k={k}")
            else:
                assert _same(ra[k], rb[k]), (env_id, t, k)
    assert a.statistics() == b.statistics()
    assert np.array_equal(_np(a.action_space.sample()), _np(b.action_space.sample()))
    a.close(), b.close()
