"""CPU: the device-sampled action space (vector/device_policy.py) through the host class with the oracle behind it -- the draws equal the NumPy
sampler's, and every consumer of the stream (sample, np_random, rollout, step(None), copies, close) keeps ONE position."""
import numpy as np
import pytest

import gymnasium_amd
import policy_suite as ps


@pytest.mark.parametrize("env_id", ps.IDS)
def test_sample_equals_the_numpy_sampler(env_id, oracle_factory):
    ps.check_sample_equals_numpy(env_id, oracle_factory, n=37 if env_id.endswith("-v5") else 300, steps=260)


@pytest.mark.parametrize("env_id", ["CartPole-v1", "Pendulum-v1", "Taxi-v4", "HalfCheetah-v5"])
@pytest.mark.parametrize("out", [dict(), dict(output="torch"), dict(output="torch", sample_output="torch")])
def test_one_stream_for_every_consumer(env_id, out, oracle_factory):
    ps.check_one_stream(env_id, oracle_factory, n=16, **out)


@pytest.mark.parametrize("env_id", ["CartPole-v1", "MountainCarContinuous-v0", "Blackjack-v1", "Hopper-v5"])
def test_step_none_equals_step_of_a_sample(env_id, oracle_factory):
    ps.check_step_none_equals_step_sample(env_id, oracle_factory, n=24, steps=60)


@pytest.mark.parametrize("env_id", ["CartPole-v1", "Pendulum-v1", "Blackjack-v1", "Taxi-v4"])
def test_step_none_with_numpy_batches_is_the_two_calls_it_stands_for(env_id, oracle_factory):
    """output="numpy": step(None) == step(action_space.sample()), through whatever a subclass does to step()'s result (Blackjack's tuple of columns:
    the fallback once went through self.step() a second time; found by scripts/r06/gpu_soak.py)."""
    a = gymnasium_amd.make_vec(env_id, num_envs=9, _engine_factory=oracle_factory)
    b = gymnasium_amd.make_vec(env_id, num_envs=9, _engine_factory=oracle_factory)
    a.reset(seed=2), b.reset(seed=2)
    a.action_space.seed(6), b.action_space.seed(6)
    for _ in range(40):
        ra, rb = a.step(None), b.step(b.action_space.sample())
        for k in range(4):
            assert ps._same(ra[k], rb[k])
    a.close(), b.close()


def test_sample_output_torch_needs_torch_output(oracle_factory):
    with pytest.raises(ValueError, match="sample_output"):
        gymnasium_amd.make_vec("CartPole-v1", num_envs=2, sample_output="torch", _engine_factory=oracle_factory)


def test_masked_sampling_takes_the_reference_path(oracle_factory):
    """Masks are outside the engine's sampler: the call falls through to the space's own NumPy implementation at the stream's position."""
    from gymnasium_amd.gym_api import HAVE_GYMNASIUM

    env = gymnasium_amd.make_vec("CartPole-v1", num_envs=5, _engine_factory=oracle_factory)
    env.action_space.seed(4)
    ref = ps.reference_space(env, 4)
    assert np.array_equal(env.action_space.sample(), ref.sample())
    if HAVE_GYMNASIUM:
        mask = tuple(np.array([1, 0], dtype=np.int8) for _ in range(5))
        assert np.array_equal(env.action_space.sample(mask=mask), ref.sample(mask=mask))
    assert np.array_equal(env.action_space.sample(), ref.sample())
    env.close()


def test_the_space_is_still_the_reference_class(oracle_factory):
    from gymnasium_amd.gym_api import spaces

    env = gymnasium_amd.make_vec("CartPole-v1", num_envs=3, _engine_factory=oracle_factory)
    assert isinstance(env.action_space, spaces.MultiDiscrete) and env.action_space == ps.reference_space(env, 0)
    assert repr(env.action_space) == repr(ps.reference_space(env, 0))
    env.close()
    env = gymnasium_amd.make_vec("Pendulum-v1", num_envs=3, _engine_factory=oracle_factory)
    assert isinstance(env.action_space, spaces.Box) and env.action_space == ps.reference_space(env, 0)
    env.close()
