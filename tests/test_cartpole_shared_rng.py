"""rng="shared": the semantics of the reference's own NumPy vector environment CartPoleVectorEnv (cartpole.py:353-505) -- one generator for all
sub-environments, float32 rewards, persistent reset bounds -- against trajectories recorded FROM that class (tests/golden/make_golden.py
make_cartpole_vector_entry_point).  Here: the oracle behind the product's host class, on the CPU; tests/test_gpu_parity.py runs the same
function on the HIP engine.  No tolerance: array_equal, and the generator state after every segment."""
import numpy as np
import pytest

import gymnasium_amd
from conftest import golden
from gymnasium_amd import _native

SEGMENTS = {"abc": dict(num_envs=8), "d": dict(num_envs=300, max_episode_steps=17), "e": dict(num_envs=5, sutton_barto_reward=True)}
RESETS = {"a": dict(seed=123), "b": dict(seed=7, options={"low": -0.1, "high": 0.08}), "c": dict(), "d": dict(seed=2**40 + 5), "e": dict(seed=0)}


def check_shared_rng_segments(make_env, via_rollout=False):
    g = golden("cartpole_vector_entry_point.npz")
    for tags, kw in SEGMENTS.items():
        env = make_env(**kw)
        assert env.metadata["autoreset_mode"] == gymnasium_amd.AutoresetMode.NEXT_STEP
        for tag in tags:
            obs0, info = env.reset(**RESETS[tag])
            obs0 = np.asarray(obs0.cpu()) if hasattr(obs0, "cpu") else obs0
            assert info == {} and obs0.dtype == np.float32 and np.array_equal(obs0, g[f"{tag}_reset_obs"]), tag
            acts = g[f"{tag}_actions"]
            if via_rollout:  # the fused entry point, teacher-forced (device tensors)
                import torch

                out = env.rollout(len(acts), actions=torch.from_numpy(acts))
                o, r, te, tr = (out[k].cpu().numpy() for k in ("obs", "rewards", "terminations", "truncations"))
                assert np.array_equal(o, g[f"{tag}_obs"]) and np.array_equal(r, g[f"{tag}_rewards"]), tag
                assert np.array_equal(te, g[f"{tag}_terminated"]) and np.array_equal(tr, g[f"{tag}_truncated"]), tag
            else:
                for t, a in enumerate(acts):
                    o, r, te, tr, info = env.step(a if not hasattr(obs0, "cpu") else a)
                    o, r, te, tr = (np.asarray(x.cpu()) if hasattr(x, "cpu") else x for x in (o, r, te, tr))
                    assert info == {} and r.dtype == np.float32, (tag, t)
                    assert np.array_equal(o, g[f"{tag}_obs"][t]), (tag, t)
                    assert np.array_equal(r, g[f"{tag}_rewards"][t]) and np.array_equal(te, g[f"{tag}_terminated"][t]) and np.array_equal(tr, g[f"{tag}_truncated"][t]), (tag, t)
                    if tag == "e":  # `-np.array(terminated, dtype=np.float32)`: a surviving pole is rewarded -0.0
                        assert np.array_equal(np.signbit(r), g["e_reward_signbit"][t]), (tag, t)
            # env.np_random IS the generator the sub-environments drew from: the host object follows the device's draws
            assert np.array_equal(_native.pcg_words(env.np_random), g[f"{tag}_rng_after"]), tag
        env.close()


def check_seed_sequence_entry_point(make_env):
    """The C ABI's own seeding entry point in this mode (mi_seed_sequence: SeedSequence(seed) -> PCG64 evaluated by the library; the host class hands over
    generator words instead): the one stream must be NumPy's default_rng(seed)."""
    for seed in (0, 42, 2**40 + 5, 2**63 + 11):
        env = make_env(num_envs=6)
        env._engine.seed_sequence(seed, 0, None)
        env._seeded = True
        obs, _ = env.reset()
        obs = np.asarray(obs.cpu()) if hasattr(obs, "cpu") else obs
        assert np.array_equal(obs, np.random.default_rng(seed).uniform(-0.05, 0.05, size=(4, 6)).T.astype(np.float32)), seed
        with pytest.raises(_native.NativeError):
            env._engine.seed_sequence(seed, 3, None)  # one generator: no shard offset
        env.close()


def test_oracle_seed_sequence_entry_point(oracle_factory):
    check_seed_sequence_entry_point(lambda **kw: gymnasium_amd.make_vec("CartPole-v1", rng="shared", _engine_factory=oracle_factory, **kw))


def test_oracle_equals_the_reference_vector_env(oracle_factory):
    check_shared_rng_segments(lambda **kw: gymnasium_amd.make_vec("CartPole-v1", rng="shared", _engine_factory=oracle_factory, **kw))


def test_stock_creator_defaults_to_the_shared_generator(oracle_factory):
    from gymnasium_amd.envs.classic_control import StockCartPoleVectorEnv

    check_shared_rng_segments(lambda **kw: StockCartPoleVectorEnv(_engine_factory=oracle_factory, **kw))


def test_shared_rng_refusals(oracle_factory):
    from gymnasium_amd.gym_api import error

    with pytest.raises(error.Error, match="NEXT_STEP"):
        gymnasium_amd.make_vec("CartPole-v1", num_envs=2, rng="shared", autoreset_mode="SameStep", _engine_factory=oracle_factory)
    with pytest.raises(error.Error, match="shard"):
        gymnasium_amd.make_vec("CartPole-v1", num_envs=2, rng="shared", env_index_offset=2, _engine_factory=oracle_factory)
    with pytest.raises(ValueError, match="rng must be"):
        gymnasium_amd.make_vec("CartPole-v1", num_envs=2, rng="global", _engine_factory=oracle_factory)
    env = gymnasium_amd.make_vec("CartPole-v1", num_envs=4, rng="shared", _engine_factory=oracle_factory)
    with pytest.raises(error.Error):
        env.reset(seed=[1, 2, 3, 4])  # one generator: a seed per sub-environment has no meaning (the reference raises there too)
    # a reset_mask is ignored like the reference's class ignores it: every sub-environment resets
    env.reset(seed=3)
    first = env.step(np.zeros(4, np.int64))[0].copy()
    env.reset(seed=3, options={"reset_mask": np.array([True, False, False, False])})
    assert np.array_equal(env.step(np.zeros(4, np.int64))[0], first)
    # assigning a generator re-bases the device's stream on it
    env.np_random = np.random.default_rng(99)
    expect = np.random.default_rng(99).uniform(-0.05, 0.05, size=(4, 4)).T.astype(np.float32)
    assert np.array_equal(env.reset()[0], expect)
    env.close()


def test_a_draw_from_np_random_moves_the_stream_of_the_sub_environments(oracle_factory):
    """in the reference `env.np_random` IS the generator the resets draw from (cartpole.py:475, 497), so a value the caller takes from it
    shifts every later reset / autoreset.  Also through a reference to the generator object that the caller kept across steps."""
    n = 5
    env = gymnasium_amd.make_vec("CartPole-v1", num_envs=n, rng="shared", _engine_factory=oracle_factory)
    env.reset(seed=11)
    ref = np.random.default_rng(11)
    ref.uniform(-0.05, 0.05, size=(4, n))          # the reset's draws
    g = env.np_random
    assert g.random() == ref.random()               # the caller's own draw continues the stream ...
    got, _ = env.reset()                            # ... and the next reset continues after it
    assert np.array_equal(got, ref.uniform(-0.05, 0.05, size=(4, n)).T.astype(np.float32))
    assert g.random() == ref.random()               # the kept object is up to date after the engine call
    # steps with autoresets in between: the kept object follows the device's draws
    env.action_space.seed(0)
    for t in range(80):
        env.step(env.action_space.sample())
    # (the real class is compared in the next test; here: the kept object equals the device's position, and a twin that does the same calls agrees)
    assert np.array_equal(_native.pcg_words(g), env._engine.get_rng()[0])
    v = g.random()
    env2 = gymnasium_amd.make_vec("CartPole-v1", num_envs=n, rng="shared", _engine_factory=oracle_factory)
    env2.reset(seed=11)
    g2 = env2.np_random
    g2.random(), env2.reset(), g2.random()
    env2.action_space.seed(0)
    for t in range(80):
        env2.step(env2.action_space.sample())
    assert env2.np_random.random() == v
    assert np.array_equal(env.reset()[0], env2.reset()[0])
    env.close(), env2.close()


def test_a_draw_from_np_random_against_the_reference_class(oracle_factory):
    from gymnasium_amd.gym_api import HAVE_GYMNASIUM

    if not HAVE_GYMNASIUM:
        pytest.skip("needs gymnasium itself")
    from gymnasium.envs.classic_control.cartpole import CartPoleVectorEnv as Ref

    n = 7
    ref, env = Ref(num_envs=n), gymnasium_amd.make_vec("CartPole-v1", num_envs=n, rng="shared", _engine_factory=oracle_factory)
    assert np.array_equal(ref.reset(seed=3)[0], env.reset(seed=3)[0])
    rng_a = np.random.default_rng(1)
    gr, ge = ref.np_random, env.np_random
    for t in range(200):
        if t % 13 == 5:  # the caller draws from the env's generator now and then (through the property and through the kept object)
            assert ref.np_random.random() == env.np_random.random()
        if t % 17 == 9:
            assert gr.random() == ge.random()
        a = (rng_a.random(n) * 2).astype(np.int64)
        r, e = ref.step(a), env.step(a)
        for k in range(4):
            assert np.array_equal(r[k], e[k]), (t, k)
    assert np.array_equal(ref.reset()[0], env.reset()[0])
    env.close()
