"""The engine's host class under the REAL gymnasium (the configuration a Gymnasium user has).

With Farama gymnasium importable, `gymnasium_amd` registers `MI355X/<id>` in gymnasium's own registry and `HipVectorEnv`
subclasses gymnasium's `VectorEnv`.  These tests create the env through `gymnasium.make_vec` (the reference's plug-in
boundary, envs/registration.py:829-988), drive it with the checker backend (`_engine_factory=oracle`: no GPU here) and
compare every step's (obs, reward, terminated, truncated, infos) with `gymnasium.make_vec(id, n, "sync")` -- the
reference's SyncVectorEnv -- using the reference's own strict `data_equivalence(..., exact=True)`
(utils/env_checker.py:34-74): types, dtypes, shapes, dict keys, masks and values.

gymnasium is not installed in the build container or on the GPU box; it is importable from the read-only reference tree, which
tests/conftest.py puts on the path when it exists -- so the default CPU suite RUNS these tests here; without the tree they skip.
"""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get("GYMNASIUM_REFERENCE_TREE", "/root/reference")

try:
    if os.environ.get("GYMNASIUM_AMD_FORCE_MIRROR", "0") == "1":
        raise ImportError("mirror forced")
    import gymnasium as gym
    from gymnasium.utils.env_checker import data_equivalence

    HAVE = True
except ImportError:
    gym, HAVE = None, False

needs_gymnasium = pytest.mark.skipif(not HAVE, reason="Farama gymnasium is not importable in this interpreter")

CLASSIC = ["CartPole-v1", "Pendulum-v1", "Acrobot-v1", "MountainCar-v0", "MountainCarContinuous-v0"]
TOYTEXT = ["FrozenLake-v1", "FrozenLake8x8-v1", "CliffWalking-v1", "Taxi-v4", "Blackjack-v1"]
MODES = ["NextStep", "SameStep", "Disabled"]


def _ulp1_f32(a, b):
    ai, bi = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    return bool((np.abs(ai - bi) <= 1).all())


def _same(a, b, env_id, after_reset_rows=None):
    if data_equivalence(a, b, exact=True):
        return True
    # The one stated exception: Acrobot's observation right after a reset -- float32 cos / sin that NumPy
    # evaluates with CPU-feature-dependent SIMD kernels -- may differ by 1 float32 ulp.
    if env_id == "Acrobot-v1" and isinstance(a, np.ndarray) and isinstance(b, np.ndarray) and a.dtype == b.dtype == np.float32 and a.shape == b.shape:
        rows = np.ones(len(a), bool) if after_reset_rows is None else after_reset_rows
        return bool(np.array_equal(a[~rows], b[~rows]) and _ulp1_f32(a[rows], b[rows]))
    return False


@needs_gymnasium
@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("env_id", CLASSIC + TOYTEXT)
def test_make_vec_equals_sync_vector_env(env_id, mode, oracle_factory):
    import gymnasium_amd  # noqa: F401  (registers MI355X/<id> in gymnasium's registry)

    n, T = 6, 300
    kw = {"max_episode_steps": 40} if env_id in ("CliffWalking-v1", "MountainCarContinuous-v0", "Acrobot-v1") else {}
    ours = gym.make_vec(f"MI355X/{env_id}", num_envs=n, autoreset_mode=mode, _engine_factory=oracle_factory, **kw)
    ref = gym.make_vec(env_id, num_envs=n, vectorization_mode="sync", vector_kwargs={"autoreset_mode": mode}, **kw)
    assert isinstance(ours, gym.vector.VectorEnv) and type(ours).__module__.startswith("gymnasium_amd")
    assert ours.metadata["autoreset_mode"] == ref.metadata["autoreset_mode"]
    assert ours.single_observation_space == ref.single_observation_space and ours.single_action_space == ref.single_action_space
    assert ours.observation_space == ref.observation_space and ours.action_space == ref.action_space
    assert ours.spec.id == f"MI355X/{env_id}" and ours.max_episode_steps == ref.envs[0].spec.max_episode_steps

    o1, i1 = ours.reset(seed=123)
    o2, i2 = ref.reset(seed=123)
    assert _same(o1, o2, env_id) and data_equivalence(i1, i2, exact=True), (i1, i2)
    ours.action_space.seed(7), ref.action_space.seed(7)
    dones_seen = 0
    pending = np.zeros(n, bool)
    for t in range(T):
        a = ref.action_space.sample()
        assert data_equivalence(a, ours.action_space.sample(), exact=True)
        s1, s2 = ours.step(a), ref.step(a)
        after_reset = pending if mode == "NextStep" else (s2[2] | s2[3] if mode == "SameStep" else np.zeros(n, bool))
        assert _same(s1[0], s2[0], env_id, after_reset), f"obs t={t}"
        for k, what in ((1, "reward"), (2, "terminated"), (3, "truncated")):
            assert data_equivalence(s1[k], s2[k], exact=True), f"{what} t={t}: {s1[k]!r} vs {s2[k]!r}"
        inf1, inf2 = dict(s1[4]), dict(s2[4])
        if env_id == "Acrobot-v1" and "final_obs" in inf2:  # final observations are post-step (not post-reset): exact
            pass
        assert data_equivalence(inf1, inf2, exact=True), f"infos t={t}: {inf1!r} vs {inf2!r}"
        done = s2[2] | s2[3]
        dones_seen += int(done.sum())
        pending = done if mode == "NextStep" else np.zeros(n, bool)
        if mode == "Disabled" and done.any():
            r1, ri1 = ours.reset(options={"reset_mask": done})
            r2, ri2 = ref.reset(options={"reset_mask": done})
            assert _same(r1, r2, env_id, done) and data_equivalence(ri1, ri2, exact=True), f"masked reset t={t}"
    assert dones_seen > 0
    ours.close(), ref.close()


@needs_gymnasium
@pytest.mark.parametrize("mode", ["NextStep", "SameStep"])
@pytest.mark.parametrize("env_id", ["Pendulum-v1", "MountainCarContinuous-v0"])
def test_float64_action_rows_equal_sync_vector_env(env_id, mode, oracle_factory):
    """A float64 action batch is handed to the scalar envs un-rounded (vector/sync_vector_env.py:274 iterate()), which changes NumPy's
    promotions inside the step (pendulum.py:127-139 all float64; continuous_mountain_car.py:153-178 np.float32 state + np.float64 force):
    the engine's MI_F64 action rows must reproduce that bit for bit -- mixed with float32 batches, Python lists, integer arrays and values
    beyond the bounds (the clip / min-max branches return Python floats there)."""
    import gymnasium_amd  # noqa: F401

    n, T = 8, 400
    kw = {"max_episode_steps": 60}
    ours = gym.make_vec(f"MI355X/{env_id}", num_envs=n, autoreset_mode=mode, _engine_factory=oracle_factory, **kw)
    ref = gym.make_vec(env_id, num_envs=n, vectorization_mode="sync", vector_kwargs={"autoreset_mode": mode}, **kw)
    assert data_equivalence(ours.reset(seed=9), ref.reset(seed=9), exact=True)
    rng = np.random.default_rng(4)
    hi = float(ref.single_action_space.high[0])
    for t in range(T):
        kind = t % 6
        if kind == 5:
            a = [[np.float64(x)] for x in rng.uniform(-hi, hi, n)]  # lists of NumPy scalars: STRONG float64 (NEP 50), unlike lists of Python floats
        elif kind == 0:
            a = rng.uniform(-hi, hi, (n, 1))  # float64
        elif kind == 1:
            a = rng.uniform(-1.5 * hi, 1.5 * hi, (n, 1))  # float64, some beyond the bounds
        elif kind == 2:
            a = rng.uniform(-1.2 * hi, 1.2 * hi, (n, 1)).astype(np.float32)
        elif kind == 3:
            a = rng.uniform(-hi, hi, (n, 1)).tolist()  # a list of lists: float64 once it is an array
        else:
            a = rng.integers(-2, 3, (n, 1))  # int64: exact in float64
        s1, s2 = ours.step(a), ref.step(a)
        for k, what in enumerate(("obs", "reward", "terminated", "truncated")):
            assert data_equivalence(s1[k], s2[k], exact=True), f"{what} t={t} kind={kind}: {s1[k]!r} vs {s2[k]!r}"
        assert data_equivalence(dict(s1[4]), dict(s2[4]), exact=True), f"infos t={t}"
    ours.close(), ref.close()


@needs_gymnasium
@pytest.mark.parametrize("state_dtype", [np.float32, np.float64])
def test_mountaincar_continuous_clamps_with_every_action_kind(state_dtype, oracle_factory):
    """continuous_mountain_car.py:150-178 at the places where its scalars change KIND (np.float32 / np.float64 / Python float): the speed
    clamps, the position clamps, the inelastic left wall, the goal test -- from teacher-forced states (float32 array = after any step,
    float64 array = right after a reset), with float32 rows, float64 rows, Python lists and out-of-range forces."""
    import gymnasium_amd  # noqa: F401

    n = 64
    # SAME_STEP: a sub-environment that reaches the goal resets within the step (its terminal observation is infos["final_obs"]), so no
    # autoreset is ever pending when the next trial's states are forced
    ours = gym.make_vec("MI355X/MountainCarContinuous-v0", num_envs=n, autoreset_mode="SameStep", _engine_factory=oracle_factory)
    ref = gym.make_vec("MountainCarContinuous-v0", num_envs=n, vectorization_mode="sync", vector_kwargs={"autoreset_mode": "SameStep"})
    ours.reset(seed=1), ref.reset(seed=1)
    goals = 0
    rng = np.random.default_rng(11)
    centres = np.array([[-1.2, -0.07], [-1.2, 0.0], [-1.199, -0.06], [0.6, 0.07], [0.599, 0.069], [0.45, 0.0], [0.449, 0.01], [-0.5, 0.07], [-0.5, -0.07], [0.3, 0.0695]])
    for trial in range(60):
        st = centres[rng.integers(0, len(centres), n)] + rng.normal(0, [2e-3, 1e-3], (n, 2)) * (rng.random((n, 1)) < 0.7)
        st = np.clip(st, [-1.2, -0.07], [0.6, 0.07]).astype(state_dtype)
        flags = np.full(n, 2 if state_dtype is np.float32 else 0, np.uint8)  # MI_FLAG_STATE_F32
        ours.set_state(st.astype(np.float64), np.zeros(n, np.int32), flags)
        for i, e in enumerate(ref.envs):
            e.unwrapped.state = st[i].copy()
        a = rng.uniform(-1.3, 1.3, (n, 1))
        a[rng.random(n) < 0.15] = rng.choice([-1.0, 1.0, 0.0])
        a = [a, a.astype(np.float32), a.tolist()][trial % 3]
        s1, s2 = ours.step(a), ref.step(a)
        for k, what in enumerate(("obs", "reward", "terminated", "truncated")):
            assert data_equivalence(s1[k], s2[k], exact=True), f"{what} trial={trial}: {s1[k]!r} vs {s2[k]!r}"
        assert data_equivalence(dict(s1[4]), dict(s2[4]), exact=True), f"infos trial={trial}"
        live = ~(s2[2] | s2[3])
        goals += int((~live).sum())
        got = ours.get_state()[0][live]
        want = np.stack([e.unwrapped.state for e, alive in zip(ref.envs, live) if alive])
        assert want.dtype == np.float32 and np.array_equal(got, want.astype(np.float64))
    assert goals > 10
    ours.close(), ref.close()


@needs_gymnasium
def test_module_prefixed_id_auto_imports_the_package(oracle_factory):
    """envs/registration.py:494-502: "module:id" imports the module, which registers the id."""
    env = gym.make_vec("gymnasium_amd:MI355X/CartPole-v1", num_envs=3, _engine_factory=oracle_factory)
    assert type(env).__module__.startswith("gymnasium_amd") and env.num_envs == 3
    assert env.spec.id == "MI355X/CartPole-v1" and env.spec.kwargs["vectorization_mode"] == "vector_entry_point"
    env.close()


@needs_gymnasium
def test_gymnasium_amd_make_vec_never_returns_the_reference_env(oracle_factory):
    """A stock id given to gymnasium_amd.make_vec resolves into MI355X/ -- it must not come back as Farama's CPU CartPoleVectorEnv."""
    import gymnasium_amd

    env = gymnasium_amd.make_vec("CartPole-v1", num_envs=4, _engine_factory=oracle_factory)
    assert isinstance(env, gymnasium_amd.HipVectorEnv) and env.spec.id == "MI355X/CartPole-v1"
    env.close()
    env = gymnasium_amd.make_vec(gym.spec("MI355X/Pendulum-v1"), num_envs=2, _engine_factory=oracle_factory)
    assert isinstance(env, gymnasium_amd.HipVectorEnv)
    env.close()
    with pytest.raises(gym.error.Error, match="MI355X engines only"):
        gymnasium_amd.make_vec("phys2d/CartPole-v1", num_envs=2)
    with pytest.raises(gym.error.Error, match="MI355X engines only"):
        gymnasium_amd.make_vec(gym.spec("CartPole-v1"), num_envs=2)
    with pytest.raises(gym.error.Error, match="gymnasium's own path"):
        gymnasium_amd.make_vec("CartPole-v1", num_envs=2, vectorization_mode="sync")
    # no GPU here and no checker passed: the product path must fail loudly, not fall back to a CPU implementation
    from gymnasium_amd import _native

    if _native.load_library().device_count() == 0:
        with pytest.raises((_native.NativeError, ImportError)):
            gymnasium_amd.make_vec("CartPole-v1", num_envs=4)
        with pytest.raises((_native.NativeError, ImportError)):
            gym.make_vec("MI355X/CartPole-v1", num_envs=4)


@needs_gymnasium
@pytest.mark.parametrize("kwargs", [
    {}, {"num_envs": 3}, {"vectorization_mode": "vector_entry_point"}, {"sutton_barto_reward": True},
    {"vectorization_mode": "vector_entry_point", "sutton_barto_reward": True}, {"max_episode_steps": 5},
])
def test_make_vec_kwargs_cases(kwargs, oracle_factory):
    """The vector_entry_point cases of tests/envs/registration/test_make_vec.py:120-133,188-191 on the MI355X id: kwargs reach the
    creator, `spec` is recreatable, num_envs / max_episode_steps are honoured."""
    import gymnasium_amd  # noqa: F401
    from gymnasium.envs.registration import VectorizeMode

    env = gym.make_vec("MI355X/CartPole-v1", _engine_factory=oracle_factory, **kwargs)
    assert env.num_envs == kwargs.get("num_envs", 1)
    assert env.spec.kwargs["vectorization_mode"] == VectorizeMode.VECTOR_ENTRY_POINT.value
    assert env.max_episode_steps == kwargs.get("max_episode_steps", 500)
    env.reset(seed=0)
    _, r, te, _, _ = env.step(env.action_space.sample())
    if kwargs.get("sutton_barto_reward"):
        assert (r == np.where(te, -1.0, 0.0)).all()  # cartpole.py:210-222
    else:
        assert (r == 1.0).all()
    if kwargs.get("max_episode_steps") == 5:
        for _ in range(4):
            _, _, te, tr, _ = env.step(env.action_space.sample())
        assert (te | tr).all()
    # the spec round-trips through gymnasium's make_vec like the reference's own (test_make_vec.py:188-191)
    spec_kwargs = {k: v for k, v in env.spec.kwargs.items() if k != "_engine_factory"}
    assert spec_kwargs.get("num_envs", 1) == env.num_envs
    env.close()
    with pytest.raises(gym.error.Error, match="vector_kwargs"):
        gym.make_vec("MI355X/CartPole-v1", vector_kwargs={"copy": False}, _engine_factory=oracle_factory)
    with pytest.raises(gym.error.Error, match="wrappers"):
        gym.make_vec("MI355X/CartPole-v1", wrappers=(gym.wrappers.TimeAwareObservation,), _engine_factory=oracle_factory)
    with pytest.raises(ValueError, match="Invalid vectorization mode"):
        gym.make_vec("MI355X/CartPole-v1", vectorization_mode="invalid")


@needs_gymnasium
def test_gymnasium_vector_wrappers_compose(oracle_factory):
    """gymnasium's own vector wrappers wrap the engine's env like any VectorEnv (it IS one)."""
    import gymnasium_amd  # noqa: F401
    from gymnasium.wrappers.vector import ClipReward, RecordEpisodeStatistics

    ours = RecordEpisodeStatistics(ClipReward(gym.make_vec("MI355X/CartPole-v1", num_envs=4, _engine_factory=oracle_factory), 0.0, 0.5))
    ref = RecordEpisodeStatistics(ClipReward(gym.make_vec("CartPole-v1", num_envs=4, vectorization_mode="sync"), 0.0, 0.5))
    ours.reset(seed=1), ref.reset(seed=1)
    ours.action_space.seed(2), ref.action_space.seed(2)
    seen = False
    for _ in range(120):
        a = ref.action_space.sample()
        s1, s2 = ours.step(a), ref.step(a)
        assert data_equivalence(s1[:4], s2[:4], exact=True)
        if "episode" in s2[4]:
            seen = True
            assert data_equivalence(s1[4]["episode"]["r"], s2[4]["episode"]["r"], exact=True)
            assert data_equivalence(s1[4]["episode"]["l"], s2[4]["episode"]["l"], exact=True)
    assert seen
    ours.close(), ref.close()


@needs_gymnasium
def test_stock_id_override_is_a_drop_in_for_cartpole_vector_env(oracle_factory):
    """`register_envs(override_stock_ids=True)` puts the engine behind the STOCK id.  gymnasium's CartPole-v1 already has a vector_entry_point (the NumPy
    CartPoleVectorEnv, cartpole.py:353-505), so plain `gymnasium.make_vec("CartPole-v1", n)` must keep returning exactly that class's numbers: one shared
    generator, float32 rewards.  Strict data_equivalence against the reference class itself, then the registry is put back."""
    import gymnasium_amd
    from gymnasium.envs.classic_control.cartpole import CartPoleVectorEnv as RefVec

    n, T = 64, 300
    ref = RefVec(num_envs=n, max_episode_steps=40)
    stock = {k: spec.vector_entry_point for k, spec in gym.registry.items()}
    try:
        gymnasium_amd.register_envs(override_stock_ids=True)
        ours = gym.make_vec("CartPole-v1", num_envs=n, max_episode_steps=40, _engine_factory=oracle_factory)
        assert type(ours).__name__ == "StockCartPoleVectorEnv" and isinstance(ours, gym.vector.VectorEnv)
    finally:
        for k, vep in stock.items():
            gym.registry[k].vector_entry_point = vep
    for seed, options in ((11, None), (None, {"low": -0.2, "high": 0.2}), (5, {"high": 0.01})):
        assert data_equivalence(ours.reset(seed=seed, options=options), ref.reset(seed=seed, options=options), exact=True) or seed is None
        if seed is None:  # both continue their own stream from the same state: equal again
            assert data_equivalence(ours.reset(options=options), ref.reset(options=options), exact=True)
        ours.action_space.seed(seed or 0)
        for t in range(T):
            a = ours.action_space.sample()
            s1, s2 = ours.step(a), ref.step(a)
            for k, what in enumerate(("obs", "reward", "terminated", "truncated", "infos")):
                assert data_equivalence(s1[k], s2[k], exact=True), f"{what} t={t}: {s1[k]!r} vs {s2[k]!r}"
        assert ours.np_random.bit_generator.state == ref.np_random.bit_generator.state
    ours.close(), ref.close()


@needs_gymnasium
def test_shared_generator_mode_property_based(oracle_factory):
    """Random batch sizes, TimeLimits (down to 1: every sub-environment finishes on every step it takes), seeds, reset bounds and action sequences:
    rng="shared" against the reference's NumPy CartPoleVectorEnv under strict data_equivalence, generator state included (hypothesis)."""
    hypothesis = pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st

    import gymnasium_amd
    from gymnasium.envs.classic_control.cartpole import CartPoleVectorEnv as RefVec

    @settings(max_examples=120, deadline=None, derandomize=True)
    @given(n=st.integers(1, 130), max_steps=st.integers(1, 25), seed=st.integers(0, 2**63 - 1), steps=st.integers(1, 60), sb=st.booleans(),
           bounds=st.one_of(st.none(), st.tuples(st.floats(-0.2, 0.0), st.floats(0.0, 0.2))), aseed=st.integers(0, 2**32 - 1))
    def run(n, max_steps, seed, steps, sb, bounds, aseed):
        ours = gymnasium_amd.make_vec("CartPole-v1", num_envs=n, rng="shared", max_episode_steps=max_steps, sutton_barto_reward=sb, _engine_factory=oracle_factory)
        ref = RefVec(num_envs=n, max_episode_steps=max_steps, sutton_barto_reward=sb)
        options = None if bounds is None else {"low": bounds[0], "high": bounds[1]}
        assert data_equivalence(ours.reset(seed=seed, options=options), ref.reset(seed=seed, options=options), exact=True)
        arng = np.random.default_rng(aseed)
        for t in range(steps):
            a = arng.integers(0, 2, n)
            s1, s2 = ours.step(a), ref.step(a)
            for k in range(5):
                assert data_equivalence(s1[k], s2[k], exact=True), (t, k, s1[k], s2[k])
            if sb:
                assert np.array_equal(np.signbit(s1[1]), np.signbit(s2[1])), t
        assert ours.np_random.bit_generator.state == ref.np_random.bit_generator.state
        ours.close(), ref.close()

    run()


@needs_gymnasium
def test_sync_semantics_property_based(oracle_factory):
    """The per-sub-environment mode against gymnasium's SyncVectorEnv with everything drawn by hypothesis: env id, autoreset mode, batch size, TimeLimit (down
    to 1), integer seeds and seed lists with holes, and partial resets (`reset_mask`) thrown in at random steps -- strict data_equivalence throughout."""
    pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st

    import gymnasium_amd  # noqa: F401

    @settings(max_examples=120, deadline=None, derandomize=True)
    @given(env_id=st.sampled_from(CLASSIC), mode=st.sampled_from(MODES), n=st.integers(1, 9), max_steps=st.integers(1, 30), seed=st.integers(0, 2**62),
           steps=st.integers(1, 50), data=st.data())
    def run(env_id, mode, n, max_steps, seed, steps, data):
        # the scalar envs' own constructor keywords, forwarded verbatim by make_vec on both sides
        kw = {"CartPole-v1": {"sutton_barto_reward": data.draw(st.booleans())}, "Pendulum-v1": {"g": data.draw(st.sampled_from([10.0, 9.81, 3.7]))},
              "MountainCar-v0": {"goal_velocity": data.draw(st.sampled_from([0, 0.02]))},
              "MountainCarContinuous-v0": {"goal_velocity": data.draw(st.sampled_from([0, 0.03]))}}.get(env_id, {})
        ours = gym.make_vec(f"MI355X/{env_id}", num_envs=n, autoreset_mode=mode, max_episode_steps=max_steps, _engine_factory=oracle_factory, **kw)
        ref = gym.make_vec(env_id, num_envs=n, vectorization_mode="sync", vector_kwargs={"autoreset_mode": mode}, max_episode_steps=max_steps, **kw)
        if data.draw(st.booleans()):  # a seed list with holes: the un-seeded sub-environments need a first full seeding to be comparable
            assert _same(ours.reset(seed=seed)[0], ref.reset(seed=seed)[0], env_id)
            seeds = [data.draw(st.one_of(st.none(), st.integers(0, 2**40))) for _ in range(n)]
            r1, r2 = ours.reset(seed=seeds), ref.reset(seed=seeds)
        else:
            r1, r2 = ours.reset(seed=seed), ref.reset(seed=seed)
        assert _same(r1[0], r2[0], env_id) and data_equivalence(r1[1], r2[1], exact=True)
        ref.action_space.seed(seed % 2**32)
        pending = np.zeros(n, bool)
        for t in range(steps):
            a = ref.action_space.sample()
            s1, s2 = ours.step(a), ref.step(a)
            done = s2[2] | s2[3]
            after_reset = pending if mode == "NextStep" else (done if mode == "SameStep" else np.zeros(n, bool))
            assert _same(s1[0], s2[0], env_id, after_reset), (t, s1[0], s2[0])
            for k in (1, 2, 3):
                assert data_equivalence(s1[k], s2[k], exact=True), (t, k)
            assert data_equivalence(dict(s1[4]), dict(s2[4]), exact=True), t
            pending = done if mode == "NextStep" else np.zeros(n, bool)
            mask = done.copy() if mode == "Disabled" else (np.array(data.draw(st.lists(st.booleans(), min_size=n, max_size=n))) if data.draw(st.integers(0, 9)) == 0 else np.zeros(n, bool))
            if mask.any():
                m1, m2 = ours.reset(options={"reset_mask": mask}), ref.reset(options={"reset_mask": mask})
                assert _same(m1[0], m2[0], env_id, mask) and data_equivalence(m1[1], m2[1], exact=True), t
                pending = pending & ~mask
        ours.close(), ref.close()

    run()


@needs_gymnasium
def test_toytext_property_based(oracle_factory):
    """The ToyText kinds against gymnasium's SyncVectorEnv with constructor kwargs drawn by hypothesis: FrozenLake on random boards (sizes 3-6, slippery or not,
    success rates), CliffWalking slippery or not, Taxi rainy / fickle, Blackjack natural / sab -- every autoreset mode, strict data_equivalence of observations
    (Blackjack's tuples included), rewards, flags and the info dicts with their dtype quirks."""
    pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st

    import gymnasium_amd  # noqa: F401
    from gymnasium.envs.toy_text.frozen_lake import generate_random_map

    kinds = st.one_of(
        st.tuples(st.just("FrozenLake-v1"), st.fixed_dictionaries({"size": st.integers(3, 6), "mapseed": st.integers(0, 1000), "is_slippery": st.booleans(),
                                                                  "success_rate": st.sampled_from([1.0 / 3.0, 0.5, 0.8])})),
        st.tuples(st.just("CliffWalking-v1"), st.fixed_dictionaries({"is_slippery": st.booleans()})),
        st.tuples(st.just("Taxi-v4"), st.fixed_dictionaries({"is_rainy": st.booleans(), "fickle_passenger": st.booleans()})),
        st.tuples(st.just("Blackjack-v1"), st.fixed_dictionaries({"natural": st.booleans(), "sab": st.booleans()})))

    @settings(max_examples=60, deadline=None, derandomize=True)
    @given(kind=kinds, mode=st.sampled_from(MODES), n=st.integers(1, 6), max_steps=st.integers(2, 40), seed=st.integers(0, 2**40), steps=st.integers(1, 60))
    def run(kind, mode, n, max_steps, seed, steps):
        env_id, kw = kind
        kw = dict(kw)
        if env_id == "FrozenLake-v1":
            kw["desc"] = generate_random_map(size=kw.pop("size"), p=0.8, seed=kw.pop("mapseed"))
        ours = gym.make_vec(f"MI355X/{env_id}", num_envs=n, autoreset_mode=mode, max_episode_steps=max_steps, _engine_factory=oracle_factory, **kw)
        ref = gym.make_vec(env_id, num_envs=n, vectorization_mode="sync", vector_kwargs={"autoreset_mode": mode}, max_episode_steps=max_steps, **kw)
        r1, r2 = ours.reset(seed=seed), ref.reset(seed=seed)
        assert data_equivalence(r1[0], r2[0], exact=True) and data_equivalence(r1[1], r2[1], exact=True), (r1, r2)
        ref.action_space.seed(seed % 2**32)
        for t in range(steps):
            a = ref.action_space.sample()
            s1, s2 = ours.step(a), ref.step(a)
            for k in range(4):
                assert data_equivalence(s1[k], s2[k], exact=True), (env_id, kw, t, k, s1[k], s2[k])
            assert data_equivalence(dict(s1[4]), dict(s2[4]), exact=True), (env_id, kw, t, s1[4], s2[4])
            done = s2[2] | s2[3]
            if mode == "Disabled" and done.any():
                m1, m2 = ours.reset(options={"reset_mask": done}), ref.reset(options={"reset_mask": done})
                assert data_equivalence(m1[0], m2[0], exact=True) and data_equivalence(m1[1], m2[1], exact=True), t
        ours.close(), ref.close()

    run()
