"""Backend-agnostic parity checks against the golden fixtures generated from the reference (tests/golden/).

Every check takes ``factory``: the oracle's engine factory (CPU tests: pins the oracle) or ``None`` (GPU tests:
the product's HIP engine through the C ABI), plus a tolerance profile:

  EXACT  bit-exact states / observations / rewards / flags (oracle vs reference; the one stated exception is
         Acrobot's cos/sin right after a reset, which NumPy evaluates with CPU-feature-dependent float32 SIMD
         kernels: 1 float32 ulp allowed there)
  FP     the stated floating-point tolerance for the GPU: rtol = atol = 1e-5 on observations and rewards -- the
         reference's own data_equivalence tolerance (gymnasium/utils/env_checker.py:68) -- and EXACT
         terminated / truncated flags.  (ocml sin/cos differ from glibc's by <= 1-2 ulp of float64.)
"""
import numpy as np

import gymnasium_amd
from conftest import ENV_IDS, golden

EXACT = dict(obs_tol=0.0, rew_tol=0.0, state_tol=0.0)
FP = dict(obs_tol=1e-5, rew_tol=1e-5, state_tol=1e-6)


def make(key, n, factory, **kw):
    return gymnasium_amd.make_vec(ENV_IDS[key], num_envs=n, _engine_factory=factory, **kw)


def _close(a, b, tol, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    if tol == 0.0:
        if not np.array_equal(a, b):
            bad = np.argwhere(a != b)
            raise AssertionError(f"{what}: {len(bad)} mismatches, first at {bad[0]}: {a[tuple(bad[0])]!r} vs {b[tuple(bad[0])]!r}")
    else:
        np.testing.assert_allclose(a, b, rtol=tol, atol=tol, err_msg=what)


def _ulp1_f32(a, b):
    """True where float32 a and b are equal or adjacent."""
    ai, bi = a.astype(np.float32).view(np.int32).astype(np.int64), b.astype(np.float32).view(np.int32).astype(np.int64)
    return np.abs(ai - bi) <= 1


def check_rollout(key, factory, tol):
    """gym.make_vec(id, 8, 'sync') trajectory: reset(seed=7), action_space.seed(11), T random steps."""
    g = golden(f"rollout_{key}.npz")
    T, n = g["actions"].shape[0], g["obs0"].shape[0]
    env = make(key, n, factory)
    assert env.metadata["autoreset_mode"] == gymnasium_amd.AutoresetMode.NEXT_STEP
    obs, info = env.reset(seed=7)
    assert info == {} and obs.dtype == np.float32 and obs.shape == g["obs0"].shape
    acro = key == "acrobot"

    def cmp_obs(o, ref, t, after_reset_rows):
        if acro and after_reset_rows.any():
            # float32 trig right after a reset: 1 ulp (see module docstring)
            rows = after_reset_rows
            assert _ulp1_f32(o[rows][:, :4], ref[rows][:, :4]).all(), f"acrobot reset obs t={t}"
            _close(o[rows][:, 4:], ref[rows][:, 4:], tol["obs_tol"], f"obs t={t}")
            _close(o[~rows], ref[~rows], tol["obs_tol"], f"obs t={t}")
        else:
            _close(o, ref, tol["obs_tol"], f"{key} obs t={t}")

    cmp_obs(obs, g["obs0"], -1, np.ones(n, dtype=bool))
    st, _, _ = env.get_state()
    _close(st, g["state0"], tol["state_tol"], "state after reset")
    env.action_space.seed(11)
    prev_done = np.zeros(n, dtype=bool)
    for t in range(T):
        a = env.action_space.sample()
        assert np.array_equal(a, g["actions"][t]), "action_space.sample() diverged from the reference"
        o, r, te, tr, info = env.step(a)
        assert r.dtype == np.float64 and te.dtype == np.bool_ and tr.dtype == np.bool_ and info == {}
        assert np.array_equal(te, g["term"][t]), f"{key} terminated t={t}"
        assert np.array_equal(tr, g["trunc"][t]), f"{key} truncated t={t}"
        cmp_obs(o, g["obs"][t], t, prev_done)
        _close(r, g["reward"][t], tol["rew_tol"], f"{key} reward t={t}")
        if t % 25 == 0 or t == T - 1:
            st, _, fl = env.get_state()
            _close(st, g["state"][t], tol["state_tol"], f"{key} state t={t}")
            if key == "mountaincar_continuous":
                assert np.array_equal((fl & 2) != 0, g["f32"][t])
        prev_done = te | tr
    stats = env.statistics()
    n_reset = int((g["term"] | g["trunc"])[:-1].sum())
    assert stats["reset_steps"] == n_reset and stats["env_steps"] == T * n - n_reset
    assert stats["episodes"] == int((g["term"] | g["trunc"]).sum())
    env.close()


def check_config1(factory, tol):
    """BASELINE.json configs[0]: CartPole-v1, 4 envs, seed 0, 1000 random steps -> sum(r)=3819, 181 terminations."""
    g = golden("config1_cartpole.npz")
    env = make("cartpole", 4, factory)
    obs, _ = env.reset(seed=0)
    _close(obs, g["obs0"], tol["obs_tol"], "reset obs")
    env.action_space.seed(0)
    total, nterm = 0.0, 0
    for t in range(1000):
        o, r, te, tr, _ = env.step(env.action_space.sample())
        total += r.sum()
        nterm += int(te.sum())
        assert np.array_equal(te, g["term"][t]) and np.array_equal(tr, g["trunc"][t])
        _close(o, g["obs"][t], tol["obs_tol"], f"obs t={t}")
    assert total == 3819.0 and nterm == 181
    env.close()


def check_appendix_c(factory, tol):
    """SURVEY.md Appendix C known answers for the other four ids (4 envs, seed 0, action seed 0)."""
    g = golden("appendix_c.npz")
    for key, T in (("pendulum", 450), ("acrobot", 1100), ("mountaincar_continuous", 2100), ("mountaincar", 450)):
        env = make(key, 4, factory)
        env.reset(seed=0)
        env.action_space.seed(0)
        tot, nte, ntr = 0.0, 0, 0
        for _ in range(T):
            o, r, te, tr, _ = env.step(env.action_space.sample())
            tot, nte, ntr = tot + r.sum(), nte + int(te.sum()), ntr + int(tr.sum())
        ref = g[key]
        assert (nte, ntr) == (int(ref[1]), int(ref[2])), key
        # a sum of T*4 rewards: only the summation order differs (running sum here, pairwise np.sum in the fixture)
        np.testing.assert_allclose(tot, ref[0], rtol=1e-12 if tol["rew_tol"] == 0.0 else 1e-6)
        _close(o[0], ref[3:].astype(np.float32), tol["obs_tol"], f"{key} final obs")
        env.close()


def check_modes(factory, tol):
    """SAME_STEP (final_obs) and DISABLED (+ reset_mask) autoreset, sync_vector_env.py:293-319,214-246."""
    g = golden("modes_cartpole.npz")
    for mode in ("SameStep", "Disabled"):
        n, T = g[f"{mode}_obs0"].shape[0], g[f"{mode}_actions"].shape[0]
        env = make("cartpole", n, factory, autoreset_mode=mode)
        assert env.metadata["autoreset_mode"].value == mode
        obs, _ = env.reset(seed=3)
        _close(obs, g[f"{mode}_obs0"], tol["obs_tol"], "reset obs")
        for t in range(T):
            o, r, te, tr, info = env.step(g[f"{mode}_actions"][t])
            assert np.array_equal(te, g[f"{mode}_term"][t]) and np.array_equal(tr, g[f"{mode}_trunc"][t])
            _close(r, g[f"{mode}_reward"][t], tol["rew_tol"], "reward")
            fm = g[f"{mode}_final_mask"][t]
            if mode == "SameStep":
                if fm.any():
                    assert np.array_equal(info["_final_obs"], fm) and np.array_equal(info["_final_info"], fm)
                    for i in np.flatnonzero(fm):
                        _close(info["final_obs"][i], g[f"{mode}_final_obs"][t][i], tol["obs_tol"], "final_obs")
                    assert all(info["final_obs"][i] is None for i in np.flatnonzero(~fm))
                else:
                    assert "final_obs" not in info
            rm = g[f"{mode}_reset_mask"][t]
            if rm.any():
                o, _ = env.reset(options={"reset_mask": rm.copy()})
            _close(o, g[f"{mode}_obs"][t], tol["obs_tol"], f"{mode} obs t={t}")
        env.close()
    # DISABLED: stepping a finished sub-env without resetting it is an error (sync_vector_env.py:295 assert)
    env = make("cartpole", 2, factory, autoreset_mode="Disabled", max_episode_steps=3)
    env.reset(seed=0)
    for _ in range(3):
        _, _, te, tr, _ = env.step(np.zeros(2, dtype=np.int64))
    assert tr.all()
    try:
        env.step(np.zeros(2, dtype=np.int64))
        raise RuntimeError("expected an assertion")
    except AssertionError:
        pass
    env.close()


def check_options(factory, tol):
    g = golden("options.npz")
    env = make("cartpole", 5, factory)
    o, _ = env.reset(seed=123, options={"low": -0.1, "high": 0.1})
    _close(o, g["cartpole_bounds"], tol["obs_tol"], "cartpole bounds")
    o, _ = env.reset(seed=[5, 9, 1, 1000000, 77])
    _close(o, g["cartpole_seedlist"], tol["obs_tol"], "cartpole seed list")
    env.close()
    env = make("pendulum", 5, factory)
    o, _ = env.reset(seed=123, options={"x_init": 1.0, "y_init": 0.5})
    _close(o, g["pendulum_init"], tol["obs_tol"], "pendulum x_init/y_init")
    o, _ = env.reset(seed=42)
    _close(o, g["pendulum_default"], tol["obs_tol"], "pendulum seed 42")
    env.close()
    env = make("acrobot", 5, factory)
    o, _ = env.reset(seed=123, options={"low": -0.2, "high": 0.3})
    ref = g["acrobot_bounds"]
    assert _ulp1_f32(o[:, :4], ref[:, :4]).all() if tol["obs_tol"] == 0.0 else True
    _close(o[:, 4:], ref[:, 4:], tol["obs_tol"], "acrobot bounds")
    _close(o, ref, max(tol["obs_tol"], 1e-6), "acrobot bounds")
    env.close()
    for key in ("mountaincar", "mountaincar_continuous"):
        env = make(key, 5, factory)
        o, _ = env.reset(seed=123, options={"low": -0.55, "high": -0.45})
        _close(o, g[f"{key}_bounds"], tol["obs_tol"], f"{key} bounds")
        env.close()
    env = make("cartpole", 3, factory, sutton_barto_reward=True)
    env.reset(seed=2)
    env.action_space.seed(2)
    R = np.stack([env.step(env.action_space.sample())[1] for _ in range(120)])
    _close(R, g["cartpole_sutton_reward"], 0.0, "sutton_barto reward")
    env.close()
    env = make("pendulum", 3, factory, g=9.81)
    env.reset(seed=2)
    env.action_space.seed(2)
    for t in range(50):
        o, r, _, _, _ = env.step(env.action_space.sample())
        _close(o, g["pendulum_g981_obs"][t], tol["obs_tol"], "pendulum g=9.81 obs")
        _close(r, g["pendulum_g981_reward"][t], tol["rew_tol"], "pendulum g=9.81 reward")
    env.close()


def check_episode_stats(factory, tol):
    """On-device RecordEpisodeStatistics == gymnasium.wrappers.vector.RecordEpisodeStatistics r / l."""
    g = golden("episode_stats.npz")
    for mode in ("NextStep", "SameStep"):
        env = make("cartpole", 6, factory, autoreset_mode=mode, record_episode_statistics=True)
        env.reset(seed=3)
        env.action_space.seed(5)
        count = 0
        for t in range(300):
            _, _, _, _, info = env.step(env.action_space.sample())
            m = g[f"{mode}_mask"][t]
            if m.any():
                assert np.array_equal(info["_episode"], m)
                _close(info["episode"]["r"], g[f"{mode}_r"][t], 0.0, "episode r")
                assert np.array_equal(info["episode"]["l"], g[f"{mode}_l"][t])
                assert (info["episode"]["t"][~m] == 0).all() and (info["episode"]["t"][m] >= 0).all()
                count += int(m.sum())
            else:
                assert "episode" not in info
        assert env.episode_count == count
        env.close()
        # the wrapper class around a plain env: same rows under its own key, the reference's bounded queues (common.py:72-109,214-217)
        from gymnasium_amd.wrappers import RecordEpisodeStatistics

        for kw, tag in ((dict(), ""), (dict(buffer_length=7, stats_key="ep"), "short_")):
            env = RecordEpisodeStatistics(make("cartpole", 6, factory, autoreset_mode=mode), **kw)
            key = kw.get("stats_key", "episode")
            env.reset(seed=3)
            env.action_space.seed(5)
            for t in range(300):
                _, _, _, _, info = env.step(env.action_space.sample())
                m = g[f"{mode}_mask"][t]
                assert (key in info) == bool(m.any()) and (key == "episode" or "episode" not in info)
                if m.any():
                    assert np.array_equal(info["_" + key], m) and np.array_equal(info[key]["l"], g[f"{mode}_l"][t])
                    _close(info[key]["r"], g[f"{mode}_r"][t], 0.0, "episode r (wrapper)")
            assert env.episode_count == int(g[f"{mode}_episode_count"])
            assert np.array_equal(np.array(env.return_queue), g[f"{mode}_{tag}return_queue"])
            assert np.array_equal(np.array(env.length_queue), g[f"{mode}_{tag}length_queue"])
            assert len(env.time_queue) == len(env.return_queue) and min(env.time_queue) >= 0
            env.close()


def check_teacher(key, factory, tol, fixture="teacher"):
    """Teacher-forced single steps from random (state, action) pairs covering the whole state box (fixture="teacher_wide": from states no trajectory
    reaches, tests/wide_states.py -- the reference's own numbers for what tests/test_gpu_wide_states.py compares with the oracle)."""
    g = golden(f"{fixture}_{key}.npz")
    M = g["state"].shape[0]
    env = make(key, M, factory, max_episode_steps=10**6, autoreset_mode="Disabled")
    env.reset(seed=0)
    flags = np.zeros(M, dtype=np.uint8)
    if key == "mountaincar_continuous":
        flags = np.where(g["f32"], 2, 0).astype(np.uint8)
    env.set_state(g["state"], np.zeros(M, dtype=np.int32), flags)
    o, r, te, tr, _ = env.step(g["action"])
    st, el, _ = env.get_state()
    assert np.array_equal(te, g["term"]), f"{key}: terminated mismatches {np.flatnonzero(te != g['term'])[:10]}"
    assert not tr.any() and (el == 1).all()
    _close(st, g["next_state"], tol["state_tol"], f"{key} next_state")
    _close(o, g["obs"], tol["obs_tol"], f"{key} obs")
    _close(r, g["reward"], tol["rew_tol"], f"{key} reward")
    env.close()


def check_rng(factory):
    """Device/oracle SeedSequence+PCG64 == NumPy's for 32-, 33-.. and 64-bit seeds (utils/seeding.py:39-41)."""
    g = golden("rng_golden.npz")
    seeds = [int(s) for s in g["seeds"]]
    for k, s in enumerate(seeds):
        env = make("cartpole", 1, factory)
        o, _ = env.reset(seed=s)
        # the 4 reset draws consumed 4 PCG64 steps: compare the obs with NumPy's uniform(-0.05, 0.05, 4)
        assert np.array_equal(o[0], g["uniform"][k][:4].astype(np.float32)), f"seed {s}"
        env.close()
    # words right after seeding (before any draw), through the seed_sequence path with an index offset
    env = make("cartpole", 3, factory, env_index_offset=41)
    env._seed_engines(0, None)
    w = env.get_rng_state()
    ref = {int(s): g["pcg"][k] for k, s in enumerate(g["seeds"])}
    assert np.array_equal(w[1], ref[42])
    env.close()


def check_action_samples():
    g = golden("action_samples.npz")
    for key in ENV_IDS:
        env = make(key, 8, _dummy_factory)
        env.action_space.seed(11)
        for k in range(4):
            a = env.action_space.sample()
            assert a.dtype == g[key].dtype and np.array_equal(a, g[key][k]), key
        env.close()


def _dummy_factory(kind, num_envs, *a, **kw):
    class _E:
        act_dtype = np.int64 if kind in ("cartpole", "acrobot", "mountain_car") else np.float32
        obs_dtype = np.float32
        obs_dim = {"cartpole": 4, "pendulum": 3, "acrobot": 6}.get(kind, 2)
        act_dim, state_dim, info_dim = 1, 2, 0

        def close(self):
            pass

    return _E()


def check_rollout_fused(key, factory, n=64, T=40, **kw):
    """rollout(T) with on-device action sampling == T x step(action_space.sample()) on the same backend."""
    import torch

    a = make(key, n, factory, output="torch", **kw)
    b = make(key, n, factory, output="torch", **kw)
    a.reset(seed=5), b.reset(seed=5)
    a.action_space.seed(9), b.action_space.seed(9)
    out = a.rollout(T)
    for t in range(T):
        act = b.action_space.sample()
        o, r, te, tr, _ = b.step(act)
        assert np.array_equal(out["actions"][t].cpu().numpy().reshape(act.shape), act), f"{key} sampled actions t={t}"
        assert torch.equal(out["obs"][t], o) and torch.equal(out["rewards"][t], r)
        assert torch.equal(out["terminations"][t], te) and torch.equal(out["truncations"][t], tr)
    # generators stay in lockstep after the rollout
    assert np.array_equal(a.action_space.sample(), b.action_space.sample())
    sa, sb = a.get_state(), b.get_state()
    assert all(np.array_equal(x, y) for x, y in zip(sa, sb))
    assert np.array_equal(a.get_rng_state(), b.get_rng_state())
    a.close(), b.close()


TOYTEXT_IDS = {"frozenlake": "FrozenLake-v1", "frozenlake8x8": "FrozenLake8x8-v1", "cliffwalking": "CliffWalking-v1",
               "cliffwalking_slippery": "CliffWalkingSlippery-v1", "taxi": "Taxi-v4"}
# constructor variants: the same kwargs tests/golden/make_golden.py gave the reference
TOYTEXT_VARIANTS = {"taxi_rainy": ("Taxi-v4", {"is_rainy": True}), "taxi_fickle": ("Taxi-v4", {"fickle_passenger": True}),
                    "taxi_rainy_fickle": ("Taxi-v4", {"is_rainy": True, "fickle_passenger": True, "rainy_probability": 0.7, "fickle_probability": 0.6}),
                    "frozenlake_random": ("FrozenLake-v1", {"desc": ("random", 6, 0.75, 5), "is_slippery": True})}
TOYTEXT_ALL = list(TOYTEXT_IDS) + list(TOYTEXT_VARIANTS)


def toytext_spec(key):
    if key in TOYTEXT_IDS:
        return TOYTEXT_IDS[key], {}
    env_id, kw = TOYTEXT_VARIANTS[key]
    kw = dict(kw)
    if isinstance(kw.get("desc"), tuple):
        from gymnasium_amd.envs.toy_text import generate_random_map

        _, size, p, seed = kw["desc"]
        kw["desc"] = generate_random_map(size=size, p=p, seed=seed)
    return env_id, kw


def check_toytext(key, factory):
    """ToyText: transition tables equal the reference's P, and the gym.make_vec(id, 8, 'sync') trajectory (reset(seed=7),
    action_space.seed(11), 400 random steps) is reproduced BIT-EXACTLY incl. info['prob'] / info['action_mask']."""
    g = golden(f"toytext_{key}.npz")
    n = g["obs0"].shape[0]
    env_id, kw = toytext_spec(key)
    env = gymnasium_amd.make_vec(env_id, num_envs=n, _engine_factory=factory, **kw)
    tab = env._tab
    assert np.array_equal(tab["count"], g["count"]) and np.array_equal(tab["next_state"], g["next_state"])
    assert np.array_equal(tab["prob"], g["prob"]) and np.array_equal(tab["reward"], g["reward_table"])
    assert np.array_equal(tab["terminated"], g["terminated_table"]) and np.array_equal(env.initial_state_distrib, g["isd"])
    obs, info = env.reset(seed=7)
    assert obs.dtype == np.int64 and np.array_equal(obs, g["obs0"]) and np.array_equal(info["prob"], g["prob0"])
    if "action_mask0" in g.files:
        assert np.array_equal(info["action_mask"], g["action_mask0"])
    env.action_space.seed(11)
    for t in range(g["actions"].shape[0]):
        a = env.action_space.sample()
        assert np.array_equal(a, g["actions"][t])
        o, r, te, tr, info = env.step(a)
        assert np.array_equal(o, g["obs"][t]) and np.array_equal(r, g["reward"][t]), f"{key} t={t}"
        assert np.array_equal(te, g["term"][t]) and np.array_equal(tr, g["trunc"][t]), f"{key} flags t={t}"
        assert np.array_equal(info["prob"], g["prob_info"][t]) and np.array_equal(info["_prob"], g["prob_mask"][t])
        if "action_mask" in g.files:
            assert np.array_equal(info["action_mask"], g["action_mask"][t])
    assert np.array_equal(env.get_rng_state(), g["rng_after"])
    env.close()



def check_blackjack(key, factory):
    """Blackjack-v1 against the reference recording (tests/golden/toytext_blackjack_<key>.npz): every observation, reward, flag
    and the final generator states, bit for bit (integer card game: Generator.choice draws, dealer play-out, natural / sab rules)."""
    import pytest

    g = golden(f"toytext_blackjack_{key}.npz")
    env = gymnasium_amd.make_vec("Blackjack-v1", num_envs=8, natural=bool(g["natural"]), sab=bool(g["sab"]), _engine_factory=factory)
    obs, info = env.reset(seed=21)
    assert isinstance(obs, tuple) and len(obs) == 3 and all(o.dtype == np.int64 for o in obs) and info == {}
    assert np.array_equal(np.stack(obs), g["obs0"])
    for t in range(g["actions"].shape[0]):
        obs, r, te, tr, info = env.step(g["actions"][t])
        assert np.array_equal(np.stack(obs), g["obs"][t]), t
        assert np.array_equal(r, g["reward"][t]) and r.dtype == np.float64, t
        assert np.array_equal(te, g["term"][t]) and np.array_equal(tr, g["trunc"][t]), t
    assert np.array_equal(env.get_rng_state(), g["rng_after"])
    with pytest.raises(AssertionError):
        env.step(np.full(8, 2))
    env.close()


def check_same_step_infos(key, factory):
    """SAME_STEP info dicts against the reference recording (tests/golden/infos_same_step_<key>.npz): the finished sub-env's top-level
    entries are its RESET info, the finishing step's info sits under final_info with its own masks (sync_vector_env.py:302-319)."""
    g = golden(f"infos_same_step_{key}.npz")
    n = g["obs0"].shape[0]
    env = gymnasium_amd.make_vec(TOYTEXT_IDS[key], num_envs=n, autoreset_mode="SameStep", _engine_factory=factory)
    obs, _ = env.reset(seed=13)
    assert np.array_equal(obs, g["obs0"])
    env.action_space.seed(5)
    finals = 0
    for t in range(g["actions"].shape[0]):
        a = env.action_space.sample()
        assert np.array_equal(a, g["actions"][t])
        o, r, te, tr, info = env.step(a)
        assert np.array_equal(o, g["obs"][t]) and np.array_equal(r, g["reward"][t]) and np.array_equal(te, g["term"][t]) and np.array_equal(tr, g["trunc"][t]), t
        assert np.array_equal(info["prob"], g["prob"][t]) and np.array_equal(info["_prob"], g["prob_mask"][t]), t
        assert bool(np.issubdtype(info["prob"].dtype, np.integer)) == bool(g["prob_is_int"][t]), t
        assert ("final_info" in info) == bool(g["has_final"][t]), t
        if "action_mask" in g.files:
            assert np.array_equal(info["action_mask"], g["action_mask"][t]), t
        if g["has_final"][t]:
            finals += 1
            assert np.array_equal(info["_final_info"], g["final_mask"][t]) and np.array_equal(info["_final_obs"], g["final_mask"][t])
            fo = np.array([-1 if x is None else int(x) for x in info["final_obs"]])
            assert np.array_equal(fo, g["final_obs"][t]), t
            fi = info["final_info"]
            assert np.array_equal(fi["prob"], g["final_prob"][t]) and np.array_equal(fi["_prob"], g["final_prob_mask"][t]), t
            assert fi["prob"].dtype == np.float64
            if "final_action_mask" in g.files:
                assert np.array_equal(fi["action_mask"], g["final_action_mask"][t]) and np.array_equal(fi["_action_mask"], g["final_action_mask_mask"][t]), t
    assert finals > 0
    env.close()


def check_partial_reset_infos(factory):
    """NEXT_STEP: an explicit reset of SOME sub-envs (options['reset_mask']) while another one is waiting for its autoreset step
    (sync_vector_env.py:232-234 clears only the masked entries): the un-reset sub-env still supplies its reset info next step."""
    g = golden("infos_partial_reset_frozenlake.npz")
    n = g["actions"].shape[1]
    env = gymnasium_amd.make_vec("FrozenLake-v1", num_envs=n, max_episode_steps=5, _engine_factory=factory)
    env.reset(seed=3)
    env.action_space.seed(9)
    resets = list(g["reset_at"])
    for t in range(g["actions"].shape[0]):
        a = env.action_space.sample()
        assert np.array_equal(a, g["actions"][t])
        o, r, te, tr, info = env.step(a)
        assert np.array_equal(o, g["obs"][t]) and np.array_equal(te, g["term"][t]) and np.array_equal(tr, g["trunc"][t]), t
        assert np.array_equal(info["prob"], g["prob"][t]) and np.array_equal(info["_prob"], g["prob_mask"][t]), t
        assert bool(np.issubdtype(info["prob"].dtype, np.integer)) == bool(g["prob_is_int"][t]), t
        if t in resets:
            ro, _ = env.reset(options={"reset_mask": g["reset_mask"]})
            assert np.array_equal(ro, g["reset_obs"][resets.index(t)])
    env.close()
