#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ FROM THE REFERENCE ITSELF.

Run in the build container only (the reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports gymnasium 1.4.0 from /root/reference (NumPy >= 2 required: the classic-control envs depend on NEP-50
weak-scalar promotion) and records

  rng_golden.npz            NumPy SeedSequence/PCG64/uniform known answers (utils/seeding.py:39-41)
  rollout_<env>.npz         gym.make_vec(id, 8, "sync") trajectories, NEXT_STEP autoreset, random policy from
                            action_space.seed(...) (vector/sync_vector_env.py:187-337), per-step scalar-env state
  modes_cartpole.npz        SAME_STEP / DISABLED autoreset trajectories incl. final_obs and reset_mask
  options.npz               reset(options=...) bounds, seed lists
  episode_stats.npz         wrappers.vector.RecordEpisodeStatistics r / l
  teacher_<env>.npz         teacher-forced single steps from random (state, action) pairs
  teacher_wide_<env>.npz    the same from states no trajectory reaches (tests/wide_states.py); --teacher-wide-only regenerates just these
  config1_cartpole.npz      BASELINE.json configs[0]: CartPole-v1, Sync, 4 envs, 1000 steps, seed 0
  wrappers_*.npz            NormalizeObservation / NormalizeReward / ClipReward (wrappers/vector) inputs and outputs
  toytext_<env>.npz         FrozenLake / CliffWalking / Taxi: the reference's transition table P and initial distribution,
                            plus a gym.make_vec(id, 8, "sync") trajectory with the info dict entries (prob, action_mask)

  toytext_frozenlake_per_env_maps.npz   SyncVectorEnv over FrozenLake envs with one (seeded random) map each: per-sub-environment transition tables
  cartpole_vector_entry_point.npz   the reference's own NumPy CartPoleVectorEnv (one shared generator, float32 rewards): what `rng="shared"` reproduces

  infos_*.npz               SAME_STEP info dicts (final_info / reset info at top level) and a partial reset during a pending autoreset

Nothing here is read at run time by the product; tests compare the oracle (oracle/) and the HIP engine to it.
"""
import os
import sys

REF = os.environ.get("GYM_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import numpy as np  # noqa: E402

import gymnasium as gym  # noqa: E402
from gymnasium.vector import AutoresetMode  # noqa: E402

assert int(np.__version__.split(".")[0]) >= 2, "golden vectors must be generated with NumPy >= 2"
OUT = os.path.dirname(os.path.abspath(__file__))

ENVS = {
    "cartpole": ("CartPole-v1", 600),
    "pendulum": ("Pendulum-v1", 450),
    "acrobot": ("Acrobot-v1", 1100),
    "mountaincar": ("MountainCar-v0", 450),
    "mountaincar_continuous": ("MountainCarContinuous-v0", 2100),
}


def pcg_words(gen):
    st = gen.bit_generator.state["state"]
    s, i = st["state"], st["inc"]
    m = (1 << 64) - 1
    return np.array([s >> 64, s & m, i >> 64, i & m], dtype=np.uint64)


def scalar_states(vec):
    out = []
    for e in vec.envs:
        s = e.unwrapped.state
        out.append(np.asarray(s, dtype=np.float64).ravel())
    return np.stack(out)


def state_is_f32(vec):
    return np.array([getattr(e.unwrapped.state, "dtype", None) == np.float32 for e in vec.envs])


def rollout(env_id, n, T, seed, aseed, **vector_kwargs):
    vec = gym.make_vec(env_id, num_envs=n, vectorization_mode="sync", vector_kwargs=vector_kwargs or None)
    obs0, _ = vec.reset(seed=seed)
    vec.action_space.seed(aseed)
    rec = dict(obs0=obs0, state0=scalar_states(vec), actions=[], obs=[], reward=[], term=[], trunc=[], state=[],
               f32=[], final_obs=[], final_mask=[])
    for _ in range(T):
        a = vec.action_space.sample()
        o, r, te, tr, info = vec.step(a)
        rec["actions"].append(a), rec["obs"].append(o), rec["reward"].append(r)
        rec["term"].append(te), rec["trunc"].append(tr), rec["state"].append(scalar_states(vec))
        rec["f32"].append(state_is_f32(vec))
        fo = np.zeros_like(o)
        fm = np.zeros(n, dtype=bool)
        if "final_obs" in info:
            fm = info["_final_obs"].copy()
            for i in np.where(fm)[0]:
                fo[i] = info["final_obs"][i]
        rec["final_obs"].append(fo), rec["final_mask"].append(fm)
    vec.close()
    return {k: (np.stack(v) if isinstance(v, list) else v) for k, v in rec.items()}


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def make_rng():
    seeds = [0, 1, 2, 3, 42, 123, 65535, 65536, 2**31, 2**32 - 1, 2**32, 2**32 + 5, 2**40 + 17, 2**63 - 1, 2**64 - 1]
    words, raw, uni, ss = [], [], [], []
    for s in seeds:
        g, _ = gym.utils.seeding.np_random(s)
        words.append(pcg_words(g))
        ss.append(np.random.SeedSequence(s).generate_state(4, np.uint64))
        g2, _ = gym.utils.seeding.np_random(s)
        raw.append(g2.bit_generator.random_raw(8))
        g3, _ = gym.utils.seeding.np_random(s)
        uni.append(np.concatenate([g3.uniform(-0.05, 0.05, size=(4,)), g3.uniform([-np.pi, -1.0], [np.pi, 1.0]), g3.random(2)]))
    save("rng_golden.npz", seeds=np.array(seeds, dtype=np.uint64), pcg=np.stack(words), seedseq=np.stack(ss),
         raw=np.stack(raw), uniform=np.stack(uni))


def make_rollouts():
    for key, (env_id, T) in ENVS.items():
        rec = rollout(env_id, 8, T, seed=7, aseed=11)
        save(f"rollout_{key}.npz", **rec)


def make_config1():
    rec = rollout("CartPole-v1", 4, 1000, seed=0, aseed=0)
    assert rec["reward"].sum() == 3819.0 and rec["term"].sum() == 181
    save("config1_cartpole.npz", obs0=rec["obs0"], actions=rec["actions"], obs=rec["obs"], reward=rec["reward"],
         term=rec["term"], trunc=rec["trunc"])
    # Appendix C of SURVEY.md for the other ids (sum of rewards, #terminated, #truncated, final obs[0])
    summ = {}
    for key, (env_id, T) in {"pendulum": ("Pendulum-v1", 450), "acrobot": ("Acrobot-v1", 1100),
                             "mountaincar_continuous": ("MountainCarContinuous-v0", 2100),
                             "mountaincar": ("MountainCar-v0", 450)}.items():
        r = rollout(env_id, 4, T, seed=0, aseed=0)
        summ[key] = np.array([r["reward"].sum(), r["term"].sum(), r["trunc"].sum()] + list(r["obs"][-1][0]), dtype=np.float64)
    save("appendix_c.npz", **summ)


def make_modes():
    out = {}
    for mode in ("SameStep", "Disabled"):
        n, T = 6, 300
        vec = gym.make_vec("CartPole-v1", num_envs=n, vectorization_mode="sync", vector_kwargs={"autoreset_mode": mode})
        obs0, _ = vec.reset(seed=3)
        vec.action_space.seed(5)
        A, O, R, TE, TR, FO, FM, RM = [], [], [], [], [], [], [], []
        for _ in range(T):
            a = vec.action_space.sample()
            o, r, te, tr, info = vec.step(a)
            fo, fm = np.zeros_like(o), np.zeros(n, dtype=bool)
            if "final_obs" in info:
                fm = info["_final_obs"].copy()
                for i in np.where(fm)[0]:
                    fo[i] = info["final_obs"][i]
            rm = np.zeros(n, dtype=bool)
            if mode == "Disabled":
                rm = np.logical_or(te, tr)
                if rm.any():
                    # the user resets finished sub-envs explicitly (sync_vector_env.py:214-246)
                    o2, _ = vec.reset(options={"reset_mask": rm.copy()})
                    o = o2  # what the caller holds after the masked reset
            A.append(a), O.append(o), R.append(r), TE.append(te), TR.append(tr), FO.append(fo), FM.append(fm), RM.append(rm)
        vec.close()
        for k, v in dict(obs0=obs0, actions=A, obs=O, reward=R, term=TE, trunc=TR, final_obs=FO, final_mask=FM, reset_mask=RM).items():
            out[f"{mode}_{k}"] = np.stack(v) if isinstance(v, list) else v
    save("modes_cartpole.npz", **out)


def make_options():
    out = {}
    v = gym.make_vec("CartPole-v1", num_envs=5, vectorization_mode="sync")
    out["cartpole_bounds"], _ = v.reset(seed=123, options={"low": -0.1, "high": 0.1})
    out["cartpole_seedlist"], _ = v.reset(seed=[5, 9, 1, 1000000, 77])
    v.close()
    v = gym.make_vec("Pendulum-v1", num_envs=5, vectorization_mode="sync")
    out["pendulum_init"], _ = v.reset(seed=123, options={"x_init": 1.0, "y_init": 0.5})
    out["pendulum_default"], _ = v.reset(seed=42)  # doctest vector/sync_vector_env.py:40-57 uses seed=42
    v.close()
    v = gym.make_vec("Acrobot-v1", num_envs=5, vectorization_mode="sync")
    out["acrobot_bounds"], _ = v.reset(seed=123, options={"low": -0.2, "high": 0.3})
    v.close()
    for key, env_id in (("mountaincar", "MountainCar-v0"), ("mountaincar_continuous", "MountainCarContinuous-v0")):
        v = gym.make_vec(env_id, num_envs=5, vectorization_mode="sync")
        out[f"{key}_bounds"], _ = v.reset(seed=123, options={"low": -0.55, "high": -0.45})
        v.close()
    # kwargs through make_vec: sutton_barto_reward, g, goal_velocity
    v = gym.make_vec("CartPole-v1", num_envs=3, vectorization_mode="sync", sutton_barto_reward=True)
    v.reset(seed=2)
    v.action_space.seed(2)
    R = []
    for _ in range(120):
        _, r, _, _, _ = v.step(v.action_space.sample())
        R.append(r)
    out["cartpole_sutton_reward"] = np.stack(R)
    v.close()
    v = gym.make_vec("Pendulum-v1", num_envs=3, vectorization_mode="sync", g=9.81)
    v.reset(seed=2)
    v.action_space.seed(2)
    O, R = [], []
    for _ in range(50):
        o, r, _, _, _ = v.step(v.action_space.sample())
        O.append(o), R.append(r)
    out["pendulum_g981_obs"], out["pendulum_g981_reward"] = np.stack(O), np.stack(R)
    v.close()
    save("options.npz", **out)


def make_episode_stats():
    from gymnasium.wrappers.vector import RecordEpisodeStatistics
    out = {}
    for mode in ("NextStep", "SameStep"):
        vec = gym.make_vec("CartPole-v1", num_envs=6, vectorization_mode="sync", vector_kwargs={"autoreset_mode": mode})
        vec = RecordEpisodeStatistics(vec)
        vec.reset(seed=3)
        vec.action_space.seed(5)
        Rr, Ll, M = [], [], []
        for _ in range(300):
            _, _, _, _, info = vec.step(vec.action_space.sample())
            if "episode" in info:
                Rr.append(info["episode"]["r"]), Ll.append(info["episode"]["l"]), M.append(info["_episode"])
            else:
                Rr.append(np.zeros(6)), Ll.append(np.zeros(6, dtype=int)), M.append(np.zeros(6, dtype=bool))
        out[f"{mode}_r"], out[f"{mode}_l"], out[f"{mode}_mask"] = np.stack(Rr), np.stack(Ll), np.stack(M)
        out[f"{mode}_return_queue"], out[f"{mode}_length_queue"] = np.array(vec.return_queue), np.array(vec.length_queue)
        out[f"{mode}_episode_count"] = np.array(vec.episode_count)
        vec.close()
        # the wrapper object itself with a short buffer and its own info key (wrappers/vector/common.py:72-109)
        vec = gym.make_vec("CartPole-v1", num_envs=6, vectorization_mode="sync", vector_kwargs={"autoreset_mode": mode})
        vec = RecordEpisodeStatistics(vec, buffer_length=7, stats_key="ep")
        vec.reset(seed=3)
        vec.action_space.seed(5)
        for _ in range(300):
            _, _, _, _, info = vec.step(vec.action_space.sample())
            assert "episode" not in info
        out[f"{mode}_short_return_queue"], out[f"{mode}_short_length_queue"] = np.array(vec.return_queue), np.array(vec.length_queue)
        vec.close()
    save("episode_stats.npz", **out)


def _teacher_run(env_id, states, actions, f32_state=None, tuple_state=False):
    env = gym.make(env_id).unwrapped
    env.reset(seed=0)
    ns, ob, rw, te = [], [], [], []
    for k in range(len(states)):
        s = states[k]
        if f32_state is not None and f32_state[k]:
            env.state = np.array(s, dtype=np.float32)
        elif tuple_state:
            env.state = (np.float64(s[0]), np.float64(s[1]))
        else:
            env.state = np.array(s, dtype=np.float64)
        if hasattr(env, "steps_beyond_terminated"):
            env.steps_beyond_terminated = None
        o, r, t, _, _ = env.step(actions[k])
        ns.append(np.asarray(env.state, dtype=np.float64).ravel()), ob.append(o), rw.append(r), te.append(t)
    return np.stack(ns), np.stack(ob), np.array(rw, dtype=np.float64), np.array(te, dtype=bool)


def make_teacher():
    """Teacher-forced single steps: poke env.unwrapped.state, step once, record everything."""
    rng = np.random.default_rng(2024)
    M = 3000
    run = _teacher_run

    # CartPole: around and beyond the thresholds
    s = np.stack([rng.uniform(-2.6, 2.6, M), rng.uniform(-3, 3, M), rng.uniform(-0.25, 0.25, M), rng.uniform(-3.5, 3.5, M)], 1)
    a = rng.integers(0, 2, M)
    ns, ob, rw, te = run("CartPole-v1", s, a)
    save("teacher_cartpole.npz", state=s, action=a, next_state=ns, obs=ob, reward=rw, term=te)
    # Pendulum: large unwrapped angles, speeds at the clip, actions beyond the torque limit
    s = np.stack([rng.uniform(-40, 40, M), rng.uniform(-8, 8, M)], 1)
    a = rng.uniform(-3, 3, (M, 1)).astype(np.float32)
    ns, ob, rw, te = run("Pendulum-v1", s, a)
    save("teacher_pendulum.npz", state=s, action=a, next_state=ns, obs=ob, reward=rw, term=te)
    # Acrobot: whole state box (wrap loops, velocity bounds, termination)
    s = np.stack([rng.uniform(-np.pi, np.pi, M), rng.uniform(-np.pi, np.pi, M), rng.uniform(-4 * np.pi, 4 * np.pi, M),
                  rng.uniform(-9 * np.pi, 9 * np.pi, M)], 1)
    a = rng.integers(0, 3, M)
    ns, ob, rw, te = run("Acrobot-v1", s, a)
    save("teacher_acrobot.npz", state=s, action=a, next_state=ns, obs=ob, reward=rw, term=te)
    # MountainCar: walls and goal
    s = np.stack([rng.uniform(-1.2, 0.6, M), rng.uniform(-0.07, 0.07, M)], 1)
    s[: M // 10, 0] = rng.choice([-1.2, -1.1995, 0.4995, 0.5, 0.6], M // 10)
    a = rng.integers(0, 3, M)
    ns, ob, rw, te = run("MountainCar-v0", s, a, tuple_state=True)
    save("teacher_mountaincar.npz", state=s, action=a, next_state=ns, obs=ob, reward=rw, term=te)
    # MountainCarContinuous: float32-held state (normal case) and float64-held state (first step after reset),
    # actions partly outside [-1, 1] (Python-float force path)
    s = np.stack([rng.uniform(-1.2, 0.6, M), rng.uniform(-0.07, 0.07, M)], 1)
    s[: M // 10, 0] = rng.choice([-1.2, -1.1995, 0.4495, 0.45, 0.6], M // 10)
    f32 = rng.random(M) < 0.7
    s[f32] = s[f32].astype(np.float32).astype(np.float64)
    a = rng.uniform(-1.3, 1.3, (M, 1)).astype(np.float32)
    ns, ob, rw, te = run("MountainCarContinuous-v0", s, a, f32_state=f32)
    save("teacher_mountaincar_continuous.npz", state=s, f32=f32, action=a, next_state=ns, obs=ob, reward=rw, term=te)


def make_teacher_wide():
    """The same, from states no trajectory reaches (tests/wide_states.py): CartPole's pole far beyond the threshold (|theta| to 1e5, the 0.855 boundary of
    the engine's short sin / cos routine from both sides, angular velocities to 1e8), Pendulum's unwrapped angle to 1e8 and within 3 ulps of the multiples
    of 2 pi, Acrobot through 16 turns of wrap() and velocities beyond bound(), MountainCar far outside the clips and at the wall / goal line."""
    sys.path.insert(0, os.path.dirname(OUT))
    import wide_states as w

    rng = np.random.default_rng(4048)
    M = 3000
    for key, env_id, states, kw in (("cartpole", "CartPole-v1", w.wide_states, {}), ("pendulum", "Pendulum-v1", w.wide_pendulum_states, {}),
                                    ("acrobot", "Acrobot-v1", w.wide_acrobot_states, {}), ("mountaincar", "MountainCar-v0", w.wide_mountaincar_states, {"tuple_state": True}),
                                    ("mountaincar_continuous", "MountainCarContinuous-v0", w.wide_mountaincar_states, {"f32_state": np.zeros(M, dtype=bool)})):
        s = states(M, 77)
        space = gym.make(env_id).action_space
        a = rng.uniform(-3, 3, (M, 1)).astype(np.float32) if key in ("pendulum", "mountaincar_continuous") else rng.integers(0, space.n, M)
        ns, ob, rw, te = _teacher_run(env_id, s, a, **kw)
        extra = {"f32": np.zeros(M, dtype=bool)} if key == "mountaincar_continuous" else {}
        save(f"teacher_wide_{key}.npz", state=s, action=a, next_state=ns, obs=ob, reward=rw, term=te, **extra)


def make_action_samples():
    out = {}
    for key, (env_id, _) in ENVS.items():
        vec = gym.make_vec(env_id, num_envs=8, vectorization_mode="sync")
        vec.action_space.seed(11)
        out[f"{key}_pcg"] = pcg_words(vec.action_space.np_random)
        out[key] = np.stack([vec.action_space.sample() for _ in range(4)])
        vec.close()
    save("action_samples.npz", **out)


TOYTEXT = {"frozenlake": "FrozenLake-v1", "frozenlake8x8": "FrozenLake8x8-v1", "cliffwalking": "CliffWalking-v1",
           "cliffwalking_slippery": "CliffWalkingSlippery-v1", "taxi": "Taxi-v4",
           # the constructor variants (taxi.py:247-334 is_rainy, :436-451,:462-464 fickle_passenger; frozen_lake.py:56-83 generate_random_map)
           "taxi_rainy": ("Taxi-v4", {"is_rainy": True}), "taxi_fickle": ("Taxi-v4", {"fickle_passenger": True}),
           "taxi_rainy_fickle": ("Taxi-v4", {"is_rainy": True, "fickle_passenger": True, "rainy_probability": 0.7, "fickle_probability": 0.6}),
           "frozenlake_random": ("FrozenLake-v1", {"desc": "generate_random_map(size=6, p=0.75, seed=5)", "is_slippery": True})}


def make_random_maps():
    """generate_random_map (frozen_lake.py:56-83) for a grid of (size, p, seed): the maps themselves."""
    from gymnasium.envs.toy_text.frozen_lake import generate_random_map

    cases = [(4, 0.8, 0), (8, 0.8, 1), (8, 0.8, 123), (5, 0.6, 7), (12, 0.9, 42), (6, 0.75, 5), (3, 0.5, 9), (8, 1.5, 2)]
    save("frozenlake_random_maps.npz", cases=np.array(cases, dtype=np.float64), maps=np.array(["|".join(generate_random_map(size=s, p=p, seed=seed)) for s, p, seed in cases]))


def make_toytext():
    from gymnasium.envs.toy_text.frozen_lake import generate_random_map

    make_random_maps()
    for key, spec in TOYTEXT.items():
        env_id, kw = (spec, {}) if isinstance(spec, str) else spec
        kw = dict(kw)
        if isinstance(kw.get("desc"), str):
            kw["desc"] = generate_random_map(size=6, p=0.75, seed=5)
        e = gym.make(env_id, **kw).unwrapped
        nS, nA = e.observation_space.n, e.action_space.n
        K = max(len(e.P[s][a]) for s in range(nS) for a in range(nA))
        prob, nxt, rew, term = np.zeros((nS, nA, K)), np.zeros((nS, nA, K), np.int32), np.zeros((nS, nA, K)), np.zeros((nS, nA, K), np.uint8)
        count = np.zeros((nS, nA), np.int32)
        for s in range(nS):
            for a in range(nA):
                count[s, a] = len(e.P[s][a])
                for k, (p, ns, r, t) in enumerate(e.P[s][a]):
                    prob[s, a, k], nxt[s, a, k], rew[s, a, k], term[s, a, k] = p, ns, r, t
        n, T = 8, 400
        if "fickle" in key:
            T = 1500  # the fickle change of mind needs a pickup first: rare under a random policy
        v = gym.make_vec(env_id, num_envs=n, vectorization_mode="sync", **kw)
        obs0, info0 = v.reset(seed=7)
        v.action_space.seed(11)
        A, O, R, TE, TR, PR, PM, AM = [], [], [], [], [], [], [], []
        for _ in range(T):
            a = v.action_space.sample()
            o, r, te, tr, info = v.step(a)
            A.append(a), O.append(o), R.append(r), TE.append(te), TR.append(tr)
            PR.append(np.asarray(info["prob"], dtype=np.float64)), PM.append(info["_prob"])
            if "action_mask" in info:
                AM.append(np.stack([np.asarray(m) for m in info["action_mask"]]))
        extra = {"action_mask": np.stack(AM), "action_mask0": np.stack([np.asarray(m) for m in info0["action_mask"]])} if AM else {}
        save(f"toytext_{key}.npz", prob=prob, next_state=nxt, reward_table=rew, terminated_table=term, count=count,
             isd=np.asarray(e.initial_state_distrib, dtype=np.float64), obs0=obs0, prob0=np.asarray(info0["prob"], dtype=np.float64),
             actions=np.stack(A), obs=np.stack(O), reward=np.stack(R), term=np.stack(TE), trunc=np.stack(TR), prob_info=np.stack(PR),
             prob_mask=np.stack(PM), rng_after=np.stack([pcg_words(x.unwrapped.np_random) for x in v.envs]), **extra)
        v.close()


def make_frozenlake_per_env_maps():
    """SyncVectorEnv over FrozenLake envs that each have their OWN map -- what `make_vec("FrozenLake-v1", n, "sync", map_name=None)` builds
    (frozen_lake.py:241-242: every scalar env draws `generate_random_map()` from OS entropy), made reproducible by handing each sub-environment a seeded
    random map.  Two of the eight boards coincide on purpose (sub-environments may share a table)."""
    from gymnasium.envs.toy_text.frozen_lake import generate_random_map

    maps = [generate_random_map(size=6, p=0.8, seed=100 + (i if i != 5 else 2)) for i in range(8)]
    out = {"maps": np.array(maps)}
    for tag, slippery in (("slip", True), ("det", False)):
        v = gym.vector.SyncVectorEnv([(lambda d=d: gym.make("FrozenLake-v1", desc=d, is_slippery=slippery)) for d in maps])
        obs0, info0 = v.reset(seed=21)
        v.action_space.seed(22)
        A, O, R, TE, TR, PR, PM = [], [], [], [], [], [], []
        for _ in range(300):
            a = v.action_space.sample()
            o, r, te, tr, info = v.step(a)
            A.append(a), O.append(o), R.append(r), TE.append(te), TR.append(tr)
            PR.append(np.asarray(info["prob"], dtype=np.float64)), PM.append(info["_prob"])
        out.update({f"{tag}_obs0": obs0, f"{tag}_actions": np.stack(A), f"{tag}_obs": np.stack(O), f"{tag}_reward": np.stack(R), f"{tag}_term": np.stack(TE),
                    f"{tag}_trunc": np.stack(TR), f"{tag}_prob_info": np.stack(PR), f"{tag}_prob_mask": np.stack(PM),
                    f"{tag}_rng_after": np.stack([pcg_words(x.unwrapped.np_random) for x in v.envs])})
        v.close()
    save("toytext_frozenlake_per_env_maps.npz", **out)


def make_blackjack():
    """Blackjack-v1 (toy_text/blackjack.py): gym.make_vec(id, 8, "sync") trajectories for the registered rules (sab) and for
    natural=True; observations are the batched Tuple (three int64 arrays), stored as (T, 3, N)."""
    for key, kw in (("sab", {}), ("natural", {"natural": True, "sab": False})):
        v = gym.make_vec("Blackjack-v1", num_envs=8, vectorization_mode="sync", **kw)
        o0, _ = v.reset(seed=21)
        v.action_space.seed(4)
        A, O, R, TE, TR = [], [], [], [], []
        for _ in range(400):
            a = v.action_space.sample()
            o, r, te, tr, _ = v.step(a)
            A.append(a), O.append(np.stack(o)), R.append(np.asarray(r, dtype=np.float64)), TE.append(te), TR.append(tr)
        save(f"toytext_blackjack_{key}.npz", obs0=np.stack(o0), actions=np.stack(A), obs=np.stack(O), reward=np.stack(R), term=np.stack(TE),
             trunc=np.stack(TR), rng_after=np.stack([pcg_words(x.unwrapped.np_random) for x in v.envs]),
             natural=np.bool_(kw.get("natural", False)), sab=np.bool_(kw.get("sab", True)))
        v.close()


def make_same_step_infos():
    """infos_<env>.npz: the info dict of SyncVectorEnv under SAME_STEP (sync_vector_env.py:302-319: final_obs / final_info, the
    finished sub-env's top-level entries are its RESET info) for FrozenLake-v1 / Taxi-v4, and a NEXT_STEP recording with a partial
    reset (options["reset_mask"]) issued while another sub-env is waiting for its autoreset step (:232-234)."""
    for key, env_id in (("frozenlake", "FrozenLake-v1"), ("taxi", "Taxi-v4")):
        n, T = 6, 250
        v = gym.make_vec(env_id, num_envs=n, vectorization_mode="sync", vector_kwargs=dict(autoreset_mode=AutoresetMode.SAME_STEP))
        o0, _ = v.reset(seed=13)
        v.action_space.seed(5)
        rec = {k: [] for k in ("actions", "obs", "reward", "term", "trunc", "prob", "prob_is_int", "prob_mask", "has_final", "final_mask", "final_obs",
                               "final_prob", "final_prob_mask", "action_mask", "final_action_mask", "final_action_mask_mask")}
        for _ in range(T):
            a = v.action_space.sample()
            o, r, te, tr, info = v.step(a)
            rec["actions"].append(a), rec["obs"].append(o), rec["reward"].append(r), rec["term"].append(te), rec["trunc"].append(tr)
            rec["prob"].append(np.asarray(info["prob"], dtype=np.float64)), rec["prob_is_int"].append(np.issubdtype(np.asarray(info["prob"]).dtype, np.integer))
            rec["prob_mask"].append(info["_prob"])
            has = "final_info" in info
            rec["has_final"].append(has)
            rec["final_mask"].append(info["_final_info"] if has else np.zeros(n, np.bool_))
            rec["final_obs"].append(np.array([-1 if (not has or x is None) else int(x) for x in (info["final_obs"] if has else [None] * n)], dtype=np.int64))
            rec["final_prob"].append(np.asarray(info["final_info"]["prob"], dtype=np.float64) if has else np.zeros(n))
            rec["final_prob_mask"].append(info["final_info"]["_prob"] if has else np.zeros(n, np.bool_))
            if "action_mask" in info:
                rec["action_mask"].append(np.stack([np.asarray(m) for m in info["action_mask"]]))
                fam = info["final_info"]["action_mask"] if has else np.zeros((n, 6), np.int8)
                rec["final_action_mask"].append(np.stack([np.asarray(m) for m in fam]))
                rec["final_action_mask_mask"].append(info["final_info"]["_action_mask"] if has else np.zeros(n, np.bool_))
        save(f"infos_same_step_{key}.npz", obs0=o0, **{k: np.stack(x) for k, x in rec.items() if x})
        v.close()
    # NEXT_STEP + partial reset while sub-env 1 is pending its autoreset: FrozenLake with a 5-step TimeLimit makes every sub-env
    # truncate at the same step; then only sub-envs {0, 2} are reset explicitly
    n = 4
    v = gym.make_vec("FrozenLake-v1", num_envs=n, vectorization_mode="sync", max_episode_steps=5)
    v.reset(seed=3)
    v.action_space.seed(9)
    rec = {k: [] for k in ("actions", "obs", "reward", "term", "trunc", "prob", "prob_is_int", "prob_mask")}
    mask = np.array([True, False, True, False])
    reset_at, reset_obs = [], []
    for t in range(40):
        a = v.action_space.sample()
        o, r, te, tr, info = v.step(a)
        rec["actions"].append(a), rec["obs"].append(o), rec["reward"].append(r), rec["term"].append(te), rec["trunc"].append(tr)
        rec["prob"].append(np.asarray(info["prob"], dtype=np.float64)), rec["prob_is_int"].append(np.issubdtype(np.asarray(info["prob"]).dtype, np.integer))
        rec["prob_mask"].append(info["_prob"])
        if (te | tr).any() and len(reset_at) < 3:  # someone is now pending: reset half of the batch by hand
            ro, _ = v.reset(options={"reset_mask": mask})
            reset_at.append(t), reset_obs.append(ro)
    save("infos_partial_reset_frozenlake.npz", reset_mask=mask, reset_at=np.array(reset_at), reset_obs=np.stack(reset_obs),
         **{k: np.stack(x) for k, x in rec.items()})
    v.close()


def make_wrappers():
    """The reference's stateful vector wrappers on its own SyncVectorEnv: raw batches (inputs) and wrapped outputs.

    wrappers_normobs_<env>.npz   NormalizeObservation (stateful_observation.py) incl. frozen statistics for the last steps
    wrappers_normrew_<env>.npz   NormalizeReward (stateful_reward.py), NEXT_STEP and SAME_STEP
    wrappers_clip.npz            ClipReward
    """
    from gymnasium.wrappers.vector import ClipReward, NormalizeObservation, NormalizeReward

    for key, env_id, T in (("cartpole", "CartPole-v1", 120), ("pendulum", "Pendulum-v1", 120)):
        raw = gym.make_vec(env_id, num_envs=16, vectorization_mode="sync")
        w = NormalizeObservation(gym.make_vec(env_id, num_envs=16, vectorization_mode="sync"))
        o_raw, _ = raw.reset(seed=3)
        o_w, _ = w.reset(seed=3)
        raw.action_space.seed(5)
        RAW, OUT_, FROZEN = [o_raw], [o_w], []
        for t in range(T):
            if t == T - 20:
                w.update_running_mean = False
            a = raw.action_space.sample()
            RAW.append(raw.step(a)[0]), OUT_.append(w.step(a)[0]), FROZEN.append(not w.update_running_mean)
        save(f"wrappers_normobs_{key}.npz", raw=np.stack(RAW), out=np.stack(OUT_), frozen=np.array(FROZEN), mean=w.obs_rms.mean, var=w.obs_rms.var,
             count=np.float64(w.obs_rms.count))
        raw.close(), w.close()
    for key, env_id, T, mode in (("cartpole", "CartPole-v1", 200, AutoresetMode.NEXT_STEP), ("cartpole_same", "CartPole-v1", 200, AutoresetMode.SAME_STEP),
                                 ("mountaincar_continuous", "MountainCarContinuous-v0", 150, AutoresetMode.NEXT_STEP)):
        kw = dict(vector_kwargs={"autoreset_mode": mode})
        raw = gym.make_vec(env_id, num_envs=16, vectorization_mode="sync", **kw)
        w = NormalizeReward(gym.make_vec(env_id, num_envs=16, vectorization_mode="sync", **kw), gamma=0.97)
        raw.reset(seed=9), w.reset(seed=9)
        raw.action_space.seed(1)
        R, TE, TR, OUT_ = [], [], [], []
        for t in range(T):
            a = raw.action_space.sample()
            _, r, te, tr, _ = raw.step(a)
            R.append(np.asarray(r, dtype=np.float64)), TE.append(te), TR.append(tr), OUT_.append(w.step(a)[1])
        save(f"wrappers_normrew_{key}.npz", reward=np.stack(R), term=np.stack(TE), trunc=np.stack(TR), out=np.stack(OUT_),
             same_step=np.bool_(mode == AutoresetMode.SAME_STEP), gamma=np.float64(0.97), var=np.float64(w.return_rms.var),
             mean=np.float64(w.return_rms.mean), count=np.float64(w.return_rms.count), acc=w.accumulated_reward)
        raw.close(), w.close()
    raw = gym.make_vec("MountainCarContinuous-v0", num_envs=8, vectorization_mode="sync")
    w = ClipReward(gym.make_vec("MountainCarContinuous-v0", num_envs=8, vectorization_mode="sync"), -0.05, -0.01)
    raw.reset(seed=2), w.reset(seed=2)
    raw.action_space.seed(2)
    R, OUT_ = [], []
    for t in range(30):
        a = raw.action_space.sample()
        R.append(np.asarray(raw.step(a)[1], dtype=np.float64)), OUT_.append(w.step(a)[1])
    save("wrappers_clip.npz", reward=np.stack(R), out=np.stack(OUT_), lo=np.float64(-0.05), hi=np.float64(-0.01))
    raw.close(), w.close()


def make_cartpole_vector_entry_point():
    """The reference's OWN vector environment of CartPole-v1 (cartpole.py:353-505: the id's vector_entry_point, i.e. what stock make_vec returns): one
    generator for all sub-environments, float32 rewards.  Segments: a seeded run; the same env re-seeded with custom reset bounds that persist for
    the autoresets; reset(seed=None) continuing the stream; a short TimeLimit; the Sutton-Barto reward (-0.0 for a surviving pole)."""
    out = {}

    def run(tag, env, T, aseed, **reset_kw):
        obs0, _ = env.reset(**reset_kw)
        env.action_space.seed(aseed)
        acts, obs, rew, te, tr = [], [], [], [], []
        for _ in range(T):
            a = env.action_space.sample()
            o, r, d, u, info = env.step(a)
            assert info == {} and r.dtype == np.float32
            acts.append(a.copy()), obs.append(o.copy()), rew.append(r.copy()), te.append(d.copy()), tr.append(u.copy())
        out.update({f"{tag}_reset_obs": obs0.copy(), f"{tag}_actions": np.stack(acts), f"{tag}_obs": np.stack(obs), f"{tag}_rewards": np.stack(rew),
                    f"{tag}_terminated": np.stack(te), f"{tag}_truncated": np.stack(tr), f"{tag}_rng_after": pcg_words(env.np_random)})

    env = gym.make_vec("CartPole-v1", num_envs=8, vectorization_mode="vector_entry_point")
    assert type(env).__name__ == "CartPoleVectorEnv"
    run("a", env, 400, 1, seed=123)
    run("b", env, 200, 2, seed=7, options={"low": -0.1, "high": 0.08})  # wider bounds: kept for the autoresets that follow
    run("c", env, 100, 3)  # no seed: the stream continues, the bounds go back to the defaults
    env.close()
    env = gym.make_vec("CartPole-v1", num_envs=300, vectorization_mode="vector_entry_point", max_episode_steps=17)  # more than one workgroup; truncations
    run("d", env, 60, 4, seed=2**40 + 5)
    env.close()
    env = gym.make_vec("CartPole-v1", num_envs=5, vectorization_mode="vector_entry_point", sutton_barto_reward=True)
    run("e", env, 150, 5, seed=0)
    out["e_reward_signbit"] = np.signbit(out["e_rewards"])
    env.close()
    save("cartpole_vector_entry_point.npz", **out)


if __name__ == "__main__":
    if "--per-env-maps-only" in sys.argv:
        make_frozenlake_per_env_maps()
        sys.exit(0)
    if "--cartpole-vector-only" in sys.argv:
        make_cartpole_vector_entry_point()
        sys.exit(0)
    if "--toytext-only" in sys.argv:
        make_toytext()
        sys.exit(0)
    if "--blackjack-only" in sys.argv:
        make_blackjack()
        sys.exit(0)
    if "--episode-stats-only" in sys.argv:
        make_episode_stats()
        sys.exit(0)
    if "--wrappers-only" in sys.argv:
        make_wrappers()
        sys.exit(0)
    if "--teacher-wide-only" in sys.argv:
        make_teacher_wide()
        sys.exit(0)
    if "--infos-only" in sys.argv:
        make_same_step_infos()
        sys.exit(0)
    print("reference gymnasium", gym.__version__, "numpy", np.__version__)
    make_rng()
    make_rollouts()
    make_config1()
    make_modes()
    make_options()
    make_episode_stats()
    make_teacher()
    make_teacher_wide()
    make_action_samples()
    make_toytext()
    make_wrappers()
    make_blackjack()
    make_same_step_infos()
    make_cartpole_vector_entry_point()
    make_frozenlake_per_env_maps()
