"""-m gpu: the HIP MuJoCo-family engine (mjx_core.h / mjx_kernels.h through the C ABI) against the CPU oracle.

The physics itself is PARITY-UNPINNED against `mujoco`; what is checked here is that the product's
HIP implementation and the independent C restatement agree:
  * reset: observations and generator states bit-exact (NumPy uniform + ziggurat normal streams on device),
  * stepping: observations / rewards within the stated tolerance over windows that are re-synchronised to the oracle state
    (contact dynamics amplify the 1e-10 solver tolerance and libm differences, like Acrobot does), flags exact,
  * reward identities and info columns as in tests/envs/mujoco/test_mujoco_v5.py:116-152,222-254.
"""
import numpy as np
import pytest

import gymnasium_amd

pytestmark = pytest.mark.gpu
IDS = {"half_cheetah": "HalfCheetah-v5", "ant": "Ant-v5", "humanoid": "Humanoid-v5"}
# more robots on the same physics core (one-lane kernel), their own glue
MORE = {"hopper": "Hopper-v5", "walker2d": "Walker2d-v5", "inverted_pendulum": "InvertedPendulum-v5",
        "inverted_double_pendulum": "InvertedDoublePendulum-v5", "reacher": "Reacher-v5", "humanoid_standup": "HumanoidStandup-v5", "swimmer": "Swimmer-v5", "pusher": "Pusher-v5"}
ALL = {**IDS, **MORE}
NSTATE = {"half_cheetah": 17, "ant": 27, "humanoid": 45, "hopper": 11, "walker2d": 17, "inverted_pendulum": 4, "inverted_double_pendulum": 0, "reacher": 0, "humanoid_standup": 45, "swimmer": 8, "pusher": 14}
FIRST_INFO = {"pusher": ("reward_dist", "reward_ctrl", "reward_near"), "humanoid_standup": ("x_position", "reward_linup", "reward_quadctrl", "reward_impact"), "reacher": ("reward_dist", "reward_ctrl"), "inverted_pendulum": ("reward_survive",), "inverted_double_pendulum": ("reward_survive", "distance_penalty", "velocity_penalty")}


@pytest.mark.parametrize("name", list(ALL))
def test_reset_bit_exact_and_windowed_parity(name, oracle_factory):
    n, window, T = (256, 10, 60) if not name.startswith("humanoid") else (128, 5, 40)
    gpu = gymnasium_amd.make_vec(ALL[name], num_envs=n)
    cpu = gymnasium_amd.make_vec(ALL[name], num_envs=n, _engine_factory=oracle_factory)
    og, _ = gpu.reset(seed=11)
    oc, _ = cpu.reset(seed=11)
    nstate = NSTATE[name]  # qpos / qvel part: pure NumPy-stream arithmetic (the double pendulum shows sin / cos of it: tolerance)
    assert og.dtype == np.float64 and np.array_equal(og[:, :nstate], oc[:, :nstate]), "reset state must be bit-exact"
    # the humanoid's reset observation also shows cinert / cvel of the forward pass at the reset state (computed quantities)
    np.testing.assert_allclose(og[:, nstate:], oc[:, nstate:], rtol=1e-9, atol=1e-9)
    assert np.array_equal(gpu.get_rng_state(), cpu.get_rng_state())
    gpu.action_space.seed(2)
    worst, worst_r, mism = 0.0, 0.0, 0
    for t in range(T):
        a = gpu.action_space.sample()
        og, rg, teg, trg, ig = gpu.step(a)
        oc, rc, tec, trc, ic = cpu.step(a)
        mism += int((teg != tec).sum() + (trg != trc).sum())
        worst, worst_r = max(worst, float(np.abs(og - oc).max())), max(worst_r, float(np.abs(rg - rc).max()))
        # asserted at the MEASURED agreement (round 4, one MI355X: worst window 4.8e-9 on observations -- HalfCheetah, 10 steps --, <= 6e-10 for the
        # other ten robots; rewards <= 1.6e-10), so that a regression of the solver to "1e-6 per window" fails instead of hiding under 1e-5
        np.testing.assert_allclose(og, oc, rtol=0, atol=1e-8, err_msg=f"{name} obs t={t}")
        np.testing.assert_allclose(rg, rc, rtol=0, atol=1e-8, err_msg=f"{name} reward t={t}")
        assert set(ig) == set(ic), f"{name} t={t}: info keys differ"  # a key no sub-env supplied this step does not appear at all
        for k in FIRST_INFO.get(name, ("x_position", "x_velocity", "reward_forward", "reward_ctrl")):
            if k not in ic:
                continue
            np.testing.assert_allclose(ig[k], ic[k], rtol=0, atol=1e-7, err_msg=k)  # (x_velocity = a position difference / dt: 1e-8 / 0.008)
            assert np.array_equal(ig["_" + k], ic["_" + k])
        if (t + 1) % window == 0:
            st, el, fl = cpu.get_state()
            gpu.set_state(st, el, fl)
    assert mism == 0
    assert np.array_equal(gpu.get_rng_state(), cpu.get_rng_state())
    sg, sc = gpu.statistics(), cpu.statistics()
    assert all(sg[k] == sc[k] for k in ("env_steps", "reset_steps", "episodes", "length_sum"))
    print(f"{name}: max |obs diff| {worst:.3e}, max |reward diff| {worst_r:.3e} over {T} steps x {n} envs (resync every {window})")
    gpu.close(), cpu.close()


@pytest.mark.parametrize("name", list(IDS))
def test_free_running_divergence_report(name, oracle_factory):
    """Free-running (no resync) for 100 steps: reports how fast the two implementations drift apart (contact dynamics are
    chaotic: 1e-14 after one step grows to 1e-3 within ~50-100 steps); asserts the first 10 steps only."""
    n, T = 128, 100
    kw = dict(terminate_when_unhealthy=False) if name != "half_cheetah" else {}
    gpu = gymnasium_amd.make_vec(IDS[name], num_envs=n, **kw)
    cpu = gymnasium_amd.make_vec(IDS[name], num_envs=n, _engine_factory=oracle_factory, **kw)
    gpu.reset(seed=5), cpu.reset(seed=5)
    gpu.action_space.seed(0)
    trace = []
    for t in range(T):
        a = gpu.action_space.sample()
        og = gpu.step(a)[0]
        oc = cpu.step(a)[0]
        trace.append(float(np.abs(og - oc).max()))
    print(f"{name} free-running max |obs diff| at t=1,10,25,50,100: " + ", ".join(f"{trace[k - 1]:.2e}" for k in (1, 10, 25, 50, 100)))
    assert max(trace[:10]) < 1e-9  # measured: 3.0e-11 (HalfCheetah), 2.8e-11 (Ant), 6.0e-10 (Humanoid)
    gpu.close(), cpu.close()


@pytest.mark.parametrize("name,n", [("half_cheetah", 4096), ("ant", 4096), ("humanoid", 2048)])
def test_free_running_return_statistics(name, n, oracle_factory):
    """50 free-running steps (autoresets included), same seeds and actions on the HIP engine and the oracle.  Individual trajectories part ways
    chaotically, so this compares what a learner sees: the per-environment 50-step returns as a distribution.  The two samples are PAIRED (same
    noise, same actions), so the mean of the differences is tested against ITS standard error, and the two variances against each other; the
    numbers of finished episodes must agree within the Poisson noise of the ones that ended differently.  A solver regression that stays
    below the per-window tolerances but biases the dynamics (wrong friction cone, truncated iterations) shows up here."""
    T = 50
    gpu = gymnasium_amd.make_vec(IDS[name], num_envs=n)
    cpu = gymnasium_amd.make_vec(IDS[name], num_envs=n, _engine_factory=oracle_factory)
    gpu.reset(seed=77), cpu.reset(seed=77)
    gpu.action_space.seed(5)
    rg, rc = np.zeros(n), np.zeros(n)
    dg = dc = 0
    for t in range(T):
        a = gpu.action_space.sample()
        sg, sc = gpu.step(a), cpu.step(a)
        rg += sg[1]
        rc += sc[1]
        dg, dc = dg + int((sg[2] | sg[3]).sum()), dc + int((sc[2] | sc[3]).sum())
    diff = rg - rc
    se = diff.std(ddof=1) / np.sqrt(n)
    print(f"{name}: 50-step return mean {rg.mean():.4f} (HIP) vs {rc.mean():.4f} (oracle), paired diff {diff.mean():+.2e} +- {se:.2e}; std {rg.std():.4f} vs {rc.std():.4f}; "
          f"finished episodes {dg} vs {dc}; envs with |diff| > 1e-6: {int((np.abs(diff) > 1e-6).sum())}")
    assert abs(diff.mean()) <= 4 * se + 1e-9 * max(1.0, abs(rc.mean()))
    assert abs(rg.std() / rc.std() - 1.0) < 0.03
    assert abs(dg - dc) <= 4 * np.sqrt(max(dg, dc, 1)) + 1
    gpu.close(), cpu.close()


@pytest.mark.parametrize("name", ["ant", "humanoid", "walker2d", "half_cheetah"])
def test_teacher_forced_fallen_poses(name, oracle_factory):
    """Many-contact states the rollouts from reset reach late or never: every robot dropped in a random orientation just above the floor with random
    joint angles and velocities (the root orientation uniformly random for the free-joint robots: backs, heads, arms and elbows on the ground,
    capsule - capsule self-collision for the Humanoid), then 6 steps of HIP engine vs oracle from identical states.  This is where the unpinned
    parts live -- free-joint RK4 with many contacts, PGS with 10 - 25 active rows -- so at least the three implementations must agree there."""
    n = 384
    kw = {} if name == "half_cheetah" else dict(terminate_when_unhealthy=False)
    gpu = gymnasium_amd.make_vec(ALL[name], num_envs=n, **kw)
    cpu = gymnasium_amd.make_vec(ALL[name], num_envs=n, _engine_factory=oracle_factory, **kw)
    gpu.reset(seed=4), cpu.reset(seed=4)
    st, el, fl = cpu.get_state()
    rng = np.random.default_rng(12)
    st = st.copy()
    nq = {"ant": 15, "humanoid": 24, "walker2d": 9, "half_cheetah": 9}[name]
    nv = nq - 1 if name in ("ant", "humanoid") else nq
    if name in ("ant", "humanoid"):
        q = rng.normal(size=(n, 4))
        st[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)          # uniformly random root orientation
        st[:, 2] = rng.uniform(0.15, 0.45, n) if name == "ant" else rng.uniform(0.05, 0.3, n)  # torso just above the floor
        st[:, 7:nq] += rng.uniform(-0.5, 0.5, (n, nq - 7))
    else:
        st[:, 1] = rng.uniform(-1.0, -0.2, n) if name == "walker2d" else rng.uniform(-0.45, -0.1, n)  # root z slider: down towards the floor
        st[:, 2] = rng.uniform(-1.5, 1.5, n)                                # pitched over
        st[:, 3:nq] += rng.uniform(-0.6, 0.6, (n, nq - 3))
    st[:, nq:nq + nv] = rng.uniform(-1.0, 1.0, (n, nv))
    st[:, nq + nv:nq + 2 * nv] = 0.0
    gpu.set_state(st, el, fl), cpu.set_state(st, el, fl)
    gpu.action_space.seed(6)
    worst = worst_r = 0.0
    for t in range(6):
        a = gpu.action_space.sample()
        og, rg, teg, trg, _ = gpu.step(a)
        oc, rc, tec, trc, _ = cpu.step(a)
        assert np.isfinite(oc).all() and np.array_equal(teg, tec)
        worst, worst_r = max(worst, float((np.abs(og - oc) / (1.0 + np.abs(oc))).max())), max(worst_r, float((np.abs(rg - rc) / (1.0 + np.abs(rc))).max()))
    contacts = float((np.abs(oc[:, -6 * (14 if name == "ant" else 13):]) > 0).any(axis=1).mean()) if name in ("ant", "humanoid") else float("nan")
    print(f"{name} fallen poses: max |obs diff| / (1 + |obs|) {worst:.3e}, rewards likewise {worst_r:.3e} over 6 steps x {n} envs; share of envs with contact forces {contacts:.2f}")
    assert worst < 1e-8 and worst_r < 1e-9  # measured (round 4): 2.5e-13 (Ant) ... 4.6e-11 (Humanoid, HalfCheetah); rewards <= 8e-13
    gpu.close(), cpu.close()


@pytest.mark.parametrize("name", list(ALL))
def test_fused_rollout_equals_stepping(name):
    import torch

    n, T = (128, 12) if not name.startswith("humanoid") else (64, 6)
    a = gymnasium_amd.make_vec(ALL[name], num_envs=n, output="torch")
    b = gymnasium_amd.make_vec(ALL[name], num_envs=n, output="torch")
    a.reset(seed=3), b.reset(seed=3)
    a.action_space.seed(7), b.action_space.seed(7)
    out = a.rollout(T)
    for t in range(T):
        act = b.action_space.sample()
        o, r, te, tr, _ = b.step(torch.from_numpy(act).cuda())
        assert np.array_equal(out["actions"][t].cpu().numpy(), act), f"sampled actions t={t}"
        assert torch.equal(out["obs"][t], o) and torch.equal(out["rewards"][t], r) and torch.equal(out["terminations"][t], te)
    assert np.array_equal(a.action_space.sample(), b.action_space.sample())
    assert np.array_equal(a.get_rng_state(), b.get_rng_state())
    a.close(), b.close()


@pytest.mark.parametrize("env_id", ["Ant-v5", "Humanoid-v5"])
def test_full_size_properties(env_id):
    """BASELINE.json configs[3] / configs[4] (per GPU): Ant-v5 / Humanoid-v5, num_envs = 32768: accounting identities, determinism, shard
    invariance (the second half of the batch as its own engine with env_index_offset reproduces the same rows)."""
    import torch

    N, T = 32768, (4 if env_id == "Ant-v5" else 2)
    a = gymnasium_amd.make_vec(env_id, num_envs=N, output="torch")
    a.reset(seed=0)
    a.action_space.seed(0)
    out = a.rollout(T)
    st = a.statistics()
    assert st["env_steps"] + st["reset_steps"] == N * T and torch.isfinite(out["obs"]).all()
    h = N // 2
    c = gymnasium_amd.make_vec(env_id, num_envs=h, output="torch", env_index_offset=h)
    c.reset(seed=0)
    out_c = c.rollout(T, actions=out["actions"][:, h:].contiguous())
    assert torch.equal(out["obs"][:, h:], out_c["obs"]) and torch.equal(out["rewards"][:, h:], out_c["rewards"])
    a.close(), c.close()


COOP_ROBOTS = {**IDS, "hopper": "Hopper-v5", "walker2d": "Walker2d-v5"}  # (round 2: the planar walkers run on the cooperative kernel too)


@pytest.mark.parametrize("name", list(COOP_ROBOTS))
def test_cooperative_kernel_equals_one_lane_simulator(name, monkeypatch):
    """The two HIP implementations of the physics -- mjx_coop.h (G lanes per env, the default) and mjx_core.h (one lane per
    env, MI355ENV_MJ_SERIAL=1) -- agree over re-synchronised windows; flags and RNG consumption are identical."""
    n, T, window = (128, 30, 5) if name != "humanoid" else (64, 20, 5)
    monkeypatch.setenv("MI355ENV_MJ_SERIAL", "1")
    ser = gymnasium_amd.make_vec(COOP_ROBOTS[name], num_envs=n)
    monkeypatch.delenv("MI355ENV_MJ_SERIAL")
    monkeypatch.setenv("MI355ENV_MJ_COOP", "1")
    coop = gymnasium_amd.make_vec(COOP_ROBOTS[name], num_envs=n)
    monkeypatch.delenv("MI355ENV_MJ_COOP")
    o1, _ = ser.reset(seed=21)
    o2, _ = coop.reset(seed=21)
    assert np.array_equal(o1, o2)
    ser.action_space.seed(4)
    worst = 0.0
    for t in range(T):
        a = ser.action_space.sample()
        o1, r1, te1, tr1, _ = ser.step(a)
        o2, r2, te2, tr2, _ = coop.step(a)
        assert np.array_equal(te1, te2) and np.array_equal(tr1, tr2)
        worst = max(worst, float(np.abs(o1 - o2).max()))
        np.testing.assert_allclose(o2, o1, rtol=0, atol=1e-8, err_msg=f"{name} obs t={t}")  # measured <= 9.2e-10 (HalfCheetah)
        np.testing.assert_allclose(r2, r1, rtol=0, atol=1e-8, err_msg=f"{name} reward t={t}")
        if (t + 1) % window == 0:
            st, el, fl = ser.get_state()
            coop.set_state(st, el, fl)
    assert np.array_equal(ser.get_rng_state(), coop.get_rng_state())
    print(f"{name}: cooperative vs one-lane max |obs diff| {worst:.3e} (resync every {window})")
    ser.close(), coop.close()


@pytest.mark.parametrize("name,kw", [("ant", {}), ("humanoid", {}), ("humanoid", {"terminate_when_unhealthy": False})], ids=["ant", "humanoid", "humanoid-on-the-ground"])
def test_full_batch_cooperative_equals_one_lane_at_the_benchmark_shape(name, kw, monkeypatch):
    """the WHOLE batch of BASELINE configs[3] / [4]'s per-GPU shape -- 32 768 robots -- stepped by the cooperative kernel and by the one-lane
    simulator from the same states, window by window (4 steps, like a bench launch), every robot compared; with termination off and a warm-up the Humanoids
    lie on the ground (4-15 contacts, PGS at its 50-sweep cap).  The oracle covers 1 024 strided robots of the same shape inside bench.py
    (mujoco_window_check); this is every robot, against the second HIP implementation."""
    import torch

    n, windows, T = 32768, 3, 4
    warm = 40 if kw else 6
    monkeypatch.setenv("MI355ENV_MJ_COOP", "1")
    coop = gymnasium_amd.make_vec(COOP_ROBOTS[name], num_envs=n, output="torch", **kw)
    monkeypatch.delenv("MI355ENV_MJ_COOP")
    monkeypatch.setenv("MI355ENV_MJ_SERIAL", "1")
    ser = gymnasium_amd.make_vec(COOP_ROBOTS[name], num_envs=n, output="torch", **kw)
    monkeypatch.delenv("MI355ENV_MJ_SERIAL")
    coop.reset(seed=0), ser.reset(seed=0)
    coop.action_space.seed(0)
    for _ in range(warm):
        coop.rollout(T, return_actions=False)
    worst = 0.0
    for w in range(windows):
        st, el, fl = coop.get_state()
        ser.set_state(st, el, fl)
        ser._engine.seed(coop.get_rng_state(), None)
        out = coop.rollout(T)
        ref = ser.rollout(T, actions=out["actions"])
        assert torch.equal(out["terminations"], ref["terminations"]) and torch.equal(out["truncations"], ref["truncations"]), (name, w)
        d = max(float((out["obs"] - ref["obs"]).abs().max()), float((out["rewards"] - ref["rewards"]).abs().max()))
        worst = max(worst, d)
        assert d <= 1e-8, (name, w, d)
    print(f"{name} {kw}: 32768 robots, cooperative vs one-lane max diff {worst:.3e} over {windows} windows of {T} steps after {warm} warm-up launches")
    coop.close(), ser.close()


def test_pusher_contacts_gpu_vs_oracle(oracle_factory):
    """Pusher-v5 with the arm lowered onto the object and the table (capsule - cylinder and plane - capsule contacts active from the
    first step): HIP vs oracle from identical states, 10 steps."""
    n = 256
    gpu = gymnasium_amd.make_vec("Pusher-v5", num_envs=n)
    cpu = gymnasium_amd.make_vec("Pusher-v5", num_envs=n, _engine_factory=oracle_factory)
    gpu.reset(seed=1), cpu.reset(seed=1)
    st, el, fl = cpu.get_state()
    rng = np.random.default_rng(0)
    st = st.copy()
    st[:, :] = 0.0
    st[:, 1] = rng.uniform(0.45, 0.62, n)        # shoulder lift: the wrist comes down to the object's / the table's height
    st[:, 3] = rng.uniform(-0.45, -0.05, n)      # elbow
    st[:, 6] = rng.uniform(-0.5, 0.5, n)         # wrist roll
    # the object just in front of the wrist (documented start pose of the tips: x = 0.821, y = -0.6): sliders are (y, x) offsets from (0.45, -0.05)
    st[:, 7] = -0.6 + rng.uniform(-0.12, 0.12, n) - (-0.05)
    st[:, 8] = 0.80 + rng.uniform(-0.08, 0.06, n) - 0.45
    st[:, 11:18] = rng.uniform(-0.5, 0.5, (n, 7))  # arm joint velocities
    gpu.set_state(st, el, fl), cpu.set_state(st, el, fl)
    gpu.action_space.seed(3)
    moved = np.zeros(n, dtype=bool)
    worst = 0.0
    for t in range(10):
        a = gpu.action_space.sample()
        og, rg, _, _, ig = gpu.step(a)
        oc, rc, _, _, ic = cpu.step(a)
        worst = max(worst, float(np.abs(og - oc).max()))
        np.testing.assert_allclose(og, oc, rtol=0, atol=1e-9, err_msg=f"t={t}")  # measured 3.5e-11
        np.testing.assert_allclose(rg, rc, rtol=0, atol=1e-9)
        moved |= np.abs(oc[:, 17] - (0.45 + st[:, 8])) > 1e-6
    assert moved.sum() > n // 8, "the arm must actually push the object in a good share of the environments"
    print(f"pusher contacts: max |obs diff| {worst:.3e} over 10 steps x {n} envs; object pushed in {int(moved.sum())} envs")
    gpu.close(), cpu.close()


@pytest.mark.parametrize("env_id", ["Humanoid-v5", "HumanoidStandup-v5"])
def test_humanoid_solver_choice(env_id, oracle_factory):
    """The humanoids run their MJCF's solver by default -- `solver="PGS" iterations="50"` (humanoid.xml:8), restated in mjx_coop.h pgs() --
    and the converged Newton solver only on request (solver="Newton").  Each agrees with the oracle's solver of the same name; the two differ
    from each other by what a truncated PGS leaves (~1e-5 relative in qacc per forward pass)."""
    n = 96
    envs = {(dev, s): gymnasium_amd.make_vec(env_id, num_envs=n, solver=s, **({} if dev == "gpu" else dict(_engine_factory=oracle_factory)))
            for dev in ("gpu", "cpu") for s in ("PGS", "Newton")}
    for e in envs.values():
        e.reset(seed=8)
    envs["gpu", "PGS"].action_space.seed(1)
    gap = 0.0
    for t in range(8):
        a = envs["gpu", "PGS"].action_space.sample()
        out = {k: e.step(a) for k, e in envs.items()}
        for s in ("PGS", "Newton"):
            np.testing.assert_allclose(out["gpu", s][0], out["cpu", s][0], rtol=0, atol=1e-7, err_msg=f"{env_id} {s} obs t={t}")
            np.testing.assert_allclose(out["gpu", s][1], out["cpu", s][1], rtol=1e-9, atol=1e-7, err_msg=f"{env_id} {s} reward t={t}")
            assert np.array_equal(out["gpu", s][2], out["cpu", s][2])
        gap = max(gap, float(np.abs(out["gpu", "PGS"][0] - out["gpu", "Newton"][0]).max()))
    assert gap > 0.0, "PGS / 50 and converged Newton must not be the same computation"
    print(f"{env_id}: max |obs(PGS) - obs(Newton)| over 8 steps = {gap:.3e}")
    with pytest.raises(Exception):
        gymnasium_amd.make_vec(env_id, num_envs=1, solver="CG")
    for e in envs.values():
        e.close()


@pytest.mark.parametrize("mode", ["NextStep", "SameStep"])
def test_device_resident_infos_equal_the_numpy_infos(mode):
    """output="torch": the MuJoCo kinds' infos are device tensors assembled without a read-back (no D2H, no synchronisation per step);
    same values as the NumPy dict of a twin env (tests/test_device_infos.py runs the same comparison on the checker backend)."""
    from test_device_infos import compare_device_infos

    compare_device_infos("Ant-v5", None, mode, steps=45)
    compare_device_infos("Hopper-v5", None, mode, steps=80)


def _kwargs_cases():
    from mujoco_kwargs_cases import CASES

    return [(env_id, k) for env_id, cases in CASES.items() for k in range(len(cases))]


@pytest.mark.parametrize("env_id,k", _kwargs_cases())
def test_non_default_constructor_kwargs_vs_oracle(env_id, k, oracle_factory):
    """The constructor keywords of the v5 classes at non-default values (tests/mujoco_kwargs_cases.py; the CPU suite pins the same cases on the reference's env
    classes): HIP engine vs oracle -- observation shapes, reset bit for bit, 12 steps within the windowed tolerance, flags and info columns."""
    from mujoco_kwargs_cases import CASES

    kw = CASES[env_id][k]
    n = 64
    gpu = gymnasium_amd.make_vec(env_id, num_envs=n, max_episode_steps=20, **kw)
    cpu = gymnasium_amd.make_vec(env_id, num_envs=n, max_episode_steps=20, _engine_factory=oracle_factory, **kw)
    assert gpu.single_observation_space == cpu.single_observation_space
    og, oc = gpu.reset(seed=8)[0], cpu.reset(seed=8)[0]
    np.testing.assert_allclose(og, oc, rtol=1e-9, atol=1e-9)
    gpu.action_space.seed(4)
    for t in range(12):
        a = gpu.action_space.sample()
        g, c = gpu.step(a), cpu.step(a)
        np.testing.assert_allclose(g[0], c[0], rtol=0, atol=2e-8, err_msg=f"obs t={t}")
        np.testing.assert_allclose(g[1], c[1], rtol=0, atol=2e-8, err_msg=f"reward t={t}")
        assert np.array_equal(g[2], c[2]) and np.array_equal(g[3], c[3]), t
        assert sorted(g[4]) == sorted(c[4])
        for key in g[4]:
            if not key.startswith("_") and isinstance(g[4][key], np.ndarray) and g[4][key].dtype.kind == "f":
                np.testing.assert_allclose(g[4][key], c[4][key], rtol=0, atol=2e-8, err_msg=f"info {key} t={t}")
        if t % 4 == 3:  # re-synchronise (contact dynamics amplify last-bit differences, like the windowed parity test does)
            gpu.set_state(*cpu.get_state())
    gpu.close(), cpu.close()
