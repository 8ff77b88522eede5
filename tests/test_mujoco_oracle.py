"""CPU tests of the MuJoCo-pipeline oracle (oracle/mujoco_core.c, oracle/mujoco_envs.c) and the model compiler.

PARITY UNPINNED for the physics: `mujoco` (third-party, not in the reference tree, not installed here) cannot be run, and
the reference holds no numeric MuJoCo trajectory.  What these tests pin instead:
  * the structural facts the reference's own tests pin (tests/envs/mujoco/test_mujoco_v5.py:503-558 model counts; :429-451
    observation structure; :222-254 reward == sum of its terms; :116-152 info velocity == finite difference; :693-710
    reset determinism / zero-noise reset),
  * everything that IS NumPy: the reset-noise streams (uniform + ziggurat standard_normal) and np.sum's pairwise order,
    bit for bit,
  * physics invariants an incorrect restatement would break: CRB mass matrix == sum_b m Jv'Jv + Jw'IJw, RNE bias ==
    Lagrangian finite differences, energy conservation in free flight, contact force == weight at rest, solver KKT.
"""
import ctypes

import numpy as np
import pytest

import gymnasium_amd
from gymnasium_amd.envs.mujoco import compiler as cp
from oracle import mujoco as omj

SPECS = {  # nq, nv, nu, nbody, ngeom, obs, frame_skip*timestep       (test_mujoco_v5.py:503-558 and the env docstrings)
    "half_cheetah": (9, 9, 6, 8, 9, 17, 0.05), "ant": (15, 14, 8, 14, 14, 105, 0.05), "humanoid": (24, 23, 17, 14, 18, 348, 0.015)}
IDS = {"half_cheetah": "HalfCheetah-v5", "ant": "Ant-v5", "humanoid": "Humanoid-v5"}


@pytest.fixture(scope="module")
def models():
    return {k: omj.OracleModel(k) for k in SPECS}


def rand_state(m, rng, scale=0.5):
    q = m.qpos0.copy()
    for j in range(m.njnt):
        a = m.jnt_qposadr[j]
        if m.jnt_type[j] == cp.FREE:
            q[a:a + 3] += rng.normal(size=3) * 0.3
            qq = rng.normal(size=4)
            q[a + 3:a + 7] = qq / np.linalg.norm(qq)
        else:
            q[a] += rng.uniform(-scale, scale)
    return q, rng.normal(size=m.nv)


def strip(m):
    """No dissipation, no constraints: joints free to swing, floor gone."""
    m.dof_damping[:] = 0
    m.jnt_limited[:] = 0
    m.jnt_stiffness[:] = 0
    for k in ("pair_geom1", "pair_geom2", "pair_condim", "pair_friction", "pair_margin", "pair_solref", "pair_solimp"):
        setattr(m, k, getattr(m, k)[:0])
    return m


@pytest.mark.parametrize("name", list(SPECS))
def test_model_counts_and_masses(name, models):
    m = models[name].m
    nq, nv, nu, nbody, ngeom, _, _ = SPECS[name]
    assert (m.nq, m.nv, m.nu, m.nbody, m.ngeom) == (nq, nv, nu, nbody, ngeom)
    if name == "half_cheetah":  # settotalmass="14"
        assert abs(m.total_mass - 14.0) < 1e-12
        np.testing.assert_allclose(m.body_mass[1:], [6.25020921, 1.54351464, 1.5874477, 1.09539749, 1.43807531, 1.20083682, 0.88451883], rtol=2e-8)
    if name == "ant":  # torso sphere r=0.25 at density 5
        assert abs(m.body_mass[1] - 5.0 * 4 / 3 * np.pi * 0.25 ** 3) < 1e-15
        # actuator order is not joint order (ant.xml:72-79): hip_4, ankle_4, hip_1, ...
        assert list(m.actuator_dofadr) == [12, 13, 6, 7, 8, 9, 10, 11]
    if name == "humanoid":  # abdomen_y before abdomen_z (humanoid.xml:103-104)
        assert list(m.actuator_dofadr[:3]) == [7, 6, 8]


@pytest.mark.parametrize("name", list(SPECS))
def test_kinematics_and_mass_matrix_vs_jacobian_formulation(name, models):
    om = models[name]
    m, d, rng = om.m, om.make_data(), np.random.default_rng(0)
    for _ in range(10):
        q, v = rand_state(m, rng)
        d.set_state(q, v, np.zeros(m.nu))
        d.forward()
        kin = cp.kinematics(m, q)
        np.testing.assert_allclose(d.get("xpos"), kin["xpos"], atol=1e-14)
        np.testing.assert_allclose(d.get("xmat").reshape(-1, 3, 3), kin["xmat"], atol=1e-14)
        np.testing.assert_allclose(d.get("xipos"), kin["xipos"], atol=1e-14)
        M = cp.mass_matrix(m, kin)
        np.testing.assert_allclose(d.get("qM"), M, rtol=1e-12, atol=1e-13 * np.abs(M).max())


def test_bias_forces_vs_lagrangian_finite_differences(models):
    om = models["half_cheetah"]  # hinge / slide coordinates only: d/dq is plain
    m, d, rng = om.m, om.make_data(), np.random.default_rng(1)

    def Mq(q):
        return cp.mass_matrix(m, cp.kinematics(m, q))

    def V(q):
        kin = cp.kinematics(m, q)
        return -sum(m.body_mass[b] * m.gravity @ kin["xipos"][b] for b in range(m.nbody))

    for _ in range(3):
        q, v = rand_state(m, rng)
        q[1] += 3.0
        d.set_state(q, v, np.zeros(m.nu))
        d.forward()
        eps, nv = 1e-6, m.nv
        dM, dV = np.zeros((nv, nv, nv)), np.zeros(nv)
        for k in range(nv):
            e = np.zeros(nv)
            e[k] = eps
            dM[:, :, k] = (Mq(q + e) - Mq(q - e)) / (2 * eps)
            dV[k] = (V(q + e) - V(q - e)) / (2 * eps)
        bias = np.einsum("ijk,j,k->i", dM, v, v) - 0.5 * np.einsum("jki,j,k->i", dM, v, v) + dV
        np.testing.assert_allclose(d.get("qfrc_bias"), bias, rtol=1e-6, atol=1e-7 * np.abs(bias).max())


@pytest.mark.parametrize("name,tol", [("half_cheetah", 1e-9), ("ant", 1e-6), ("humanoid", 1e-3)])
def test_energy_conservation_in_free_flight(name, tol):
    m = strip(cp.compile_model(name))
    m.integrator, m.timestep = "RK4", 0.002
    om = omj.OracleModel(m)
    d, rng = om.make_data(), np.random.default_rng(2)
    q, v = rand_state(m, rng)
    d.set_state(q, 2 * v, np.zeros(m.nu))

    def energy():
        d.forward()
        M, vv, xi = d.get("qM"), d.get("qvel"), d.get("xipos")
        return 0.5 * vv @ M @ vv - sum(m.body_mass[b] * m.gravity @ xi[b] for b in range(m.nbody))

    e0 = energy()
    d.step(250)
    assert abs(energy() - e0) / abs(e0) < tol


@pytest.mark.parametrize("name", ["half_cheetah", "ant"])
def test_resting_contact_carries_the_weight_and_solver_kkt(name, models):
    om = models[name]
    m, d = om.m, om.make_data()
    d.reset()
    d.step(500)
    d.forward()
    d.rne_post_constraint()
    assert np.abs(d.get("qvel")).max() < 0.05 and d.get("ncon") >= 2
    fz = d.get("cfrc_ext")[:, 5].sum()
    assert abs(fz - m.total_mass * 9.81) / (m.total_mass * 9.81) < 2e-3
    M, J, f, qa, qs, aref, D = (d.get(k) for k in ("qM", "efc_J", "efc_force", "qacc", "qacc_smooth", "efc_aref", "efc_D"))
    assert np.abs(M @ (qa - qs) - J.T @ f).max() < 1e-8 * max(1.0, np.abs(J.T @ f).max())
    jar = J @ qa - aref
    np.testing.assert_allclose(f, np.where(jar < 0, -D * jar, 0.0), rtol=1e-9, atol=1e-9)
    assert (f >= 0).all()


def test_norm_matches_numpy():
    """np.linalg.norm of the 2- / 3-vectors the reference's glue takes (ant_v5.py:362 distance_from_origin, reacher_v5.py:190,
    pusher_v5.py:268-271, the re-draw loops of their resets): sqrt(x.dot(x)) with the BLAS dot's fused accumulation -- NOT the separately
    rounded sum of squares, which differs in the last bit for ~8 % of vectors (asserted, so that a NumPy / BLAS with another arithmetic shows up
    here and not as a 1-ulp reward difference)."""
    dll = omj.dll()
    dll.orc_test_np_norm.restype, dll.orc_test_np_norm.argtypes = ctypes.c_double, [ctypes.c_void_p, ctypes.c_int]
    rng = np.random.default_rng(11)
    plain_differs = 0
    for n in (2, 3):
        for _ in range(20000):
            v = rng.normal(size=n) * 10.0 ** rng.integers(-3, 3)
            ref = np.linalg.norm(v, ord=2)
            assert dll.orc_test_np_norm(v.ctypes.data, n) == ref == np.linalg.norm(v), (n, v)
            plain_differs += float(np.sqrt(np.sum(v * v))) != ref
    assert plain_differs > 100


def test_numpy_sum_order_and_standard_normal_bit_exact():
    dll = omj.dll()
    dll.orc_test_np_sum_f64.restype, dll.orc_test_np_sum_f64.argtypes = ctypes.c_double, [ctypes.c_void_p, ctypes.c_int]
    dll.orc_test_np_sum_f32.restype, dll.orc_test_np_sum_f32.argtypes = ctypes.c_float, [ctypes.c_void_p, ctypes.c_int]
    rng = np.random.default_rng(3)
    for n in (1, 5, 6, 7, 8, 9, 16, 17, 23, 83, 84, 128, 129, 130, 300):
        for _ in range(20):
            a = rng.normal(size=n) * 10.0 ** rng.integers(-3, 4)
            assert dll.orc_test_np_sum_f64(a.ctypes.data, n) == np.sum(a), n
            if n == 84:  # Ant / Humanoid contact cost: np.sum over the (14, 6) cfrc_ext array
                assert dll.orc_test_np_sum_f64(a.ctypes.data, n) == np.sum(np.square(np.sqrt(np.abs(a))).reshape(14, 6) * 0 + a.reshape(14, 6)), n
            if n <= 17:
                af = a.astype(np.float32)
                assert np.float32(dll.orc_test_np_sum_f32(af.ctypes.data, n)) == np.sum(af), n
    dll.orc_test_standard_normal.restype = None
    dll.orc_test_standard_normal.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    from gymnasium_amd import _native

    for seed in (0, 1, 42, 2 ** 40 + 7):
        g = np.random.Generator(np.random.PCG64(seed))
        words = _native.pcg_words(g)
        n = 200_000
        out, wout = np.zeros(n), np.zeros(4, dtype=np.uint64)
        dll.orc_test_standard_normal(words.ctypes.data, n, out.ctypes.data, wout.ctypes.data)
        assert np.array_equal(out, g.standard_normal(n)), seed
        assert np.array_equal(wout, _native.pcg_words(g)), "generator states diverged"


@pytest.mark.parametrize("name", list(SPECS))
def test_reset_noise_streams_match_numpy(name, oracle_factory, models):
    m = models[name].m
    env = gymnasium_amd.make_vec(IDS[name], num_envs=3, _engine_factory=oracle_factory, exclude_current_positions_from_observation=False)
    obs, info = env.reset(seed=100)
    assert obs.dtype == np.float64 and obs.shape[0] == 3
    for i in range(3):
        g = np.random.Generator(np.random.PCG64(np.random.SeedSequence(100 + i)))
        scale = 1e-2 if name == "humanoid" else 0.1
        qpos = m.qpos0 + g.uniform(low=-scale, high=scale, size=m.nq)
        qvel = g.uniform(low=-scale, high=scale, size=m.nv) if name == "humanoid" else scale * g.standard_normal(m.nv)
        assert np.array_equal(obs[i, :m.nq], qpos) and np.array_equal(obs[i, m.nq:m.nq + m.nv], qvel)
        assert np.array_equal(env.get_rng_state()[i], __import__("gymnasium_amd")._native.pcg_words(g))
    env.close()
    env = gymnasium_amd.make_vec(IDS[name], num_envs=2, _engine_factory=oracle_factory, reset_noise_scale=0)
    obs, _ = env.reset(seed=5)
    skip = 1 if name == "half_cheetah" else 2
    assert np.array_equal(obs[0, :m.nq - skip], m.qpos0[skip:]) and not obs[0, m.nq - skip:m.nq - skip + m.nv].any()
    env.close()


@pytest.mark.parametrize("name", list(SPECS))
def test_env_structure_rewards_and_determinism(name, oracle_factory):
    nq, nv, nu, nbody, ngeom, obs_dim, dt = SPECS[name]
    a = gymnasium_amd.make_vec(IDS[name], num_envs=4, _engine_factory=oracle_factory)
    b = gymnasium_amd.make_vec(IDS[name], num_envs=4, _engine_factory=oracle_factory)
    assert a.single_observation_space.shape == (obs_dim,) and a.single_observation_space.dtype == np.float64
    assert a.single_action_space.shape == (nu,) and a.single_action_space.dtype == np.float32
    assert sum(a.observation_structure[k] for k in a.observation_structure if k != "skipped_qpos") == obs_dim
    oa, _ = a.reset(seed=9)
    ob, _ = b.reset(seed=9)
    assert np.array_equal(oa, ob)
    a.action_space.seed(1)
    prev_x, prev_done = None, np.zeros(4, dtype=bool)
    for t in range(60):
        act = a.action_space.sample()
        oa, ra, tea, tra, ia = a.step(act)
        ob, rb, teb, trb, _ = b.step(act)
        assert np.array_equal(oa, ob) and np.array_equal(ra, rb) and np.array_equal(tea, teb)
        live = ~prev_done
        terms = ia["reward_forward"] + ia["reward_ctrl"] + (ia["reward_contact"] + ia["reward_survive"] if name != "half_cheetah" else 0.0)
        np.testing.assert_allclose(ra[live], terms[live], rtol=1e-12, atol=1e-12)       # test_mujoco_v5.py:222-254
        assert np.array_equal(ia["_x_velocity"], live) and ia["_x_position"].all()
        if name == "half_cheetah" and prev_x is not None:                                # test_mujoco_v5.py:116-152
            np.testing.assert_allclose(ia["x_velocity"][live], ((ia["x_position"] - prev_x) / dt)[live], rtol=1e-9, atol=1e-9)
        assert (ra[prev_done] == 0).all() and not tea[prev_done].any()
        prev_x, prev_done = ia["x_position"].copy(), tea | tra
        assert np.isfinite(oa).all()
    if name == "half_cheetah":
        assert not tea.any()
    a.close(), b.close()


def test_checkpoint_resume_is_exact(oracle_factory):
    a = gymnasium_amd.make_vec("Ant-v5", num_envs=3, _engine_factory=oracle_factory)
    b = gymnasium_amd.make_vec("Ant-v5", num_envs=3, _engine_factory=oracle_factory)
    a.reset(seed=3), b.reset(seed=4)
    a.action_space.seed(0)
    for _ in range(7):
        a.step(a.action_space.sample())
    b.set_state(*a.get_state())
    for _ in range(5):
        act = a.action_space.sample()
        ra, rb = a.step(act), b.step(act)
        live = ~(ra[2] | ra[3])
        assert np.array_equal(ra[1][live], rb[1][live]) and np.allclose(ra[0][live], rb[0][live], rtol=0, atol=1e-9)
    a.close(), b.close()


def test_humanoid_pgs50_residual_vs_converged_newton():
    """humanoid.xml:8 asks for solver="PGS" iterations="50"; oracle and engine solve the same convex problem to convergence
    with Newton instead (compiler.compile_model docstring).  This measures what that substitution changes: the PGS/50
    acceleration differs from the converged one at the 1e-3 level at a typical standing-contact state."""
    newton = omj.OracleModel(cp.compile_model("humanoid", faithful_solver=False))
    pgs = omj.OracleModel(cp.compile_model("humanoid", faithful_solver=True))
    assert newton.m.solver == "Newton" and pgs.m.solver == "PGS" and newton.m.reference_solver == "PGS"
    dn, dp = newton.make_data(), pgs.make_data()
    rng = np.random.default_rng(4)
    q = newton.m.qpos0.copy()
    q[2] = 1.28  # feet into the floor margin
    q[7:] += rng.uniform(-0.05, 0.05, size=newton.m.nq - 7)
    v = rng.normal(size=newton.m.nv) * 0.1
    for d in (dn, dp):
        d.set_state(q, v, np.zeros(newton.m.nu))
        d.forward()
    assert dn.get("nefc") == dp.get("nefc") > 0
    an, ap = dn.get("qacc"), dp.get("qacc")
    rel = np.abs(an - ap).max() / np.abs(an).max()
    print(f"PGS/50 vs converged Newton: max relative qacc difference {rel:.2e}, PGS sweeps {dp.get('solver_iter')}")
    assert rel < 5e-2


# ---- Hopper / Walker2d / InvertedPendulum / InvertedDoublePendulum on the same pipeline ----------------------
MORE_SPECS = {  # nq, nv, nu, nbody, njnt (test_mujoco_v5.py:526-580), obs, frame_skip * timestep, reset-noise scale, uniform velocity noise?
    "hopper": ("Hopper-v5", 6, 6, 3, 5, 6, 11, 0.008, 5e-3, True), "walker2d": ("Walker2d-v5", 9, 9, 6, 8, 9, 17, 0.008, 5e-3, True),
    "inverted_pendulum": ("InvertedPendulum-v5", 2, 2, 1, 3, 2, 4, 0.04, 0.01, True),
    "inverted_double_pendulum": ("InvertedDoublePendulum-v5", 3, 3, 1, 4, 3, 9, 0.05, 0.1, False),
    "reacher": ("Reacher-v5", 4, 4, 2, 5, 4, 10, 0.02, 0.1, True)}


@pytest.mark.parametrize("name", list(MORE_SPECS))
def test_more_robots_model_counts_reset_streams_and_rewards(name, oracle_factory):
    env_id, nq, nv, nu, nbody, njnt, obs_dim, dt, scale, uniform_vel = MORE_SPECS[name]
    m = cp.compile_model(name)
    assert (m.nq, m.nv, m.nu, m.nbody, m.njnt) == (nq, nv, nu, nbody, njnt)
    if name in ("hopper", "walker2d"):  # rootz has ref="1.25" (hopper.xml:19): init_qpos[1] = 1.25, the rest 0
        assert m.qpos0[1] == 1.25 and not np.delete(m.qpos0, 1).any()
    if name == "inverted_double_pendulum":  # test_mujoco_v5.py:489-498: the tip site starts 1.2 above the rail
        d = omj.OracleModel(name).make_data()
        d.reset(), d.forward()
        assert abs(d.get("xpos")[-1][2] + 0.6 - 1.2) < 1e-15
    pend = name.startswith("inverted") or name == "reacher"
    kw = {} if pend else dict(exclude_current_positions_from_observation=False)
    env = gymnasium_amd.make_vec(env_id, num_envs=3, _engine_factory=oracle_factory, **kw)
    assert env.single_observation_space.shape == ((obs_dim + (0 if pend else 1)),) and env.single_action_space.shape == (nu,)
    obs, _ = env.reset(seed=100)
    if name == "reacher":  # reacher_v5.py:209-226: its own reset sequence (goal rejection loop), observation = functions of the state
        for i in range(3):
            g = np.random.Generator(np.random.PCG64(np.random.SeedSequence(100 + i)))
            qpos = g.uniform(low=-0.1, high=0.1, size=4) + m.qpos0
            while True:
                goal = g.uniform(low=-0.2, high=0.2, size=2)
                if np.linalg.norm(goal) < 0.2:
                    break
            qpos[-2:] = goal
            qvel = g.uniform(low=-0.005, high=0.005, size=4)
            assert np.array_equal(obs[i, 4:8], np.concatenate([qpos[2:], qvel[:2]]))
            np.testing.assert_allclose(obs[i, :4], np.concatenate([np.cos(qpos[:2]), np.sin(qpos[:2])]), rtol=0, atol=1e-15)
            assert np.array_equal(env.get_rng_state()[i], gymnasium_amd._native.pcg_words(g))
    elif name != "inverted_double_pendulum":  # (its observation is sin / cos of the state)
        for i in range(3):
            g = np.random.Generator(np.random.PCG64(np.random.SeedSequence(100 + i)))
            qpos = m.qpos0 + g.uniform(low=-scale, high=scale, size=m.nq)
            qvel = g.uniform(low=-scale, high=scale, size=m.nv) if uniform_vel else scale * g.standard_normal(m.nv)
            assert np.array_equal(obs[i, :m.nq], qpos) and np.array_equal(obs[i, m.nq:m.nq + m.nv], qvel)
            assert np.array_equal(env.get_rng_state()[i], gymnasium_amd._native.pcg_words(g))
    env.close()
    a = gymnasium_amd.make_vec(env_id, num_envs=4, _engine_factory=oracle_factory)
    b = gymnasium_amd.make_vec(env_id, num_envs=4, _engine_factory=oracle_factory)
    assert a.single_observation_space.shape == (obs_dim,) and a.single_observation_space.dtype == np.float64
    oa, _ = a.reset(seed=9)
    ob, _ = b.reset(seed=9)
    assert np.array_equal(oa, ob)
    a.action_space.seed(1)
    prev_done, prev_x, terms = np.zeros(4, dtype=bool), None, 0
    for t in range(120):
        act = a.action_space.sample()
        oa, ra, tea, tra, ia = a.step(act)
        ob, rb, teb, _, _ = b.step(act)
        assert np.array_equal(oa, ob) and np.array_equal(ra, rb) and np.array_equal(tea, teb) and np.isfinite(oa).all()
        live = ~prev_done
        if not live.any():  # every sub-env is in its autoreset step: no step-only key is supplied, so none appears (VectorEnv._add_info)
            assert "reward_ctrl" not in ia and "reward_survive" not in ia and (ra == 0).all() and not tea.any()
            prev_done = np.logical_or(tea, tra)
            if name in ("hopper", "walker2d"):
                prev_x, was_reset = ia["x_position"].copy(), np.ones(4, dtype=bool)
            continue
        if name in ("hopper", "walker2d"):
            total = ia["reward_forward"] + ia["reward_ctrl"] + ia["reward_survive"]
            if prev_x is not None:  # info velocity = finite difference of info positions (test_mujoco_v5.py:116-152)
                np.testing.assert_allclose(ia["x_velocity"][live & ~was_reset], ((ia["x_position"] - prev_x) / dt)[live & ~was_reset], rtol=1e-9, atol=1e-9)
            prev_x, was_reset = ia["x_position"].copy(), prev_done.copy()
        elif name == "inverted_pendulum":
            total = ia["reward_survive"]
            assert np.array_equal(tea[live], np.abs(oa[live, 1]) > 0.2)
        elif name == "reacher":
            total = ia["reward_dist"] + ia["reward_ctrl"]
            np.testing.assert_allclose(ia["reward_dist"][live], -np.hypot(oa[live, 8], oa[live, 9]), rtol=1e-12)  # z offsets cancel
        else:
            total = ia["reward_survive"] + ia["distance_penalty"] + ia["velocity_penalty"]
        np.testing.assert_allclose(ra[live], total[live], rtol=1e-12, atol=1e-12)
        assert (ra[prev_done] == 0).all() and not tea[prev_done].any()
        terms += int(tea.sum())
        prev_done = tea | tra
    assert terms > 0 or name == "reacher", "a random policy ends episodes of these robots within 120 steps (Reacher only truncates)"
    a.close(), b.close()


def test_humanoid_standup_model_and_reward_identity(oracle_factory):
    """HumanoidStandup-v5: same tree / masses as humanoid.xml laid on its back (test_mujoco_v5.py:548-558 counts), reward = height /
    opt.timestep - control - impact + 1, never terminates (humanoidstandup_v5.py:423-462)."""
    a, b = cp.compile_model("humanoid"), cp.compile_model("humanoid_standup")
    assert (b.nq, b.nv, b.nu, b.nbody, b.njnt, b.ngeom) == (24, 23, 17, 14, 18, 18)
    np.testing.assert_allclose(a.body_mass, b.body_mass, rtol=1e-14)
    assert b.qpos0[2] == 0.105 and list(b.jnt_range[b.jnt_names.index("left_hip_y")]) == [np.radians(-120), np.radians(20)]
    env = gymnasium_amd.make_vec("HumanoidStandup-v5", num_envs=3, _engine_factory=oracle_factory, exclude_current_positions_from_observation=False)
    obs, info = env.reset(seed=100)
    for i in range(3):
        g = np.random.Generator(np.random.PCG64(np.random.SeedSequence(100 + i)))
        qpos = b.qpos0 + g.uniform(low=-1e-2, high=1e-2, size=b.nq)
        qvel = g.uniform(low=-1e-2, high=1e-2, size=b.nv)
        assert np.array_equal(obs[i, :b.nq], qpos) and np.array_equal(obs[i, b.nq:b.nq + b.nv], qvel)
    np.testing.assert_allclose(info["z_distance_from_origin"], obs[:, 2] - 0.105, rtol=0, atol=1e-15)
    env.action_space.seed(3)
    for t in range(20):
        o, r, te, tr, ia = env.step(env.action_space.sample())
        np.testing.assert_allclose(r, ia["reward_linup"] + ia["reward_quadctrl"] + ia["reward_impact"] + 1, rtol=1e-12)
        np.testing.assert_allclose(ia["reward_linup"], o[:, 2] / 0.003, rtol=1e-12)
        assert not te.any() and o.shape == (3, 350)
    env.close()


# ---- Swimmer-v5: the one robot whose dynamics come from the medium (option density / viscosity), not from contacts ---------------
def _swimmer_fluid_reference(m, d, qpos, qvel):
    """MuJoCo's inertia-box fluid model written independently of oracle/mujoco_core.c fluid(): body Jacobians by central differences of the
    oracle's kinematics (xipos, planar yaw), forces from the documented formulas, generalised force = sum_b Jp^T f + Jr^T t."""
    def kin(q):
        d.set_state(q, qvel, np.zeros(m.nu))
        d.forward()
        xm = d.get("xmat").reshape(-1, 3, 3)
        return d.get("xipos").copy(), np.arctan2(xm[:, 1, 0], xm[:, 0, 0]), xm
    eps, nb = 1e-6, m.nbody
    Jp, Jr = np.zeros((nb, 3, m.nv)), np.zeros((nb, m.nv))
    for k in range(m.nv):
        dq = np.zeros(m.nq)
        dq[k] = eps
        (pa, ya, _), (pb, yb, _) = kin(qpos + dq), kin(qpos - dq)
        Jp[:, :, k] = (pa - pb) / (2 * eps)
        Jr[:, k] = np.angle(np.exp(1j * (ya - yb))) / (2 * eps)
    _, _, xm = kin(qpos)
    out = np.zeros(m.nv)
    for b in range(1, nb):
        R = xm[b] @ m.body_imat[b].reshape(3, 3)
        v, w = R.T @ (Jp[b] @ qvel), R.T @ np.array([0.0, 0.0, Jr[b] @ qvel])
        bx = m.body_fluidbox[b]
        diam = bx.sum() / 3
        f = -3 * np.pi * diam * m.viscosity * v
        t = -np.pi * diam ** 3 * m.viscosity * w
        for i, (j, k) in enumerate(((1, 2), (0, 2), (0, 1))):
            f[i] -= 0.5 * m.density * bx[j] * bx[k] * abs(v[i]) * v[i]
            t[i] -= m.density * bx[i] * (bx[j] ** 4 + bx[k] ** 4) * abs(w[i]) * w[i] / 64
        out += Jp[b].T @ (R @ f) + Jr[b] * (R @ t)[2]
    return out


def test_swimmer_model_and_fluid_forces():
    """swimmer.xml:1-30: 3 capsules of density 1000 (length 1, radius 0.1) on two sliders + three hinges, no contact pair, RK4; the medium
    (density 4000, viscosity 0.1) acts through each body's equivalent inertia box."""
    m = cp.compile_model("swimmer")
    assert (m.nq, m.nv, m.nu, m.nbody, m.njnt, len(m.pair_geom1)) == (5, 5, 2, 4, 5, 0) and m.integrator == "RK4" and m.timestep == 0.01
    vol = np.pi * 0.1 ** 2 * 1.0 + 4 / 3 * np.pi * 0.1 ** 3  # capsule = cylinder + sphere
    np.testing.assert_allclose(m.body_mass[1:], 1000 * vol, rtol=1e-14)
    assert (m.density, m.viscosity) == (4000.0, 0.1) and (m.dof_armature == 0.1).all() and list(m.actuator_gear) == [150.0, 150.0]
    np.testing.assert_allclose(m.jnt_range[3:], np.radians([[-100, 100]] * 2), rtol=1e-15)
    # the box that has the capsule's mass and principal moments: long axis x, equal y / z edges
    I = m.body_inertia[1].reshape(3, 3)
    bx = m.body_fluidbox[1]
    np.testing.assert_allclose(m.body_mass[1] * (bx[1] ** 2 + bx[2] ** 2) / 12, I[0, 0], rtol=1e-12)
    np.testing.assert_allclose(m.body_mass[1] * (bx[0] ** 2 + bx[2] ** 2) / 12, I[1, 1], rtol=1e-12)
    om = omj.OracleModel("swimmer")
    d = om.make_data()
    # straight body sliding along / across itself: closed forms
    for axis, area in ((0, bx[1] * bx[2]), (1, bx[0] * bx[2])):
        for v in (0.7, -1.3):
            qv = np.zeros(5)
            qv[axis] = v
            d.reset(), d.set_state(np.zeros(5), qv, np.zeros(2)), d.forward()
            want = 3 * (-3 * np.pi * bx.sum() / 3 * 0.1 * v - 0.5 * 4000 * area * abs(v) * v)
            np.testing.assert_allclose(d.get("qfrc_passive")[axis], want, rtol=1e-13)
            assert d.get("ncon") == 0 and d.get("nefc") == 0
    rng = np.random.default_rng(3)
    for _ in range(8):
        qpos, qvel = rng.uniform(-1.5, 1.5, 5), rng.uniform(-3, 3, 5)
        want = _swimmer_fluid_reference(m, d, qpos, qvel)
        d.set_state(qpos, qvel, np.zeros(2)), d.forward()
        got = d.get("qfrc_passive")
        np.testing.assert_allclose(got, want, rtol=2e-7, atol=2e-7)
        assert got @ qvel < 0  # the medium only dissipates
        # yaw invariance: turning the swimmer and its velocity together turns the force on the sliders and leaves the hinge torques
        th = 0.9
        c, s = np.cos(th), np.sin(th)
        q2, v2 = qpos.copy(), qvel.copy()
        q2[2] += th
        q2[:2], v2[:2] = [c * qpos[0] - s * qpos[1], s * qpos[0] + c * qpos[1]], [c * qvel[0] - s * qvel[1], s * qvel[0] + c * qvel[1]]
        d.set_state(q2, v2, np.zeros(2)), d.forward()
        g2 = d.get("qfrc_passive")
        np.testing.assert_allclose(g2[:2], [c * got[0] - s * got[1], s * got[0] + c * got[1]], rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(g2[2:], got[2:], rtol=1e-11, atol=1e-11)


def test_swimmer_env_reset_stream_reward_and_info(oracle_factory):
    m = cp.compile_model("swimmer")
    env = gymnasium_amd.make_vec("Swimmer-v5", num_envs=3, _engine_factory=oracle_factory, exclude_current_positions_from_observation=False)
    assert env.single_observation_space.shape == (10,) and env.single_action_space.shape == (2,)
    obs, info = env.reset(seed=100)
    for i in range(3):  # swimmer_v5.py:279-294: uniform noise on positions and velocities
        g = np.random.Generator(np.random.PCG64(np.random.SeedSequence(100 + i)))
        qpos = m.qpos0 + g.uniform(low=-0.1, high=0.1, size=5)
        qvel = g.uniform(low=-0.1, high=0.1, size=5)
        assert np.array_equal(obs[i], np.concatenate([qpos, qvel]))
        assert np.array_equal(env.get_rng_state()[i], gymnasium_amd._native.pcg_words(g))
    assert np.array_equal(info["x_position"], obs[:, 0]) and np.array_equal(info["y_position"], obs[:, 1])
    np.testing.assert_allclose(info["distance_from_origin"], np.hypot(obs[:, 0], obs[:, 1]), rtol=1e-15)
    env.action_space.seed(0)
    prev = obs
    for t in range(30):
        act = env.action_space.sample()
        obs, rew, term, trunc, info = env.step(act)
        assert not term.any() and not trunc.any() and np.isfinite(obs).all()
        assert np.array_equal(info["x_position"], obs[:, 0]) and np.array_equal(info["y_position"], obs[:, 1])  # test_mujoco_v5.py:48-77
        assert np.array_equal(info["x_velocity"], (obs[:, 0] - prev[:, 0]) / 0.04) and np.array_equal(info["y_velocity"], (obs[:, 1] - prev[:, 1]) / 0.04)
        assert np.array_equal(rew, info["reward_forward"] + info["reward_ctrl"])  # test_mujoco_v5.py:297-301
        ctrl = np.float32(1e-4) * np.sum(np.square(act), axis=1, dtype=np.float32)
        assert np.array_equal(info["reward_ctrl"], -ctrl.astype(np.float64)) and np.array_equal(info["reward_forward"], info["x_velocity"])
        prev = obs
    env.close()
    env = gymnasium_amd.make_vec("Swimmer-v5", num_envs=2, _engine_factory=oracle_factory, max_episode_steps=7)
    assert env.single_observation_space.shape == (8,) and env.spec.reward_threshold == 360.0 if env.spec is not None else True
    env.reset(seed=1)
    for t in range(7):
        _, _, term, trunc, _ = env.step(np.zeros((2, 2), dtype=np.float32))
    assert trunc.all() and not term.any()
    env.close()


def test_swimmer_swims():
    """The gait that makes a three-link swimmer move: a travelling wave down the two joints propels it along its own axis, against the
    drag of the medium (without the medium the centre of mass could not move at all: no external force)."""
    om = omj.OracleModel("swimmer")
    d = om.make_data()
    d.reset()
    x0 = None
    for t in range(1500):
        ph = 2 * np.pi * t * 0.01 / 1.0
        d.set_state(None, None, np.array([np.sin(ph), np.sin(ph - 2.0)]) * 0.6)
        d.step(1)
        if t == 0:
            x0 = d.get("subtree_com")[1].copy()
    moved = d.get("subtree_com")[1] - x0
    assert np.isfinite(d.get("qpos")).all() and np.hypot(moved[0], moved[1]) > 0.3


# ---- Pusher-v5: arm + sliding cylinder; the one pair type MuJoCo itself has no analytic function for (capsule - cylinder) ---------
def _segment_cylinder_distance(a, b, c, R, H):
    """Independent formulation: signed distance of the segment a-b to the z-aligned solid cylinder (centre c), by scalar minimisation of the
    point distance (outside: hypot of the axial / radial excess; inside: the larger, negative, of the two)."""
    from scipy.optimize import minimize_scalar

    def sd(t):
        p = a + t * (b - a) - c
        ea, er = abs(p[2]) - H, np.hypot(p[0], p[1]) - R
        return max(ea, er) if (ea <= 0 and er <= 0) else float(np.hypot(max(ea, 0.0), max(er, 0.0)))
    ts = np.linspace(0, 1, 4001)
    vals = np.array([sd(t) for t in ts])
    k = int(vals.argmin())
    lo, hi = ts[max(k - 1, 0)], ts[min(k + 1, len(ts) - 1)]
    res = minimize_scalar(sd, bounds=(lo, hi), method="bounded", options=dict(xatol=1e-13))
    return min(res.fun, vals[k])


def test_pusher_model_and_capsule_cylinder_contacts():
    m = cp.compile_model("pusher")
    assert (m.nq, m.nv, m.nu, m.nbody, m.njnt) == (11, 11, 7, 13, 11) and m.integrator == "Euler" and not m.gravity.any()
    np.testing.assert_allclose(m.body_mass[11], 0.01 * np.pi * 0.05 ** 2 * 0.1, rtol=1e-14)  # the object: density 0.01 (pusher_v5.xml:77)
    names = [(m.geom_names[a], m.geom_names[b]) for a, b in zip(m.pair_geom1, m.pair_geom2)]
    assert names == [("floor", "wr0"), ("floor", "wr1"), ("floor", "wr2"), ("wr0", "object"), ("wr1", "object"), ("wr2", "object")]
    assert (m.pair_condim == 1).all() and (m.pair_margin == 0.002).all()
    om = omj.OracleModel("pusher")
    d = om.make_data()
    d.reset(), d.forward()
    np.testing.assert_allclose(d.get("xpos")[10], [0.821, -0.6, 0.0], atol=1e-15)  # tips_arm of the documented start pose
    assert d.get("ncon") == 0
    rng = np.random.default_rng(5)
    gi = {n: m.geom_names.index(n) for n in ("wr0", "wr1", "wr2", "object")}
    seen = {"touching": 0, "penetrating": 0, "deep": 0}
    for trial in range(60):
        q = np.zeros(11)
        q[:7] = [rng.uniform(lo, hi) for lo, hi in m.jnt_range[:7]]
        d.set_state(q, np.zeros(11), np.zeros(7)), d.forward()
        gx, gm = d.get("geom_xpos"), d.get("geom_xmat").reshape(-1, 3, 3)
        # move the object next to one of the wrist capsules: a random point of the segment plus a random offset around the contact range
        w = gi[("wr0", "wr1", "wr2")[trial % 3]]
        half, axis = m.geom_size[w][1], gm[w][:, 2]
        target = gx[w] + rng.uniform(-half, half) * axis + rng.normal(size=3) * (0.0 if trial % 4 == 0 else 0.05)
        q[7], q[8] = target[1] - m.body_pos[11][1], target[0] - m.body_pos[11][0]  # slider y, then slider x
        d.set_state(q, np.zeros(11), np.zeros(7)), d.forward()
        gx = d.get("geom_xpos")
        con = d.get("contact")
        cyl = gx[gi["object"]]
        for name in ("wr0", "wr1", "wr2"):
            g = gi[name]
            half, axis, rc = m.geom_size[g][1], gm[g][:, 2], m.geom_size[g][0]
            want = _segment_cylinder_distance(gx[g] - half * axis, gx[g] + half * axis, cyl, 0.05, 0.05) - rc
            rows = [r for r in con if int(r[13]) == g and int(r[14]) == gi["object"]]
            if want < 0.002 - 1e-9:
                assert len(rows) == 1
                r = rows[0]
                np.testing.assert_allclose(r[0], want, rtol=0, atol=2e-9)
                n = r[4:7]
                np.testing.assert_allclose(np.linalg.norm(n), 1.0, rtol=1e-12)
                seen["touching" if want > 0 else ("penetrating" if want > -rc else "deep")] += 1
                if want > -rc + 1e-3:  # the segment itself is outside the cylinder: the normal points from the capsule into the cylinder
                    inside_pt = r[1:4] + n * 1e-3 * 0 + n * (abs(r[0]) * 0.5 + 1e-4)
                    p = inside_pt - cyl
                    assert abs(p[2]) <= 0.05 + 1e-9 and np.hypot(p[0], p[1]) <= 0.05 + 1e-9
            elif want > 0.002 + 1e-9:
                assert not rows
    assert seen["touching"] > 0 and seen["penetrating"] > 5


def test_pusher_arm_pushes_the_object():
    """A wrist capsule overlapping the cylinder accelerates it away from the arm (frictionless contact, equal and opposite force)."""
    m = cp.compile_model("pusher")
    om = omj.OracleModel("pusher")
    d = om.make_data()
    q = np.zeros(11)
    q[1], q[3] = 0.55, -0.45  # the arm reaches down to the table plane's height range
    d.reset(), d.set_state(q, np.zeros(11), np.zeros(7)), d.forward()
    gx = d.get("geom_xpos")
    w = m.geom_names.index("wr0")
    # place the object 0.06 in front (+x) of the wrist cross-bar: the gap between the surfaces is 0.06 - 0.05 - 0.02 < 0
    q[7], q[8] = gx[w][1] - m.body_pos[11][1], gx[w][0] + 0.06 - m.body_pos[11][0]
    d.set_state(q, np.zeros(11), np.zeros(7)), d.forward()
    assert abs(gx[w][2] - (-0.275)) < 0.05 + 0.02  # the bar is at the cylinder's height: a contact must exist
    assert d.get("ncon") >= 1 and d.get("nefc") >= 1
    qacc = d.get("qacc")
    assert qacc[8] > 0, "the object is pushed along +x, away from the bar"
    d.step(20)
    assert d.get("qpos")[8] > q[8]


def test_pusher_env_reset_stream_reward_and_truncation(oracle_factory):
    m = cp.compile_model("pusher")
    env = gymnasium_amd.make_vec("Pusher-v5", num_envs=3, _engine_factory=oracle_factory)
    assert env.single_observation_space.shape == (23,) and env.single_action_space.shape == (7,)
    assert (env.single_action_space.low == -2).all() and (env.single_action_space.high == 2).all()
    obs, info = env.reset(seed=100)
    assert info == {}
    for i in range(3):  # pusher_v5.py:293-315
        g = np.random.Generator(np.random.PCG64(np.random.SeedSequence(100 + i)))
        while True:
            cyl = np.concatenate([g.uniform(low=-0.3, high=0, size=1), g.uniform(low=-0.2, high=0.2, size=1)])
            if np.linalg.norm(cyl - np.asarray([0, 0])) > 0.17:
                break
        qvel = g.uniform(low=-0.005, high=0.005, size=11)
        qvel[-4:] = 0
        assert np.array_equal(obs[i, :7], np.zeros(7)) and np.array_equal(obs[i, 7:14], qvel[:7])
        # object = body pos + (slider x, slider y); the FIRST draw drives the y slider (joint order of the XML)
        np.testing.assert_allclose(obs[i, 17:20], m.body_pos[11] + [cyl[1], cyl[0], 0.0], rtol=0, atol=1e-16)
        np.testing.assert_allclose(obs[i, 20:23], m.body_pos[12], rtol=0, atol=1e-16)
        np.testing.assert_allclose(obs[i, 14:17], [0.821, -0.6, 0.0], rtol=0, atol=1e-15)
        assert np.array_equal(env.get_rng_state()[i], gymnasium_amd._native.pcg_words(g))
    env.action_space.seed(0)
    for t in range(100):
        act = env.action_space.sample()
        obs, rew, term, trunc, info = env.step(act)
        assert np.array_equal(rew, info["reward_dist"] + info["reward_ctrl"] + info["reward_near"])  # test_mujoco_v5.py:285-289
        ctrl = -np.square(act).sum(axis=1, dtype=np.float32) * np.float32(0.1)
        assert np.array_equal(info["reward_ctrl"], ctrl.astype(np.float64))
        np.testing.assert_allclose(info["reward_near"], -np.linalg.norm(obs[:, 17:20] - obs[:, 14:17], axis=1) * 0.5, rtol=1e-14)
        np.testing.assert_allclose(info["reward_dist"], -np.linalg.norm(obs[:, 17:20] - obs[:, 20:23], axis=1), rtol=1e-14)
        assert not term.any() and trunc.all() == (t == 99) and np.isfinite(obs).all()
    env.close()
