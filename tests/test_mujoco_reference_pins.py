"""The few NUMERIC facts about the MuJoCo-family envs that the reference's own test-suite pins (tests/envs/mujoco/test_mujoco_v5.py),
restated against the oracle (CPU) and the HIP engine (-m gpu):

  :480-488  test_inverted_double_pendulum_max_height   site "tip" z == 1.2 at the zero-noise reset
  :353-372  test_ant_com                               qpos[0] == body("torso").xpos[0] after mj_kinematics
  :659-670  test_dt                                    env.dt = model.opt.timestep * frame_skip
  :693-699  test_reset_noise_scale                     reset_noise_scale=0 -> qpos == init_qpos, qvel == init_qvel
and the info entries humanoid_v5.py:486-487 / humanoidstandup_v5.py:433-434 add (tendon_length / tendon_velocity: fixed tendons are
linear in qpos / qvel, evaluated at the LAST forward pass like every other mjData field the env reads after mj_step).
"""
import numpy as np
import pytest

import gymnasium_amd
from gymnasium_amd.envs.mujoco import compiler as cp
from oracle import mujoco as omj


def test_inverted_double_pendulum_max_height_oracle():
    om = omj.OracleModel("inverted_double_pendulum")
    d = om.make_data()
    d.reset()  # qpos = init_qpos, qvel = 0: what reset_noise_scale=0 gives
    d.forward()
    (body, pos), = om.m.sites
    tip = d.get("xpos")[body] + d.get("xmat")[body].reshape(3, 3) @ np.asarray(pos)
    assert tip[2] == 1.2  # exact, as in the reference test
    kin = cp.kinematics(om.m, om.m.qpos0)  # the independent NumPy formulation
    assert (kin["xpos"][body] + kin["xmat"][body] @ np.asarray(pos))[2] == 1.2


def test_ant_com_oracle():
    """data.qpos[0] == data.body('torso').xpos[0] at a freshly evaluated state (after reset's mj_forward, and after
    env.step + mj_kinematics)."""
    om = omj.OracleModel("ant")
    d, rng = om.make_data(), np.random.default_rng(0)
    q = om.m.qpos0 + rng.uniform(-0.1, 0.1, om.m.nq)
    d.set_state(q, rng.normal(size=om.m.nv) * 0.1, rng.uniform(-1, 1, om.m.nu))
    d.forward()
    assert d.get("qpos")[0] == d.get("xpos")[1][0]
    d.step(5)
    d.forward()  # mj_kinematics at the stepped state
    assert d.get("qpos")[0] == d.get("xpos")[1][0]


def test_dt(oracle_factory):
    a = gymnasium_amd.make_vec("Ant-v5", num_envs=2, include_cfrc_ext_in_observation=False, _engine_factory=oracle_factory)
    b = gymnasium_amd.make_vec("Ant-v5", num_envs=2, include_cfrc_ext_in_observation=False, frame_skip=1, _engine_factory=oracle_factory)
    assert a.dt == 0.01 * 5 and b.dt == 0.01  # test_dt compares 5 x 0.01 with 1 x 0.05 (it patches the timestep)
    h = gymnasium_amd.make_vec("Humanoid-v5", num_envs=1, _engine_factory=oracle_factory)
    assert h.dt == 0.003 * 5
    # info velocity = displacement / dt with that dt (test_mujoco_v5.py:116-152)
    a.reset(seed=0)
    x0 = a.get_state()[0][:, -2].copy()  # the tracked x the next step differences against
    _, _, _, _, info = a.step(np.zeros((2, 8), np.float32))
    x1 = a.get_state()[0][:, -2]
    np.testing.assert_allclose(info["x_velocity"], (x1 - x0) / a.dt, rtol=0, atol=1e-15)
    a.close(), b.close(), h.close()


@pytest.mark.parametrize("env_id", ["Humanoid-v5", "HumanoidStandup-v5"])
def test_tendon_infos_oracle(env_id, oracle_factory):
    n = 4
    env = gymnasium_amd.make_vec(env_id, num_envs=n, _engine_factory=oracle_factory)
    _, info = env.reset(seed=2)
    st = env.get_state()[0]
    qpos, qvel = st[:, :24], st[:, 24:47]
    # left_hipknee = -left_hip_y + left_knee, right_hipknee = -right_hip_y + right_knee (humanoid.xml:91-100)
    want_len = np.stack([qpos[:, 17] - qpos[:, 16], qpos[:, 13] - qpos[:, 12]], axis=1)
    want_vel = np.stack([qvel[:, 16] - qvel[:, 15], qvel[:, 12] - qvel[:, 11]], axis=1)
    assert info["tendon_length"].shape == (n, 2) and info["_tendon_length"].all()
    np.testing.assert_array_equal(info["tendon_length"], want_len)
    np.testing.assert_array_equal(info["tendon_velocity"], want_vel)
    order = [k for k in info if not k.startswith("_")]  # humanoid_v5.py:534-541 / humanoidstandup_v5.py:479-486
    assert order == (["x_position", "y_position", "tendon_length", "tendon_velocity", "distance_from_origin"] if env_id == "Humanoid-v5" else
                     ["x_position", "y_position", "z_distance_from_origin", "tendon_length", "tendon_velocity"])
    env.action_space.seed(0)
    _, _, _, _, info = env.step(env.action_space.sample())
    assert info["tendon_length"].shape == (n, 2) and info["tendon_velocity"].shape == (n, 2) and info["_tendon_velocity"].all()
    # the values are those of the LAST forward pass of the step (RK4: its fourth stage), not of the integrated state: close, not equal
    st = env.get_state()[0]
    now = np.stack([st[:, 17] - st[:, 16], st[:, 13] - st[:, 12]], axis=1)
    assert np.abs(info["tendon_length"] - now).max() < 0.05 and not np.array_equal(info["tendon_length"], now)
    env.close()


def test_tendon_values_are_those_of_the_last_forward_pass():
    om = omj.OracleModel("humanoid")
    d, rng = om.make_data(), np.random.default_rng(1)
    d.set_state(om.m.qpos0 + rng.uniform(-0.01, 0.01, om.m.nq), rng.uniform(-0.5, 0.5, om.m.nv), rng.uniform(-0.4, 0.4, om.m.nu))
    d.step(1)
    stage_len = d.get("ten_length")[:2].copy()
    q, v = d.get("qpos"), d.get("qvel")
    d.forward()
    fresh = d.get("ten_length")[:2]
    np.testing.assert_array_equal(fresh, [q[17] - q[16], q[13] - q[12]])
    np.testing.assert_array_equal(d.get("ten_velocity")[:2], [v[16] - v[15], v[12] - v[11]])
    assert not np.array_equal(stage_len, fresh)


# ------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_inverted_double_pendulum_max_height_gpu():
    """Zero-noise reset = the upright configuration: after one zero-action step the distance penalty the env reports is
    0.01 x^2 + (y - 2)^2 with the tip height y = 1.2 of the last forward pass (gravity's 1e-5 x-component moves it by ~1e-9)."""
    env = gymnasium_amd.make_vec("InvertedDoublePendulum-v5", num_envs=64, reset_noise_scale=0.0)
    obs, _ = env.reset(seed=0)
    assert np.array_equal(obs[:, :1], np.zeros((64, 1))) and np.array_equal(obs[:, 3:5], np.ones((64, 2)))  # x = 0, cos = 1
    _, _, te, _, info = env.step(np.zeros((64, 1), np.float32))
    assert not te.any()
    np.testing.assert_allclose(-info["distance_penalty"], (1.2 - 2.0) ** 2, rtol=0, atol=1e-7)
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("env_id", ["Humanoid-v5", "HumanoidStandup-v5"])
def test_tendon_infos_gpu(env_id, oracle_factory):
    n = 64
    gpu = gymnasium_amd.make_vec(env_id, num_envs=n)
    cpu = gymnasium_amd.make_vec(env_id, num_envs=n, _engine_factory=oracle_factory)
    _, ig = gpu.reset(seed=2)
    _, ic = cpu.reset(seed=2)
    assert np.array_equal(ig["tendon_length"], ic["tendon_length"]) and np.array_equal(ig["tendon_velocity"], ic["tendon_velocity"])
    gpu.action_space.seed(0)
    for _ in range(5):
        a = gpu.action_space.sample()
        ig, ic = gpu.step(a)[4], cpu.step(a)[4]
        np.testing.assert_allclose(ig["tendon_length"], ic["tendon_length"], rtol=0, atol=1e-7)
        np.testing.assert_allclose(ig["tendon_velocity"], ic["tendon_velocity"], rtol=0, atol=1e-5)
        assert ig["tendon_length"].shape == (n, 2)
    gpu.close(), cpu.close()


@pytest.mark.gpu
def test_dt_gpu():
    env = gymnasium_amd.make_vec("Ant-v5", num_envs=8, include_cfrc_ext_in_observation=False)
    assert env.dt == 0.05
    env.reset(seed=0)
    x0 = env.get_state()[0][:, -2].copy()
    info = env.step(np.zeros((8, 8), np.float32))[4]
    np.testing.assert_allclose(info["x_velocity"], (env.get_state()[0][:, -2] - x0) / env.dt, rtol=0, atol=1e-15)
    env.close()
