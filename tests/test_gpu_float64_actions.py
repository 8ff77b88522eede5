"""-m gpu: float64 action rows through the C ABI (mi_step_io.actions_dtype = MI_F64 / MI_F64_WEAK, mi_rollout_io.actions_in_dtype).

The reference hands the caller's action rows to the scalar envs as they are (vector/sync_vector_env.py:274 iterate(); pendulum.py:127-139,
continuous_mountain_car.py:150-178, mujoco_env.py:148 `data.ctrl[:] = ctrl`): a float64 batch is not rounded to the space's float32 and
NumPy's promotions make parts of the step float64 arithmetic.  The oracle's float64 paths are pinned on the real reference under its strict
data_equivalence (tests/test_real_gymnasium.py::test_float64_action_rows_*, ::test_mountaincar_continuous_clamps_*,
tests/test_mujoco_fixture_pipeline.py VECTOR_CHECK); here the HIP engine is compared with the oracle: classic control bit for bit, the
MuJoCo kinds to the tolerance of tests/test_gpu_mujoco.py.
"""
import numpy as np
import pytest

import gymnasium_amd
import parity_suite as ps

pytestmark = pytest.mark.gpu


def _action_batch(rng, kind, n, hi):
    a = rng.uniform(-1.3 * hi, 1.3 * hi, (n, 1))
    a[rng.random(n) < 0.1] = rng.choice([-hi, hi, 0.0])
    return [a, a.astype(np.float32), a.tolist(), rng.integers(-2, 3, (n, 1))][kind]


@pytest.mark.parametrize("mode", ["NextStep", "SameStep"])
@pytest.mark.parametrize("key,hi", [("pendulum", 2.0), ("mountaincar_continuous", 1.0)])
def test_classic_float64_rows_equal_the_oracle(key, hi, mode, oracle_factory):
    n, T = 2048, 260
    gpu = ps.make(key, n, None, autoreset_mode=mode, max_episode_steps=70)
    cpu = ps.make(key, n, oracle_factory, autoreset_mode=mode, max_episode_steps=70)
    og, _ = gpu.reset(seed=5)
    oc, _ = cpu.reset(seed=5)
    assert np.array_equal(og, oc)
    rng = np.random.default_rng(8)
    for t in range(T):
        a = _action_batch(rng, t % 4, n, hi)
        sg, sc = gpu.step(a), cpu.step(a)
        for k, what in enumerate(("obs", "reward", "terminated", "truncated")):
            assert sg[k].dtype == sc[k].dtype and np.array_equal(sg[k], sc[k]), f"{key} {what} t={t} kind={t % 4}"
        if mode == "SameStep" and "final_obs" in sc[4]:
            assert np.array_equal(sg[4]["_final_obs"], sc[4]["_final_obs"])
            for i in np.flatnonzero(sc[4]["_final_obs"]):
                assert np.array_equal(sg[4]["final_obs"][i], sc[4]["final_obs"][i])
    sgs, scs = gpu.get_state(), cpu.get_state()
    assert np.array_equal(sgs[0], scs[0]) and np.array_equal(sgs[1], scs[1]) and np.array_equal(sgs[2], scs[2])
    sg, sc = gpu.statistics(), cpu.statistics()
    assert all(sg[k] == sc[k] for k in ("env_steps", "reset_steps", "episodes", "length_sum"))
    np.testing.assert_allclose(sg["return_sum"], sc["return_sum"], rtol=1e-12)  # (a sum over workgroups: another order than the oracle's)
    gpu.close(), cpu.close()


@pytest.mark.parametrize("state_f32", [True, False])
def test_mountaincar_continuous_clamps_equal_the_oracle(state_f32, oracle_factory):
    """The places where continuous_mountain_car.py's scalars change kind (speed / position clamps, the wall, the goal), teacher-forced."""
    n = 4096
    gpu = ps.make("mountaincar_continuous", n, None, autoreset_mode="SameStep")
    cpu = ps.make("mountaincar_continuous", n, oracle_factory, autoreset_mode="SameStep")
    gpu.reset(seed=1), cpu.reset(seed=1)
    rng = np.random.default_rng(11)
    centres = np.array([[-1.2, -0.07], [-1.2, 0.0], [-1.199, -0.06], [0.6, 0.07], [0.599, 0.069], [0.45, 0.0], [0.449, 0.01], [-0.5, 0.07], [-0.5, -0.07], [0.3, 0.0695]])
    goals = 0
    for trial in range(24):
        st = centres[rng.integers(0, len(centres), n)] + rng.normal(0, [2e-3, 1e-3], (n, 2)) * (rng.random((n, 1)) < 0.7)
        st = np.clip(st, [-1.2, -0.07], [0.6, 0.07])
        if state_f32:
            st = st.astype(np.float32).astype(np.float64)
        flags = np.full(n, 2 if state_f32 else 0, np.uint8)
        for env in (gpu, cpu):
            env.set_state(st, np.zeros(n, np.int32), flags)
        a = _action_batch(rng, trial % 3, n, 1.0)
        sg, sc = gpu.step(a), cpu.step(a)
        for k, what in enumerate(("obs", "reward", "terminated", "truncated")):
            assert np.array_equal(sg[k], sc[k]), f"{what} trial={trial}"
        goals += int(sc[2].sum())
        assert np.array_equal(gpu.get_state()[0], cpu.get_state()[0])
    assert goals > 100
    gpu.close(), cpu.close()


@pytest.mark.parametrize("key", ["pendulum", "mountaincar_continuous"])
def test_fused_rollout_with_float64_actions_equals_stepping(key):
    import torch

    n, T = 1024, 96
    a = torch.from_numpy(np.random.default_rng(3).uniform(-2.2, 2.2, (T, n, 1)))  # float64
    one = ps.make(key, n, None, output="torch", max_episode_steps=40)
    two = ps.make(key, n, None, output="torch", max_episode_steps=40)
    one.reset(seed=4), two.reset(seed=4)
    out = one.rollout(T, a.cuda())
    for t in range(T):
        o, r, te, tr, _ = two.step(a[t].cuda())
        assert torch.equal(out["obs"][t], o) and torch.equal(out["rewards"][t], r) and torch.equal(out["terminations"][t], te) and torch.equal(out["truncations"][t], tr), t
    # ... and the float32-rounded batch gives a DIFFERENT trajectory (the rows really are taken un-rounded)
    three = ps.make(key, n, None, output="torch", max_episode_steps=40)
    three.reset(seed=4)
    out32 = three.rollout(T, a.float().cuda())
    assert not torch.equal(out32["rewards"], out["rewards"])
    assert np.array_equal(one.get_state()[0], two.get_state()[0])
    one.close(), two.close(), three.close()


MJ = ["HalfCheetah-v5", "Ant-v5", "Hopper-v5", "Walker2d-v5", "Reacher-v5", "Pusher-v5", "Swimmer-v5", "InvertedPendulum-v5", "InvertedDoublePendulum-v5", "Humanoid-v5"]


@pytest.mark.parametrize("env_id", MJ)
def test_mujoco_float64_rows_equal_the_oracle(env_id, oracle_factory):
    """Float64 action rows reach data.ctrl un-rounded and make the control cost float64 arithmetic (half_cheetah_v5.py:216-218 ...): the HIP
    engine (the default kernel of each robot) vs the oracle over re-synchronised 5-step windows, float32 and float64 batches alternating;
    `reward_ctrl` has the dtype NumPy gives it (float32 only for a float32 row)."""
    n = 256
    gpu = gymnasium_amd.make_vec(env_id, num_envs=n)
    cpu = gymnasium_amd.make_vec(env_id, num_envs=n, _engine_factory=oracle_factory)
    og, _ = gpu.reset(seed=21)
    oc, _ = cpu.reset(seed=21)
    np.testing.assert_allclose(og, oc, rtol=0, atol=1e-12)
    rng = np.random.default_rng(2)
    lo, hi = cpu.single_action_space.low.astype(np.float64), cpu.single_action_space.high.astype(np.float64)
    for t in range(20):
        a = rng.uniform(lo, hi, (n, len(lo)))
        if t % 2:
            a = a.astype(np.float32)
        sg, sc = gpu.step(a), cpu.step(a)
        np.testing.assert_allclose(sg[0], sc[0], rtol=0, atol=1e-8, err_msg=f"{env_id} obs t={t}")
        np.testing.assert_allclose(sg[1], sc[1], rtol=0, atol=1e-8, err_msg=f"{env_id} reward t={t}")
        assert np.array_equal(sg[2], sc[2]) and np.array_equal(sg[3], sc[3])
        for k in sc[4]:
            assert sg[4][k].dtype == sc[4][k].dtype, (env_id, k, t)
        if "reward_ctrl" in sc[4] and env_id != "Humanoid-v5":  # a pure function of the action row: every bit, in either dtype
            assert sc[4]["reward_ctrl"].dtype == (np.float32 if t % 2 else np.float64)
            live = sc[4]["_reward_ctrl"]
            assert np.array_equal(sg[4]["reward_ctrl"][live], sc[4]["reward_ctrl"][live]), (env_id, t)
        if (t + 1) % 5 == 0:
            gpu.set_state(*cpu.get_state())
    gpu.close(), cpu.close()


def test_invalid_actions_dtype_is_rejected():
    from gymnasium_amd import _native

    env = gymnasium_amd.make_vec("Pendulum-v1", num_envs=4)
    env.reset(seed=0)
    eng = env._engine
    with pytest.raises(_native.NativeError, match="actions_dtype"):
        eng.step(np.zeros((4, 1), np.float32), env._obs, env._rew, env._term, env._trunc, actions_dtype=_native.MI_I64)
    env.close()
