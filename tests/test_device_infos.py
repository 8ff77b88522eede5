"""output="torch": the infos of a step are DEVICE tensors assembled without reading anything back (HipVectorEnv._build_infos_device) --
x_position / reward terms of the MuJoCo kinds, SAME_STEP final observations, episode statistics.  Checked here against the NumPy dict of
the same env on the checker backend (whose "device" tensors are host tensors: same code path, no GPU needed); the GPU run of the same
comparison is tests/test_gpu_mujoco.py::test_device_resident_infos_equal_the_numpy_infos."""
import numpy as np
import pytest

import gymnasium_amd


def compare_device_infos(env_id, factory, mode, steps=80, **kw):
    import torch

    common = dict(num_envs=6, autoreset_mode=mode, record_episode_statistics=True, max_episode_steps=30, **kw)
    extra = {} if factory is None else {"_engine_factory": factory}
    a = gymnasium_amd.make_vec(env_id, **common, **extra)
    b = gymnasium_amd.make_vec(env_id, output="torch", **common, **extra)
    host = lambda x: x.cpu().numpy()  # noqa: E731
    oa, _ = a.reset(seed=1)
    ob, _ = b.reset(seed=1)
    assert np.array_equal(oa, host(ob))
    a.action_space.seed(0)
    seen_final = seen_episode = False
    for t in range(steps):
        act = a.action_space.sample()
        ra, rb = a.step(act), b.step(torch.from_numpy(act).to(ob.device))
        assert np.array_equal(ra[0], host(rb[0])), t
        ia, ib = ra[4], rb[4]
        assert all(isinstance(v, (torch.Tensor, dict)) for v in ib.values()), "every info entry is a device tensor"
        for k, v in ia.items():
            if k not in ("final_obs", "final_info", "episode"):
                assert np.array_equal(v, host(ib[k])) and v.dtype == host(ib[k]).dtype, (mode, t, k, v.dtype, host(ib[k]).dtype)
        assert set(ia) <= set(ib), "the device dict has a static key set: a superset of the host dict's"
        for k in set(ib) - set(ia):  # keys the host dict drops because no sub-env supplied them in this step
            if k.startswith("_") and not isinstance(ib[k], dict):
                assert not host(ib[k]).any(), (mode, t, k)
        if "final_obs" in ia:
            seen_final = True
            m = ia["_final_obs"]
            assert np.array_equal(m, host(ib["_final_obs"]))
            for i in np.flatnonzero(m):
                assert np.array_equal(ia["final_obs"][i], host(ib["final_obs"][i]))
            for k, v in ia["final_info"].items():
                assert np.array_equal(v, host(ib["final_info"][k])) and v.dtype == host(ib["final_info"][k]).dtype, k
        if "episode" in ia:
            seen_episode = True
            m = ia["_episode"]
            assert np.array_equal(m, host(ib["_episode"]))
            assert np.array_equal(ia["episode"]["r"], np.where(m, host(ib["episode"]["r"]), 0.0))
            assert np.array_equal(ia["episode"]["l"], np.where(m, host(ib["episode"]["l"]), 0))
            assert (host(ib["episode"]["t"])[m] > 0).all() and (host(ib["episode"]["t"])[~m] == 0).all()
        else:
            assert not host(ib["_episode"]).any()
        d = ra[2] | ra[3]
        if mode == "Disabled" and d.any():
            a.reset(options={"reset_mask": d}), b.reset(options={"reset_mask": d})
    assert seen_episode and (seen_final or mode != "SameStep")
    assert a.episode_count == b.episode_count > 0
    a.close(), b.close()


@pytest.mark.parametrize("mode", ["NextStep", "SameStep", "Disabled"])
@pytest.mark.parametrize("env_id", ["Hopper-v5", "Ant-v5", "InvertedPendulum-v5"])  # (InvertedPendulum: an int64 info entry, reward_survive)
def test_device_infos_equal_numpy_infos(env_id, mode, oracle_factory):
    compare_device_infos(env_id, oracle_factory, mode, steps=80 if env_id == "Hopper-v5" else 45)


def test_record_episode_statistics_wrapper_on_device_infos(oracle_factory):
    import torch

    from gymnasium_amd import wrappers as gw

    a = gw.RecordEpisodeStatistics(gymnasium_amd.make_vec("CartPole-v1", num_envs=5, _engine_factory=oracle_factory))
    b = gw.RecordEpisodeStatistics(gymnasium_amd.make_vec("CartPole-v1", num_envs=5, output="torch", _engine_factory=oracle_factory))
    a.reset(seed=2), b.reset(seed=2)
    a.action_space.seed(1)
    for _ in range(150):
        act = a.action_space.sample()
        a.step(act), b.step(torch.from_numpy(act))
    assert a.episode_count == b.episode_count > 10
    assert list(a.return_queue) == list(b.return_queue) and list(a.length_queue) == list(b.length_queue)


def test_episode_count_survives_a_switch_of_the_output_mode(oracle_factory):
    """episode_count lives on the device with output="torch" (no read-back per step): re-allocating the buffers -- set_output(), enabling the
    statistics late -- folds it into the host total instead of dropping it."""
    import torch

    env = gymnasium_amd.make_vec("CartPole-v1", num_envs=8, output="torch", record_episode_statistics=True, _engine_factory=oracle_factory)
    env.reset(seed=3)
    env.action_space.seed(1)
    for _ in range(120):
        env.step(torch.from_numpy(env.action_space.sample()))
    before = env.episode_count
    assert before > 5
    env.set_output("numpy")
    assert env.episode_count == before
    env.reset(seed=3)
    for _ in range(60):
        env.step(env.action_space.sample())
    mid = env.episode_count
    assert mid > before
    env.set_output("torch")
    assert env.episode_count == mid
    env.close()


def test_device_info_masks_are_private_copies(oracle_factory):
    """The masks handed out with the device infos are the caller's to edit: an in-place change must not reach the next step's bookkeeping."""
    import torch

    a = gymnasium_amd.make_vec("Hopper-v5", num_envs=6, output="torch", record_episode_statistics=True, max_episode_steps=7, _engine_factory=oracle_factory)
    b = gymnasium_amd.make_vec("Hopper-v5", num_envs=6, output="torch", record_episode_statistics=True, max_episode_steps=7, _engine_factory=oracle_factory)
    a.reset(seed=5), b.reset(seed=5)
    a.action_space.seed(2)
    for t in range(30):
        act = torch.from_numpy(a.action_space.sample())
        ra, rb = a.step(act), b.step(act)
        for k in ("x_position", "_x_position", "_episode", "_x_velocity"):
            assert torch.equal(ra[4][k], rb[4][k]), (t, k)
        assert torch.equal(ra[4]["episode"]["l"], rb[4]["episode"]["l"])
        for k, v in ra[4].items():  # vandalise every mask of env a
            if k.startswith("_") and isinstance(v, torch.Tensor):
                v.fill_(True) if t % 2 else v.zero_()
    assert a.episode_count == b.episode_count > 0
    a.close(), b.close()


def test_capture_steps_needs_a_gpu_engine(oracle_factory):
    """HipVectorEnv.capture_steps records step() into a HIP graph: refused, with a message, for NumPy output and for the CPU checker backend
    (the GPU behaviour: tests/test_gpu_graph_capture.py)."""
    from gymnasium_amd.gym_api import error

    for kw in ({}, {"output": "torch"}):
        env = gymnasium_amd.make_vec("CartPole-v1", num_envs=4, _engine_factory=oracle_factory, **kw)
        env.reset(seed=0)
        with pytest.raises(error.Error, match="output='torch'"):
            env.capture_steps(actions=np.zeros(4, dtype=np.int64))
        env.close()
