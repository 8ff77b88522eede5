"""-m gpu: the two-role rollout kernel (rollout_duo_kernel: an "env" and an "aux" wavefront per 64 sub-environments, engine.hip) against the one-role
kernel it replaces for the collector's configuration (NEXT_STEP, on-device policy, every output) -- and both against step().

The two kernels run the same arithmetic on different lanes, so everything must be bit-identical: the trajectory, the state and generators left
behind, the running totals.  Cases: batches that do not fill the last workgroup, a number of steps the chunk size does not divide (falls back to the
one-role kernel), episodes so short that a sub-environment needs two resets within one refill period of its reset queue."""
import os

import numpy as np
import pytest

import gymnasium_amd

pytestmark = pytest.mark.gpu


def collect(env_id, duo, num_envs, steps, launches, **kw):
    import torch

    old = os.environ.get("MI355ENV_ROLLOUT_DUO")
    os.environ["MI355ENV_ROLLOUT_DUO"] = "1" if duo else "0"
    try:
        env = gymnasium_amd.make_vec(env_id, num_envs=num_envs, device=0, output="torch", **kw)
        env.reset(seed=7)
        env.action_space.seed(11)
        outs = []
        for _ in range(launches):
            outs.append({k: v.cpu().numpy() for k, v in env.rollout(steps).items()})
        torch.cuda.synchronize()
        state = [np.asarray(x).copy() for x in env.get_state()]
        rng = env.get_rng_state().copy()
        stats = env.statistics()
        nxt = env.action_space.sample()  # the host generator was advanced by the draws the device consumed
        env.close()
        return outs, state, rng, stats, nxt
    finally:
        if old is None:
            os.environ.pop("MI355ENV_ROLLOUT_DUO", None)
        else:
            os.environ["MI355ENV_ROLLOUT_DUO"] = old


@pytest.mark.parametrize("env_id", ["CartPole-v1", "Pendulum-v1", "MountainCar-v0", "MountainCarContinuous-v0"])
@pytest.mark.parametrize("num_envs,steps,kw", [(1000, 128, {}), (256, 36, {}), (777, 8, {"max_episode_steps": 3}), (4096, 64, {"max_episode_steps": 7}),
                                               (300, 30, {}), (65536, 32, {}),
                                               # round 6: the reset queue is fed by the aux role three entries ahead of the resets it knows of; a TimeLimit of one or
                                               # two steps resets faster than that, so the env role computes its entries itself (fifo_entry_from_start)
                                               (513, 64, {"max_episode_steps": 1}), (1300, 128, {"max_episode_steps": 2})])
def test_two_roles_equal_one_role(env_id, num_envs, steps, kw):
    a = collect(env_id, False, num_envs, steps, 3, **kw)
    b = collect(env_id, True, num_envs, steps, 3, **kw)
    for la, lb in zip(a[0], b[0]):
        assert set(la) == set(lb)
        for k in la:
            assert np.array_equal(la[k], lb[k]), (env_id, k)
    for x, y in zip(a[1], b[1]):
        assert np.array_equal(x, y)
    assert np.array_equal(a[2], b[2])
    for k in ("env_steps", "reset_steps", "episodes", "length_sum"):
        assert a[3][k] == b[3][k], k
    assert a[3]["return_sum"] == pytest.approx(b[3]["return_sum"], rel=1e-12)  # (per-workgroup partial sums: same values, the order of the final host sum is the same too)
    assert np.array_equal(a[4], b[4])
    assert a[3]["env_steps"] + a[3]["reset_steps"] == num_envs * steps * 3
    if "max_episode_steps" in kw:
        assert a[3]["episodes"] >= num_envs * (steps * 3 // (kw["max_episode_steps"] + 1) - 1)


@pytest.mark.parametrize("env_id", ["CartPole-v1", "Pendulum-v1", "MountainCarContinuous-v0"])
def test_two_role_rollout_equals_stepping(env_id):
    import torch

    kw = dict(num_envs=500, device=0, output="torch", max_episode_steps=20)
    a, b = gymnasium_amd.make_vec(env_id, **kw), gymnasium_amd.make_vec(env_id, **kw)
    a.reset(seed=5), b.reset(seed=5)
    a.action_space.seed(9), b.action_space.seed(9)
    traj = a.rollout(48)
    for t in range(48):
        act = b.action_space.sample()
        o, r, te, tr, _ = b.step(torch.from_numpy(act).cuda())
        assert np.array_equal(traj["actions"][t].cpu().numpy().reshape(act.shape), act), t
        assert np.array_equal(traj["obs"][t].cpu().numpy(), o.cpu().numpy()), t
        assert np.array_equal(traj["rewards"][t].cpu().numpy(), r.cpu().numpy()), t
        assert np.array_equal(traj["terminations"][t].cpu().numpy(), te.cpu().numpy()) and np.array_equal(traj["truncations"][t].cpu().numpy(), tr.cpu().numpy()), t
    assert a.statistics()["episodes"] == b.statistics()["episodes"] > 0
    a.close(), b.close()
