"""-m gpu: step() captured into a HIP graph (HipVectorEnv.capture_steps) replays the very trajectory eager stepping produces.

The per-step API is launch-bound at the benchmark's batch (a CartPole step kernel runs ~3 us); a captured graph replays K steps -- and the
policy between them -- with one launch.  What must hold: bit-identical observations, rewards, flags and infos, with the device-side
bookkeeping (pending autoresets, episode statistics) carried from replay to replay and into later eager steps."""
import numpy as np
import pytest

import gymnasium_amd
from gymnasium_amd.gym_api import error

pytestmark = pytest.mark.gpu


def host(x):
    return x.cpu().numpy()


def same_infos(ia, ib, skip=("t",)):
    assert set(ia) == set(ib)
    for k in ia:
        if isinstance(ia[k], dict):
            same_infos(ia[k], ib[k], skip)
        elif k not in skip:
            assert np.array_equal(host(ia[k]), host(ib[k])), k


@pytest.mark.parametrize("env_id,mode,kw", [("CartPole-v1", "NextStep", {}), ("CartPole-v1", "SameStep", {"record_episode_statistics": True}),
                                            ("Pendulum-v1", "NextStep", {"record_episode_statistics": True}),
                                            ("Ant-v5", "NextStep", {}), ("Hopper-v5", "SameStep", {"record_episode_statistics": True})])
def test_replayed_single_step_equals_eager_stepping(env_id, mode, kw):
    import torch

    common = dict(num_envs=256, device=0, output="torch", autoreset_mode=mode, max_episode_steps=25, **kw)
    a, b = gymnasium_amd.make_vec(env_id, **common), gymnasium_amd.make_vec(env_id, **common)
    a.reset(seed=3), b.reset(seed=3)
    a.action_space.seed(5)
    acts = [torch.from_numpy(a.action_space.sample()).cuda() for _ in range(60)]
    for k in range(2):  # kernels load on first use
        ra, rb = a.step(acts[k]), b.step(acts[k])
    slot = acts[2].clone()
    g = b.capture_steps(actions=slot)
    for k in range(2, 50):
        slot.copy_(acts[k])
        ra, rb = a.step(acts[k]), g.replay()
        for x, y in zip(ra[:4], rb[:4]):
            assert np.array_equal(host(x), host(y)), (env_id, k)
        same_infos(ra[4], rb[4])
    for k in range(50, 60):  # and eager steps continue from where the replays left the env
        ra, rb = a.step(acts[k]), b.step(acts[k])
        for x, y in zip(ra[:4], rb[:4]):
            assert np.array_equal(host(x), host(y)), (env_id, k)
        same_infos(ra[4], rb[4])
    assert a.statistics() == b.statistics() and a.statistics()["episodes"] > 0
    assert a.episode_count == b.episode_count
    a.close(), b.close()


def test_policy_and_eight_steps_in_one_graph():
    import torch

    def policy(obs):  # push the cart AWAY from the side the pole leans to: episodes end (and reset, inside the graph) every dozen steps
        return (obs[:, 2] + 0.5 * obs[:, 3] < 0).to(torch.int64)

    common = dict(num_envs=4096, device=0, output="torch", copy=True)
    a, b = gymnasium_amd.make_vec("CartPole-v1", **common), gymnasium_amd.make_vec("CartPole-v1", **common)
    oa, _ = a.reset(seed=11)
    ob, _ = b.reset(seed=11)
    oa, ob = a.step(policy(oa))[0], b.step(policy(ob))[0]
    g = b.capture_steps(policy=policy, steps=8)
    assert len(g.results) == 8
    for rep in range(40):
        eager = []
        for _ in range(8):
            r = a.step(policy(oa))
            oa = r[0]
            eager.append(r)
        last = g.replay()
        assert last is g.results[-1]
        for k in range(8):  # copy=True: every captured step kept its own outputs
            for x, y in zip(eager[k][:4], g.results[k][:4]):
                assert np.array_equal(host(x), host(y)), (rep, k)
    st = a.statistics()
    assert st == b.statistics() and st["env_steps"] + st["reset_steps"] == 4096 * (1 + 8 * 40) and st["episodes"] > 0
    a.close(), b.close()


def test_what_cannot_be_captured_says_so():
    import torch

    env = gymnasium_amd.make_vec("CartPole-v1", num_envs=8, device=0)
    env.reset(seed=0)
    with pytest.raises(error.Error, match="output='torch'"):
        env.capture_steps(actions=np.zeros(8, dtype=np.int64))
    env.close()
    env = gymnasium_amd.make_vec("CartPole-v1", num_envs=8, device=0, output="torch", strict_actions=True)
    env.reset(seed=0)
    with pytest.raises(error.Error, match="strict_actions"):
        env.capture_steps(actions=torch.zeros(8, dtype=torch.int64, device="cuda"))
    env.close()
    env = gymnasium_amd.make_vec("CartPole-v1", num_envs=8, device=0, output="torch")
    with pytest.raises(AssertionError):
        env.capture_steps(actions=torch.zeros(8, dtype=torch.int64, device="cuda"))
    env.reset(seed=0)
    with pytest.raises(ValueError, match="in place"):
        env.capture_steps(actions=torch.zeros(8, dtype=torch.int32, device="cuda"))
    with pytest.raises(ValueError, match="either"):
        env.capture_steps()
    env.close()
