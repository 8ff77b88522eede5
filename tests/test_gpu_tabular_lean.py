"""The branch-free rollout of a plain transition table (engine.hip tab_rollout_lean_kernel: FrozenLake, CliffWalking, Taxi; NEXT_STEP, on-device policy)
against the CPU oracle stepped with the actions the rollout drew: every observation, reward and flag, then the state rows (state index and the
``prob`` of the last transition), the generators, the episode statistics -- and that stepping on from the rollout's final state continues in lockstep.

Covers what the kernel special-cases: one outcome per (state, action) (Taxi, CliffWalking) and three (slippery maps), a start state that is not state 0
(CliffWalking: 36) and 300 start states (Taxi: the guide table + forward scan), TimeLimits that end episodes every few steps, partly filled
wavefronts / workgroups, T = 1, T odd (the loop is unrolled twice), a rollout without the action array (FULL = false), two rollouts in a row.
Taxi with rain (three outcomes x 3 000 cells do not fit into LDS) exercises the fallback to tab_rollout_kernel through the same checks.

Blackjack-v1 has its own branch-free rollout (bj_rollout_lean_kernel): the same checks for the three rule sets, and once more with every third
lane-step forced through the kernel's general routines (MI355ENV_BJ_FORCE_SLOW: the path a rejected draw or a dealer's seventh card takes).
"""
import numpy as np
import pytest

import gymnasium_amd
import parity_suite as ps

pytestmark = pytest.mark.gpu

KEYS = ["frozenlake", "frozenlake8x8", "cliffwalking", "cliffwalking_slippery", "taxi", "taxi_rainy", "frozenlake_random"]


def _np(x):
    return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)


def _eq(a, b):
    """a: what the HIP engine returned (tensors, a tuple of them, or an (N, 3) tensor for Blackjack's Tuple observation); b: the oracle's NumPy value"""
    if isinstance(b, (tuple, list)):
        if isinstance(a, (tuple, list)):
            return len(a) == len(b) and all(_eq(x, y) for x, y in zip(a, b))
        a = _np(a)
        return a.shape[-1] == len(b) and all(np.array_equal(a[..., j], np.asarray(y)) for j, y in enumerate(b))
    return np.array_equal(_np(a), np.asarray(b))


def _run(key, n, T, oracle_factory, max_episode_steps=None, return_actions=True, rollouts=1):
    import torch

    eid, kw = (key, {}) if isinstance(key, str) and key.startswith("Blackjack") else (key if isinstance(key, tuple) else ps.toytext_spec(key))
    if max_episode_steps is not None:
        kw = dict(kw, max_episode_steps=max_episode_steps)
    gpu = gymnasium_amd.make_vec(eid, num_envs=n, output="torch", **kw)
    cpu = gymnasium_amd.make_vec(eid, num_envs=n, _engine_factory=oracle_factory, **kw)
    og, _ = gpu.reset(seed=11)
    oc, _ = cpu.reset(seed=11)
    assert _eq(og, oc)
    gpu.action_space.seed(3)
    ref = gymnasium_amd.gym_api.batch_space(gpu.single_action_space, n)
    ref.seed(3)
    for r in range(rollouts):
        out = gpu.rollout(T, return_actions=return_actions)
        for t in range(T):
            act = ref.sample()
            if return_actions:
                assert np.array_equal(out["actions"][t].cpu().numpy(), act), f"{key}: action t={t}"
            o, rew, te, tr, _ = cpu.step(act)
            assert _eq(out["obs"][t], o), f"{key}: obs t={t} (rollout {r})"
            assert np.array_equal(out["rewards"][t].cpu().numpy(), rew), f"{key}: reward t={t}"
            assert np.array_equal(out["terminations"][t].cpu().numpy(), te) and np.array_equal(out["truncations"][t].cpu().numpy(), tr), f"{key}: flags t={t}"
        sg, sc = gpu.get_state(), cpu.get_state()
        assert all(np.array_equal(x, y) for x, y in zip(sg, sc)), f"{key}: state rows after rollout {r}"
        assert np.array_equal(gpu.get_rng_state(), cpu.get_rng_state())
        assert gpu.statistics() == cpu.statistics()
    # stepping on from where the rollout stopped: the pending autoresets, the TimeLimit counters and info["prob"] are the oracle's
    for t in range(5):
        act = ref.sample()
        rg, rc = gpu.step(torch.from_numpy(act).cuda()), cpu.step(act)
        for k in range(4):
            assert _eq(rg[k], rc[k]), f"{key}: step {t} after the rollout, output {k}"
        assert set(rg[4]) == set(rc[4])
        for name in rc[4]:
            assert _eq(rg[4][name], rc[4][name]), f"{key}: info[{name}]"
    gpu.close(), cpu.close()


@pytest.mark.parametrize("key", KEYS)
def test_lean_rollout_vs_oracle(key, oracle_factory):
    _run(key, 1000, 40, oracle_factory)


@pytest.mark.parametrize("key", ["frozenlake", "cliffwalking", "taxi"])
@pytest.mark.parametrize("n,T", [(1, 1), (63, 7), (257, 33), (4097, 130)])
def test_lean_rollout_ragged_sizes_and_odd_lengths(key, n, T, oracle_factory):
    _run(key, n, T, oracle_factory)


@pytest.mark.parametrize("key", ["frozenlake", "cliffwalking_slippery", "taxi"])
def test_lean_rollout_short_time_limit(key, oracle_factory):
    _run(key, 300, 31, oracle_factory, max_episode_steps=3)


@pytest.mark.parametrize("key", ["frozenlake8x8", "taxi"])
def test_lean_rollout_without_action_array_and_twice(key, oracle_factory):
    _run(key, 500, 16, oracle_factory, return_actions=False, rollouts=2)


BLACKJACK = [("Blackjack-v1", {}), ("Blackjack-v1", {"natural": True}), ("Blackjack-v1", {"sab": True})]


@pytest.mark.parametrize("spec", BLACKJACK, ids=["plain", "natural", "sab"])
@pytest.mark.parametrize("n,T", [(1000, 60), (65, 7), (4097, 129)])
def test_blackjack_lean_rollout_vs_oracle(spec, n, T, oracle_factory):
    _run(spec, n, T, oracle_factory)


@pytest.mark.parametrize("every", [1, 3])
def test_blackjack_lean_rollout_general_routines(every, oracle_factory, monkeypatch):
    monkeypatch.setenv("MI355ENV_BJ_FORCE_SLOW", str(every))
    _run(("Blackjack-v1", {"natural": True}), 700, 45, oracle_factory, rollouts=2)


def test_blackjack_lean_rollout_time_limit_and_no_action_array(oracle_factory):
    _run(("Blackjack-v1", {}), 300, 30, oracle_factory, max_episode_steps=2, return_actions=False, rollouts=2)
