"""bench.py's known answer: the first timed launch at BASELINE.json configs[1]'s exact shape (CartPole-v1, 65 536 sub-environments, 128 fused
steps, reset(seed=0), policy stream seeded 0).  tests/golden/bench_digest.json holds the sha256 of that trajectory as the REFERENCE computes it
(tests/golden/make_bench_digest.py: gymnasium's own SyncVectorEnv over 65 536 scalar CartPoleEnv objects); here the oracle has to reproduce it,
byte for byte, on the CPU -- and the -m gpu contract test requires the same digest from the rollout kernel (`output_sha256` of the bench line)."""
import json
import os

import numpy as np
import pytest

import bench
from conftest import GOLDEN

KEY = "CartPole-v1:65536:128:rank0"


def _golden():
    return json.load(open(os.path.join(GOLDEN, "bench_digest.json")))


def test_oracle_reproduces_the_reference_digest_at_configs1_shape():
    g = _golden()
    traj = bench.oracle_trajectory("CartPole-v1", 65536, 128, offset=0, policy_seed=0)
    acts, obs, rew, te, tr = traj
    assert acts.dtype == np.int64 and obs.dtype == np.float32 and obs.shape == (128, 65536, 4) and rew.dtype == np.float64 and te.dtype == np.bool_
    assert int((te | tr).sum()) == g["episodes_finished"] and float(rew.sum()) == g["reward_sum"]
    assert bench.trajectory_digest(tuple(np.ascontiguousarray(x[:, :1024]) for x in traj)) == g[KEY + ":first_1024_envs"]
    assert bench.trajectory_digest(tuple(np.ascontiguousarray(x[:, ::64]) for x in traj)) == g[KEY + ":every_64th_env"]
    assert bench.trajectory_digest(traj) == g[KEY]


@pytest.mark.parametrize("env_id", ["Pendulum-v1", "Acrobot-v1", "MountainCarContinuous-v0", "MountainCar-v0", "FrozenLake-v1", "FrozenLake8x8-v1", "CliffWalking-v1", "Taxi-v4", "Blackjack-v1"])
def test_oracle_reproduces_the_reference_digest_at_configs2_shape(env_id):
    """BASELINE.json configs[2] at its exact shape (65 536 sub-environments, 128 steps): the digest of the reference's own SyncVectorEnv rollout
    (tests/golden/make_bench_digest.py <id>) from the oracle -- whole episodes of Pendulum (200-step TimeLimit not reached: 128 steps), the chaotic Acrobot
    and MountainCarContinuous's float32 / float64 mixed arithmetic, every byte of 8.4 M env-steps each -- and, at the same shape, two ToyText kinds
    (bit-exact integer kernels; Blackjack's Tuple observation as three int64 columns; Taxi's reference rollout was computed 2 048 sub-environments at a time:
    65 536 scalar Taxi envs with their own 500 x 6 transition dicts do not fit in memory at once, tests/golden/make_bench_digest.py run_in_chunks)."""
    g = json.load(open(os.path.join(GOLDEN, "bench_digest_configs2.json")))
    key = f"{env_id}:65536:128:rank0"
    if key not in g:
        pytest.skip(f"{key} not generated yet (tests/golden/make_bench_digest.py {env_id})")
    traj = bench.oracle_trajectory(env_id, 65536, 128, offset=0, policy_seed=0)
    assert bench.trajectory_digest(tuple(np.ascontiguousarray(x[:, ::64]) for x in traj)) == g[key + ":every_64th_env"]
    assert bench.trajectory_digest(traj) == g[key]


def test_teacher_forced_subset_equals_the_policy_rollout():
    """oracle_check's two halves agree with each other: a strided subset, seeded by global index and teacher-forced with the policy rollout's actions,
    is the same trajectory (what the MuJoCo kinds' in-run check relies on)."""
    N, T = 512, 16
    acts, obs, rew, te, tr = bench.oracle_trajectory("Pendulum-v1", N, T, offset=1000, policy_seed=3)
    idx = np.arange(0, N, 8)
    a2, o2, r2, te2, tr2 = bench.oracle_trajectory("Pendulum-v1", N, T, offset=1000, actions=np.ascontiguousarray(acts[:, idx]), env_indices=idx)
    assert np.array_equal(o2, obs[:, idx]) and np.array_equal(r2, rew[:, idx]) and np.array_equal(te2, te[:, idx]) and np.array_equal(tr2, tr[:, idx])


def test_line_fits_and_is_strict_json():
    big = {"metric": bench.METRIC, "value": 1.0, "secondary": {str(i): "x" * 100 for i in range(100)}, "devices": [{"uuid": "u" * 40}] * 8}
    line = bench.fit_line(dict(big))
    assert len(line) < bench.LINE_LIMIT and "secondary" not in json.loads(line)
    try:
        bench.fit_line({"value": float("nan")})
        raise AssertionError("NaN must not be serialised")
    except ValueError:
        pass
