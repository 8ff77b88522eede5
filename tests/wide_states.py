"""States no trajectory from reset() reaches, for the teacher-forced parity cases: tests/golden/make_golden.py steps the REFERENCE from them
(teacher_wide_<env>.npz), tests/test_oracle_golden.py the oracle, tests/test_gpu_wide_states.py the HIP engine (step() and both rollout kernels)."""
import numpy as np

EDGE = np.frombuffer(np.array([0x3FEB6000 << 32], dtype=np.uint64).tobytes(), dtype=np.float64)[0]  # in_main_range()'s bound: 0.85546875


def wide_states(n, seed):
    rng = np.random.default_rng(seed)
    s = np.empty((n, 4), dtype=np.float64)
    s[:, 0] = rng.uniform(-2.3, 2.3, n)
    s[:, 1] = rng.uniform(-30.0, 30.0, n)
    kind = rng.integers(0, 6, n)
    theta = np.where(kind == 0, rng.uniform(-0.2, 0.2, n), 0.0)  # the common lanes, interleaved with the rare ones inside every wavefront
    theta = np.where(kind == 1, rng.uniform(-0.9, 0.9, n), theta)
    theta = np.where(kind == 2, rng.uniform(-10.0, 10.0, n), theta)
    theta = np.where(kind == 3, rng.uniform(-1.0, 1.0, n) * 1e5, theta)
    around = np.nextafter(EDGE, np.where(rng.integers(0, 2, n) == 0, 0.0, 1.0)) * np.where(rng.integers(0, 2, n) == 0, -1.0, 1.0)
    theta = np.where(kind == 4, np.where(rng.integers(0, 3, n) == 0, EDGE, around), theta)
    theta = np.where(kind == 5, rng.uniform(-np.pi, np.pi, n), theta)
    s[:, 2] = theta
    mag = rng.integers(0, 4, n)
    s[:, 3] = rng.uniform(-1.0, 1.0, n) * np.choose(mag, [5.0, 1e3, 1e8, 1e-300])  # (theta_dot ** 2 carries the divisions' operands out of their range)
    return s


def wide_pendulum_states(n, seed):
    """Angles far outside what 200 steps can reach (|theta| grows by at most 0.4 per step): fmod(theta + pi, 2 pi) of pendulum.py:262 by the rounded
    reciprocal (sincos_exact.h fmod_const) has its quotient off by one exactly around the multiples of 2 pi -- those, from both sides, are a third of the lanes."""
    rng = np.random.default_rng(seed)
    s = np.empty((n, 2), dtype=np.float64)
    kind = rng.integers(0, 6, n)
    k = rng.integers(-10**6, 10**6, n).astype(np.float64)
    near = k * (2 * np.pi) - np.pi  # theta + pi lands on a multiple of 2 pi, give or take the roundings
    for _ in range(3):
        near = np.where(rng.integers(0, 2, n) == 0, near, np.nextafter(near, np.where(rng.integers(0, 2, n) == 0, -np.inf, np.inf)))
    th = np.where(kind == 0, rng.uniform(-np.pi, np.pi, n), 0.0)
    th = np.where(kind == 1, rng.uniform(-100.0, 100.0, n), th)
    th = np.where(kind == 2, rng.uniform(-1.0, 1.0, n) * 1e6, th)
    th = np.where(kind == 3, rng.uniform(-1.0, 1.0, n) * 1e8, th)  # (the restated sin / cos are glibc's below 1.05e8: docs/classic_kernels.md; beyond it the kernels defer to ocml)
    th = np.where(kind >= 4, near, th)
    s[:, 0] = th
    s[:, 1] = rng.uniform(-8.0, 8.0, n)
    return s


def wide_acrobot_states(n, seed):
    """acrobot.py:244-279: angles the wrap() loop has to bring back over up to 16 turns, velocities far outside bound()'s limits (4 pi, 9 pi), into RK4."""
    rng = np.random.default_rng(seed)
    s = np.empty((n, 4), dtype=np.float64)
    big = rng.integers(0, 3, (n, 2))
    s[:, :2] = rng.uniform(-1.0, 1.0, (n, 2)) * np.choose(big, [np.pi, 10.0, 100.0])
    s[:, 2:] = rng.uniform(-1.0, 1.0, (n, 2)) * np.choose(rng.integers(0, 3, (n, 2)), [4 * np.pi, 30.0, 1e-200])
    return s


def wide_mountaincar_states(n, seed):
    """mountain_car.py:137-160 / continuous_mountain_car.py:150-178: positions and velocities far outside the clips, the goal line and the wall from both sides."""
    rng = np.random.default_rng(seed)
    s = np.empty((n, 2), dtype=np.float64)
    kind = rng.integers(0, 5, n)
    p = np.where(kind == 0, rng.uniform(-1.2, 0.6, n), 0.0)
    p = np.where(kind == 1, rng.uniform(-1e3, 1e3, n), p)
    p = np.where(kind == 2, rng.uniform(-1.0, 1.0, n) * 1e7, p)
    p = np.where(kind == 3, np.nextafter(-1.2, rng.choice([-2.0, 0.0], n)), p)
    p = np.where(kind == 4, np.nextafter(rng.choice([0.5, 0.45, 0.6], n), rng.choice([-1.0, 1.0], n)), p)
    s[:, 0] = p
    s[:, 1] = rng.uniform(-1.0, 1.0, n) * np.choose(rng.integers(0, 4, n), [0.07, 1.0, 1e-3, 1e-300])
    return s
