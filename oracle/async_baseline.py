"""TEST INFRASTRUCTURE / CPU BASELINE -- the reference's AsyncVectorEnv ARCHITECTURE restated for the one place the reference cannot run:
the GPU box (no gymnasium there), where BASELINE.json's metric wants "Gymnasium's AsyncVectorEnv on the box's host cores in the same run".

What is restated, and from where (paths relative to the reference tree, gymnasium v1.4.0):

  vector/async_vector_env.py:63-330   one worker PROCESS per sub-environment, a Pipe each, observations in shared memory
                                      (shared_memory=True is the default: workers write their row, the parent reads the batch)
  vector/async_vector_env.py:440-521  step_async: one ("step", action) message per pipe; step_wait: poll, one recv per pipe, np.array()
                                      of the rewards / flags, deepcopy of the shared observation buffer (copy=True)
  vector/async_vector_env.py:773-904  _async_worker: command loop; NEXT_STEP autoreset inside the worker (:829-846): the step after a
                                      finished episode resets, returns reward 0 and not-done
  envs/classic_control/cartpole.py:119-247  the scalar CartPole-v1 the workers step (float64 Euler, sin / cos of NumPy scalars, reward 1.0)
  wrappers/common.py:116-150          TimeLimit (500 steps), folded into the worker
  utils/seeding.py:10-42, core.py:157-159   Generator(PCG64(SeedSequence(seed + i)))
  utils/performance.py:57-103         benchmark_vector_step: the counting rule (NEXT_STEP reset steps are not env steps) and the loop

Pinned: tests/test_async_baseline.py runs this next to the real `gym.make_vec("CartPole-v1", n, "async")` (importable in the build container)
and requires identical observations, rewards and flags; bench.py times it (`cpu_reference`, kind "port") only where gymnasium itself is not
importable.  The product never imports this module.
"""
import math
import multiprocessing as mp
import time

import numpy as np


class ScalarCartPole:
    """cartpole.py:119-247 behind TimeLimit(500): what `gym.make("CartPole-v1")` steps, minus the passive checkers (which compute nothing)."""

    def __init__(self, max_episode_steps=500):
        self.gravity, self.masscart, self.masspole = 9.8, 1.0, 0.1
        self.total_mass = self.masspole + self.masscart
        self.length = 0.5
        self.polemass_length = self.masspole * self.length
        self.force_mag, self.tau = 10.0, 0.02
        self.theta_threshold_radians = 12 * 2 * math.pi / 360
        self.x_threshold = 2.4
        self.max_episode_steps = max_episode_steps
        self.np_random = None
        self.state = None
        self.elapsed = 0

    def reset(self, seed=None):
        if seed is not None:  # core.py:157-159 -> seeding.np_random
            self.np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
        elif self.np_random is None:
            self.np_random = np.random.default_rng()
        self.state = self.np_random.uniform(low=-0.05, high=0.05, size=(4,))
        self.elapsed = 0
        return np.array(self.state, dtype=np.float32), {}

    def step(self, action):
        x, x_dot, theta, theta_dot = self.state
        force = self.force_mag if action == 1 else -self.force_mag
        costheta, sintheta = np.cos(theta), np.sin(theta)
        temp = (force + self.polemass_length * np.square(theta_dot) * sintheta) / self.total_mass
        thetaacc = (self.gravity * sintheta - costheta * temp) / (self.length * (4.0 / 3.0 - self.masspole * np.square(costheta) / self.total_mass))
        xacc = temp - self.polemass_length * thetaacc * costheta / self.total_mass
        x = x + self.tau * x_dot
        x_dot = x_dot + self.tau * xacc
        theta = theta + self.tau * theta_dot
        theta_dot = theta_dot + self.tau * thetaacc
        self.state = np.array((x, x_dot, theta, theta_dot), dtype=np.float64)
        terminated = bool(x < -self.x_threshold or x > self.x_threshold or theta < -self.theta_threshold_radians or theta > self.theta_threshold_radians)
        self.elapsed += 1
        truncated = self.elapsed >= self.max_episode_steps  # TimeLimit.step (wrappers/common.py:129-133)
        return np.array(self.state, dtype=np.float32), 1.0, terminated, truncated, {}


def _worker(index, pipe, parent_pipe, shared, obs_dim):
    """async_vector_env.py:773-904 (shared-memory variant): commands over the pipe, the observation row into shared memory."""
    parent_pipe.close()
    env = ScalarCartPole()
    row = np.frombuffer(shared, dtype=np.float32).reshape(-1, obs_dim)[index]
    autoreset = False
    try:
        while True:
            command, data = pipe.recv()
            if command == "reset":
                obs, info = env.reset(seed=data)
                autoreset = False
                row[:] = obs
                pipe.send(((None, info), True))
            elif command == "step":
                if autoreset:  # NEXT_STEP (:829-834)
                    obs, info = env.reset()
                    reward, terminated, truncated = 0, False, False
                else:
                    obs, reward, terminated, truncated, info = env.step(data)
                autoreset = terminated or truncated
                row[:] = obs
                pipe.send(((None, reward, terminated, truncated, info), True))
            elif command == "close":
                pipe.send((None, True))
                break
    except (KeyboardInterrupt, EOFError):
        pass
    finally:
        pipe.close()


class AsyncCartPoleVectorEnv:
    """`gym.make_vec("CartPole-v1", num_envs, vectorization_mode="async")`, the parent side (async_vector_env.py:187-521)."""

    def __init__(self, num_envs, context=None):
        ctx = mp.get_context(context)
        self.num_envs, self.obs_dim = num_envs, 4
        self.shared = ctx.Array("f", num_envs * self.obs_dim, lock=False)  # create_shared_memory (vector/utils/shared_memory.py)
        self.observations = np.frombuffer(self.shared, dtype=np.float32).reshape(num_envs, self.obs_dim)
        self.parent_pipes, self.processes = [], []
        for i in range(num_envs):
            parent, child = ctx.Pipe()
            p = ctx.Process(target=_worker, args=(i, child, parent, self.shared, self.obs_dim), daemon=True)
            p.start()
            child.close()
            self.parent_pipes.append(parent), self.processes.append(p)
        self._rng = np.random.default_rng(0)

    def reset(self, seed=None):
        for i, pipe in enumerate(self.parent_pipes):  # reset_async: seed + i fan-out (:374-377)
            pipe.send(("reset", None if seed is None else seed + i))
        for pipe in self.parent_pipes:
            pipe.recv()
        return self.observations.copy(), {}

    def step_async(self, actions):
        for pipe, action in zip(self.parent_pipes, actions):  # iterate(action_space, actions): one np.int64 per env
            pipe.send(("step", action))

    def step_wait(self):
        rewards, terminations, truncations = [], [], []
        for pipe in self.parent_pipes:
            (_, reward, terminated, truncated, _), _ = pipe.recv()
            rewards.append(reward), terminations.append(terminated), truncations.append(truncated)
        return self.observations.copy(), np.array(rewards, dtype=np.float64), np.array(terminations, dtype=np.bool_), np.array(truncations, dtype=np.bool_), {}

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        for pipe in self.parent_pipes:
            try:
                pipe.send(("close", None))
                pipe.recv()
            except (BrokenPipeError, EOFError, OSError):
                pass
            pipe.close()
        for p in self.processes:
            p.join(timeout=5)


def benchmark_vector_step(env, target_duration=5.0, seed=0):
    """utils/performance.py:57-103: env-steps/s of `env.step(action_space.sample())`; a NEXT_STEP reset step does not count."""
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))  # action_space.seed(seed)
    sample = lambda: (rng.random(env.num_envs) * 2).astype(np.int64)  # noqa: E731  MultiDiscrete([2] * n).sample()
    env.reset(seed=seed)
    env.step(sample())  # warm-up
    env.reset(seed=seed)
    steps, prev_done = 0, np.zeros(env.num_envs, dtype=np.bool_)
    end = 0.0
    start = time.monotonic()
    while True:
        _, _, terminated, truncated, _ = env.step(sample())
        steps += env.num_envs - int(prev_done.sum())
        prev_done = terminated | truncated
        if time.monotonic() - start > target_duration:
            end = time.monotonic()
            break
    return steps / (end - start)


if __name__ == "__main__":  # python -m oracle.async_baseline <num_envs> <seconds>: prints env-steps/s (bench.py runs it in a clean process: no HIP context to fork)
    import sys

    _env = AsyncCartPoleVectorEnv(int(sys.argv[1]))
    try:
        print(benchmark_vector_step(_env, target_duration=float(sys.argv[2]), seed=0))
    finally:
        _env.close()
