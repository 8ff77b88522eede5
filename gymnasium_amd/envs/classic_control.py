"""The five classic-control environments as lockstep MI355X vector environments.

Each class is the ``vector_entry_point`` creator for one id; its constructor accepts the kwargs of the reference's
scalar env (``make_vec`` forwards them verbatim, envs/registration.py:957-963) and describes the same spaces:

  CartPoleVectorEnv               envs/classic_control/cartpole.py:119-162      (CartPole-v1, 500 steps)
  PendulumVectorEnv               envs/classic_control/pendulum.py:102-124      (Pendulum-v1, 200)
  AcrobotVectorEnv                envs/classic_control/acrobot.py:172-184       (Acrobot-v1, 500)
  MountainCarVectorEnv            envs/classic_control/mountain_car.py:108-130  (MountainCar-v0, 200)
  MountainCarContinuousVectorEnv  envs/classic_control/continuous_mountain_car.py:116-148 (MountainCarContinuous-v0, 999)

The dynamics themselves run in gymnasium_amd/csrc (HIP); nothing here computes a step.
"""
from __future__ import annotations

import math

import numpy as np

from ..gym_api import spaces
from ..vector.hip_vector_env import HipVectorEnv, _verify_number_and_cast, parse_low_high

DEFAULT_X = np.pi  # pendulum.py:14-15
DEFAULT_Y = 1.0


class _ClassicControlVectorEnv(HipVectorEnv):
    """``fast_math=False`` (default): sin / cos / ``** 2`` are the reference's libm bit for bit, trajectories are ``array_equal`` to
    ``SyncVectorEnv``'s.  ``fast_math=True`` (MI_CFG_FAST_MATH) trades that for the device's own sin / cos and ``x * x``: results within
    1 ulp per call of the reference's, more env-steps/s."""

    FUSES_WRAPPERS = True

    def __init__(self, *args, fast_math: bool = False, **kwargs):
        self.fast_math = bool(fast_math)
        super().__init__(*args, **kwargs)

    def _engine_options(self) -> int:
        from .. import _native

        return _native.CFG_FAST_MATH if self.fast_math else 0


class CartPoleVectorEnv(_ClassicControlVectorEnv):
    KIND = "cartpole"
    DEFAULT_MAX_EPISODE_STEPS = 500

    def __init__(self, num_envs: int = 1, max_episode_steps: int | None = None, sutton_barto_reward: bool = False, **kwargs):
        self._sutton_barto_reward = bool(sutton_barto_reward)
        super().__init__(num_envs=num_envs, max_episode_steps=max_episode_steps, **kwargs)

    def _single_spaces(self):
        theta_threshold_radians = 12 * 2 * math.pi / 360
        x_threshold = 2.4
        high = np.array([x_threshold * 2, np.inf, theta_threshold_radians * 2, np.inf], dtype=np.float32)
        return spaces.Box(-high, high, dtype=np.float32), spaces.Discrete(2)

    def _engine_params(self):
        return (1.0 if self._sutton_barto_reward else 0.0,)

    def _parse_reset_options(self, options):
        return parse_low_high(options, -0.05, 0.05)


class PendulumVectorEnv(_ClassicControlVectorEnv):
    KIND = "pendulum"
    DEFAULT_MAX_EPISODE_STEPS = 200

    def __init__(self, num_envs: int = 1, max_episode_steps: int | None = None, g: float = 10.0, **kwargs):
        self.g = float(g)
        super().__init__(num_envs=num_envs, max_episode_steps=max_episode_steps, **kwargs)

    def _single_spaces(self):
        high = np.array([1.0, 1.0, 8], dtype=np.float32)
        return (spaces.Box(low=-high, high=high, dtype=np.float32),
                spaces.Box(low=-2.0, high=2.0, shape=(1,), dtype=np.float32))

    def _engine_params(self):
        return (self.g,)

    def _parse_reset_options(self, options):
        if options is None:
            return None
        x = _verify_number_and_cast(options.get("x_init") if "x_init" in options else DEFAULT_X)
        y = _verify_number_and_cast(options.get("y_init") if "y_init" in options else DEFAULT_Y)
        return (x, y)


class AcrobotVectorEnv(_ClassicControlVectorEnv):
    KIND = "acrobot"
    DEFAULT_MAX_EPISODE_STEPS = 500

    def __init__(self, num_envs: int = 1, max_episode_steps: int | None = None, **kwargs):
        super().__init__(num_envs=num_envs, max_episode_steps=max_episode_steps, **kwargs)

    def _single_spaces(self):
        high = np.array([1.0, 1.0, 1.0, 1.0, 4 * np.pi, 9 * np.pi], dtype=np.float32)
        return spaces.Box(low=-high, high=high, dtype=np.float32), spaces.Discrete(3)

    def _parse_reset_options(self, options):
        return parse_low_high(options, -0.1, 0.1)


class _MountainCarBase(_ClassicControlVectorEnv):
    def __init__(self, num_envs: int = 1, max_episode_steps: int | None = None, goal_velocity: float = 0, **kwargs):
        self.goal_velocity = goal_velocity
        super().__init__(num_envs=num_envs, max_episode_steps=max_episode_steps, **kwargs)

    def _obs_space(self):
        low = np.array([-1.2, -0.07], dtype=np.float32)
        high = np.array([0.6, 0.07], dtype=np.float32)
        return spaces.Box(low, high, dtype=np.float32)

    def _engine_params(self):
        return (float(self.goal_velocity),)

    def _parse_reset_options(self, options):
        return parse_low_high(options, -0.6, -0.4)


class MountainCarVectorEnv(_MountainCarBase):
    KIND = "mountain_car"
    DEFAULT_MAX_EPISODE_STEPS = 200

    def _single_spaces(self):
        return self._obs_space(), spaces.Discrete(3)


class MountainCarContinuousVectorEnv(_MountainCarBase):
    KIND = "mountain_car_continuous"
    DEFAULT_MAX_EPISODE_STEPS = 999

    def _single_spaces(self):
        return self._obs_space(), spaces.Box(low=-1.0, high=1.0, shape=(1,), dtype=np.float32)


# id -> (creator, max_episode_steps, reward_threshold): gymnasium/envs/__init__.py:26-59
ENV_TABLE = {
    "CartPole-v1": (CartPoleVectorEnv, 500, 475.0),
    "MountainCar-v0": (MountainCarVectorEnv, 200, -110.0),
    "MountainCarContinuous-v0": (MountainCarContinuousVectorEnv, 999, 90.0),
    "Pendulum-v1": (PendulumVectorEnv, 200, None),
    "Acrobot-v1": (AcrobotVectorEnv, 500, -100.0),
}
