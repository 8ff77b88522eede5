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

from .. import _native
from ..gym_api import AutoresetMode, error, spaces
from ..gym_api import VectorEnv as VectorEnvBase
from ..vector.hip_vector_env import _SHORT_STEP_OWNERS, HipVectorEnv, _verify_number_and_cast, parse_low_high

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

    # set_state() is this package's own entry (checkpoint / resume; the reference has none), so its domain is stated here: the restated libm
    # sin / cos are glibc's below EXACT_TRIG_RANGE (beyond it glibc switches to Payne-Hanek, docs/classic_kernels.md), and the routines of the
    # environments that wrap or clip their angle leave the test for it out.  _STATE_LIMITS: (columns, largest magnitude accepted).  The limits are
    # far outside anything a trajectory reaches (tests/test_gpu_wide_states.py steps states up to them bit for bit against the oracle).
    EXACT_TRIG_RANGE = 105414336.0
    _STATE_LIMITS: tuple = ()

    def set_state(self, state=None, elapsed_steps=None, flags=None):
        if state is not None:
            arr = np.asarray(state, dtype=np.float64)
            for cols, limit in self._STATE_LIMITS:
                bad = np.abs(arr[:, list(cols)]) > limit  # (NaN compares false: it propagates, as it does in the reference)
                if bad.any():
                    i, j = np.argwhere(bad)[0]
                    raise ValueError(f"{type(self).__name__}.set_state: state[{i}, {cols[j]}] = {arr[i, cols[j]]!r} is outside the accepted range "
                                     f"|x| <= {limit:g} (the exact sin / cos cover |angle| < {self.EXACT_TRIG_RANGE:g})")
        super().set_state(state, elapsed_steps, flags)


class CartPoleVectorEnv(_ClassicControlVectorEnv):
    """``rng="per_env"`` (default): the semantics of ``SyncVectorEnv`` over scalar ``CartPoleEnv`` objects -- sub-environment ``i`` owns the stream
    ``default_rng(seed + i)``, float64 rewards -- which is what the north_star's parity oracle is and what shards across GPUs.

    ``rng="shared"``: the semantics of the reference's own NumPy vector environment of the same name (cartpole.py:353-505; what stock
    ``gymnasium.make_vec("CartPole-v1", n)`` returns, because the id registers it as ``vector_entry_point``): ONE generator for all
    sub-environments -- ``reset(seed=s)`` draws ``uniform(low, high, size=(4, n))`` from ``default_rng(s)``, a step re-draws the ``k``
    sub-environments that finished in the previous step with ``size=(4, k)`` --, float32 rewards, reset bounds that persist for the autoresets,
    no ``reset_mask``, NEXT_STEP only.  Trajectories are ``array_equal`` to that class's (tests/golden/cartpole_vector_entry_point.npz);
    ``register_envs(override_stock_ids=True)`` attaches this mode to the stock id, so that switching changes no seeded trajectory."""

    KIND = "cartpole"
    _STATE_LIMITS = (((2,), 1e8),)  # theta
    DEFAULT_MAX_EPISODE_STEPS = 500
    DEFAULT_RNG = "per_env"

    def __init__(self, num_envs: int = 1, max_episode_steps: int | None = None, sutton_barto_reward: bool = False, rng: str | None = None, **kwargs):
        self._sutton_barto_reward = bool(sutton_barto_reward)
        rng = self.DEFAULT_RNG if rng is None else rng
        if rng not in ("per_env", "shared"):
            raise ValueError(f"rng must be 'per_env' or 'shared', got {rng!r}")
        self._shared_rng = rng == "shared"
        if self._shared_rng:
            mode = kwargs.get("autoreset_mode", AutoresetMode.NEXT_STEP)
            if (mode if isinstance(mode, AutoresetMode) else AutoresetMode(mode)) != AutoresetMode.NEXT_STEP:
                raise error.Error("rng='shared' is CartPoleVectorEnv's semantics (cartpole.py:353-505): NEXT_STEP autoreset only")
            if kwargs.get("env_index_offset", 0):
                raise error.Error("rng='shared': one generator for all sub-environments -- they do not shard across devices (env_index_offset must be 0)")
            self.FUSES_WRAPPERS = False  # (the step epilogue belongs to the per-sub-environment step kernel)
        super().__init__(num_envs=num_envs, max_episode_steps=max_episode_steps, **kwargs)

    def _engine_options(self) -> int:
        return super()._engine_options() | (_native.CFG_SHARED_RNG if self._shared_rng else 0)

    def _short_step_allowed(self) -> bool:
        return not self._shared_rng  # (the shared-generator mode returns float32 rewards: step() below converts)

    # -- rng="shared": VectorEnv.np_random IS the generator the sub-environments draw from (cartpole.py:475, 497) ------------------------
    def _seed_engines(self, seed, mask):
        if not self._shared_rng:
            return super()._seed_engines(seed, mask)
        if seed is None:
            if not self._seeded:  # Env.np_random's lazy OS-entropy seeding
                self._engine.seed(np.tile(_native.pcg_words(super().np_random), (self.num_envs, 1)), None)  # (one generator: the engine reads row 0)
                self._seeded = True
            return
        if not (isinstance(seed, (int, np.integer)) and not isinstance(seed, bool)) or int(seed) < 0:
            raise error.Error(f"Seed must be a python integer, actual type: {type(seed)}" if not isinstance(seed, (int, np.integer)) else
                              f"Seed must be greater or equal to zero, actual value: {seed}")
        # (HipVectorEnv.reset has just seeded the host generator through VectorEnv.reset(seed=...): hand ITS words to the engine, so the two
        # cannot disagree whatever width the seed has)
        self._engine.seed(np.tile(_native.pcg_words(super().np_random), (self.num_envs, 1)), None)  # (one generator: the engine reads row 0)
        self._seeded = True

    @property
    def np_random(self):
        """The generator of ``reset`` / the autoresets.  With rng="shared" the draws happen on the device: reading this property brings the
        host object up to date with them (one small device-to-host copy; synchronises).  In the reference this object IS the generator the
        sub-environments draw from, so a draw the caller takes from it moves the stream for the next reset / autoreset too: once the object has
        been handed out, every engine call of this env first takes over what the caller drew (_shared_push) and afterwards brings the object up to
        date again (_shared_pull) -- the price is one synchronisation per call, paid only by callers that touch ``np_random``."""
        gen = VectorEnvBase.np_random.fget(self)
        if getattr(self, "_shared_rng", False) and getattr(self, "_seeded", False) and getattr(self, "_engine", None) is not None:
            self._shared_pull(gen)
            self._shared_handed_out = True
        return gen

    @np_random.setter
    def np_random(self, value):
        VectorEnvBase.np_random.fset(self, value)
        if getattr(self, "_shared_rng", False) and getattr(self, "_engine", None) is not None:
            self._engine.seed(np.tile(_native.pcg_words(value), (self.num_envs, 1)), None)
            self._seeded = True
            self._shared_words = _native.pcg_words(value)

    def _shared_pull(self, gen=None):
        """host generator <- the device's position"""
        gen = VectorEnvBase.np_random.fget(self) if gen is None else gen
        words = self._engine.get_rng()[0]
        _native.set_pcg_words(gen, words)
        self._shared_words = np.array(words, dtype=np.uint64)

    def _shared_push(self):
        """device <- the host generator, if the caller has drawn from it since the last pull"""
        if not getattr(self, "_shared_handed_out", False):
            return
        words = _native.pcg_words(VectorEnvBase.np_random.fget(self))
        if self.__dict__.get("_shared_words") is None or not np.array_equal(words, self._shared_words):
            self._engine.seed(np.tile(words, (self.num_envs, 1)), None)
            self._shared_words = words

    def reset(self, *, seed=None, options=None):
        if self._shared_rng and options is not None and "reset_mask" in options:
            options = {k: v for k, v in options.items() if k != "reset_mask"}  # CartPoleVectorEnv.reset knows no reset_mask: every sub-environment resets
        if not self._shared_rng:
            return super().reset(seed=seed, options=options)
        if seed is None:
            self._shared_push()
        out = super().reset(seed=seed, options=options)
        if getattr(self, "_shared_handed_out", False):
            self._shared_pull()
        return out

    def rollout(self, *args, **kwargs):
        if not self._shared_rng:
            return super().rollout(*args, **kwargs)
        self._shared_push()
        out = super().rollout(*args, **kwargs)
        if getattr(self, "_shared_handed_out", False):
            self._shared_pull()
        return out

    def step(self, actions):
        if not self._shared_rng:  # (nothing to add: HipVectorEnv.step's short path stays reachable, see _SHORT_STEP_OWNERS)
            return HipVectorEnv.step(self, actions)
        self._shared_push()
        out = super().step(actions)
        if getattr(self, "_shared_handed_out", False):
            self._shared_pull()
        rew = out[1]
        rew = rew.to(self._torch.float32) if self.output == "torch" else rew.astype(np.float32)  # reward arrays of cartpole.py:466-468 are float32
        return (out[0], rew) + tuple(out[2:])

    def _single_spaces(self):
        theta_threshold_radians = 12 * 2 * math.pi / 360
        x_threshold = 2.4
        high = np.array([x_threshold * 2, np.inf, theta_threshold_radians * 2, np.inf], dtype=np.float32)
        return spaces.Box(-high, high, dtype=np.float32), spaces.Discrete(2)

    def _engine_params(self):
        return (1.0 if self._sutton_barto_reward else 0.0,)

    def _parse_reset_options(self, options):
        return parse_low_high(options, -0.05, 0.05)


_SHORT_STEP_OWNERS.add(CartPoleVectorEnv.step)  # (it only adds to the result with rng="shared", and then _short_step_allowed() says no)


class PendulumVectorEnv(_ClassicControlVectorEnv):
    KIND = "pendulum"
    _STATE_LIMITS = (((0,), 1e8),)  # th (never wrapped: pendulum.py:139-150)
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
    _STATE_LIMITS = (((0, 1), 1e6), ((2, 3), 100.0))  # (the velocities too: RK4's intermediate angles grow with their fourth power)
    DEFAULT_MAX_EPISODE_STEPS = 500

    def __init__(self, num_envs: int = 1, max_episode_steps: int | None = None, **kwargs):
        super().__init__(num_envs=num_envs, max_episode_steps=max_episode_steps, **kwargs)

    def _single_spaces(self):
        high = np.array([1.0, 1.0, 1.0, 1.0, 4 * np.pi, 9 * np.pi], dtype=np.float32)
        return spaces.Box(low=-high, high=high, dtype=np.float32), spaces.Discrete(3)

    def _parse_reset_options(self, options):
        return parse_low_high(options, -0.1, 0.1)


class _MountainCarBase(_ClassicControlVectorEnv):
    _STATE_LIMITS = (((0,), 3e7),)  # position: cos(3 * position)

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


class StockCartPoleVectorEnv(CartPoleVectorEnv):
    """What ``register_envs(override_stock_ids=True)`` attaches to the STOCK id ``CartPole-v1``: the engine with the semantics of the class it
    replaces there (the reference's NumPy CartPoleVectorEnv), so that a seeded ``gymnasium.make_vec("CartPole-v1", n)`` user sees the same numbers."""

    DEFAULT_RNG = "shared"


# id -> (creator, max_episode_steps, reward_threshold): gymnasium/envs/__init__.py:26-59
ENV_TABLE = {
    "CartPole-v1": (CartPoleVectorEnv, 500, 475.0),
    "MountainCar-v0": (MountainCarVectorEnv, 200, -110.0),
    "MountainCarContinuous-v0": (MountainCarContinuousVectorEnv, 999, 90.0),
    "Pendulum-v1": (PendulumVectorEnv, 200, None),
    "Acrobot-v1": (AcrobotVectorEnv, 500, -100.0),
}
