"""AutoresetMode and the VectorEnv base class (mirror of gymnasium/vector/vector_env.py:34-351)."""
from __future__ import annotations

from enum import Enum
from typing import Any

import numpy as np

from . import seeding


class AutoresetMode(Enum):
    """When a finished sub-environment is reset (vector_env.py:34-39)."""

    NEXT_STEP = "NextStep"
    SAME_STEP = "SameStep"
    DISABLED = "Disabled"


class VectorEnv:
    """Base class of vectorised environments: batched reset/step over ``num_envs`` sub-environments."""

    metadata: dict[str, Any] = {}
    spec = None
    render_mode = None
    closed = False

    observation_space = None
    action_space = None
    single_observation_space = None
    single_action_space = None
    num_envs: int

    _np_random = None
    _np_random_seed = None

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self._np_random, self._np_random_seed = seeding.np_random(seed)

    def step(self, actions):
        raise NotImplementedError(f"{self.__str__()} step function is not implemented.")

    def render(self):
        raise NotImplementedError(f"{self.__str__()} render function is not implemented.")

    def close(self, **kwargs):
        if self.closed:
            return
        self.close_extras(**kwargs)
        self.closed = True

    def close_extras(self, **kwargs):
        pass

    @property
    def np_random(self):
        if self._np_random is None:
            self._np_random, self._np_random_seed = seeding.np_random()
        return self._np_random

    @np_random.setter
    def np_random(self, value):
        self._np_random = value
        self._np_random_seed = -1

    @property
    def np_random_seed(self):
        if self._np_random_seed is None:
            self._np_random, self._np_random_seed = seeding.np_random()
        return self._np_random_seed

    @property
    def unwrapped(self):
        return self

    def _add_info(self, vector_infos, env_info, env_num):
        """Merge one sub-env's info dict into the batched dict with ``_key`` presence masks (vector_env.py:277-338)."""
        for key, value in env_info.items():
            if key == "final_obs":
                slot = vector_infos.get("final_obs")
                if slot is None:
                    slot = np.full(self.num_envs, None, dtype=object)
                slot[env_num] = value
            elif isinstance(value, dict):
                slot = self._add_info(vector_infos.get(key, {}), value, env_num)
            else:
                slot = vector_infos.get(key)
                if slot is None:
                    if type(value) in (int, float, bool) or issubclass(type(value), np.number):
                        slot = np.zeros(self.num_envs, dtype=type(value))
                    elif isinstance(value, np.ndarray):
                        slot = np.zeros((self.num_envs, *value.shape), dtype=value.dtype)
                    else:
                        slot = np.full(self.num_envs, None, dtype=object)
                slot[env_num] = value
            mask = vector_infos.get(f"_{key}")
            if mask is None:
                mask = np.zeros(self.num_envs, dtype=np.bool_)
            mask[env_num] = True
            vector_infos[key], vector_infos[f"_{key}"] = slot, mask
        return vector_infos

    def __del__(self):
        if not getattr(self, "closed", True):
            self.close()

    def __repr__(self):
        if self.spec is None:
            return f"{self.__class__.__name__}(num_envs={self.num_envs})"
        return f"{self.__class__.__name__}({self.spec.id}, num_envs={self.num_envs})"
