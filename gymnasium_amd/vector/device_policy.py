"""``action_space.sample()`` of a HipVectorEnv served by the engine's action stream.

The metric's own loop is ``env.step(env.action_space.sample())`` (gymnasium/utils/performance.py:82-97).  In the reference
``sample()`` is one NumPy call on the batched space's generator -- ``(np_random.random(nvec.shape) * nvec).astype(dtype)``
(spaces/multi_discrete.py:176-178) or ``np_random.uniform(low, high, size)`` (spaces/box.py:463-465) -- which at 65 536
sub-environments costs more host time than the step costs the GPU.  The engine restates that generator (PCG64 with skip-ahead,
``mi_action_seed`` / ``mi_action_sample``), so the batched action space of a HipVectorEnv hands out the SAME draws from the device:

* ``sample()`` returns the next batch of a block the engine drew ahead in one launch (``mi_action_sample``: ``K`` batches); with
  ``output="torch"`` the batch is a device tensor and the loop above enqueues nothing but step kernels, with ``output="numpy"`` it is a
  NumPy view of one pinned device-to-host copy per ``K`` steps.  A fresh block is allocated per refill, so a returned batch is never
  overwritten, exactly like the reference's fresh arrays.
* ``np_random`` stays the space's NumPy generator: reading it first returns the draws that were made ahead but not handed out
  (``mi_action_skip``) and moves the generator to the stream's position (``mi_action_get``), so mixing ``sample()``, ``np_random.random()``,
  ``env.rollout()`` and ``env.step(None)`` consumes ONE stream in call order, as in the reference.  ``seed()`` behaves as always.
* masks / probabilities, spaces the sampler does not cover (unbounded Box, MultiDiscrete with a start) and detached spaces (pickled,
  deep-copied, the env closed) take the reference's NumPy path.

Bit-equality with the NumPy sampler is pinned by tests/test_device_policy.py (host) and tests/test_gpu_device_policy.py (GPU).
"""
from __future__ import annotations

import weakref

import numpy as np

from .. import _native
from ..gym_api import spaces

RING_BYTES = 32 << 20  # draw-ahead per refill (at 65 536 CartPoles: 64 batches of 512 KB)
RING_MAX = 256


class _DevicePolicyMixin:
    """Mixed into the batched MultiDiscrete / Box of a HipVectorEnv (``attach``)."""

    __slots__ = ()

    # -- which side holds the stream's position ------------------------------------------------------------
    def _hip_engine(self):
        ref = self.__dict__.get("_hip_env")
        env = ref() if ref is not None else None
        if env is None or getattr(env, "_engine", None) is None or not getattr(env, "_device_policy", False):
            return None, None
        return env, env._engine

    def _hip_drop_ahead(self, eng):
        """Give back the batches that were drawn ahead and not handed out."""
        left = len(self.__dict__.get("_hip_ring", ())) - self.__dict__.get("_hip_pos", 0)
        self._hip_ring, self._hip_pos = (), 0
        if left > 0 and eng is not None:
            eng.action_skip(-left * self._hip_batch_draws)

    def _hip_to_host(self):
        """The NumPy generator becomes the stream's position (no-op while it already is)."""
        if not self.__dict__.get("_hip_on_engine", False):
            return
        self._hip_on_engine = False
        _, eng = self._hip_engine()
        if eng is None:  # the env is gone: whatever the generator holds is all there is
            self._hip_ring, self._hip_pos = (), 0
            return
        self._hip_drop_ahead(eng)
        _native.set_pcg_words(self._np_random, eng.action_get())

    def _hip_to_engine(self, eng):
        """The engine's action stream becomes the position (no-op while it already is)."""
        if self.__dict__.get("_hip_on_engine", False):
            return
        eng.action_seed(_native.pcg_words(super().np_random))
        self._hip_on_engine = True

    def hip_use_stream(self):
        """For the env's own consumers of the stream (rollout(), step(None), capture): position on the engine, nothing drawn ahead.
        Returns the engine, or None when the space is detached."""
        _, eng = self._hip_engine()
        if eng is not None:
            self._hip_to_engine(eng)
            self._hip_drop_ahead(eng)
        return eng

    # -- the Space interface ------------------------------------------------------------------------------------
    @property
    def np_random(self):
        self._hip_to_host()
        return super().np_random

    def seed(self, seed=None):
        self._hip_on_engine = False  # (whatever was drawn ahead belongs to the old stream)
        self._hip_ring, self._hip_pos = (), 0
        return super().seed(seed)

    def sample(self, mask=None, probability=None):
        d = self.__dict__
        if mask is None and probability is None:  # the common call: the next batch of the block drawn ahead (a non-empty block implies an attached engine)
            pos, ring = d.get("_hip_pos", 0), d.get("_hip_ring", ())
            if pos < len(ring):
                d["_hip_pos"] = pos + 1
                return ring[pos]
        env, eng = self._hip_engine()
        if eng is None or mask is not None or probability is not None:
            return super().sample(mask, probability)
        pos = self.__dict__.get("_hip_pos", 0)
        ring = self.__dict__.get("_hip_ring", ())
        if pos >= len(ring):
            self._hip_to_engine(eng)
            ring = self._hip_ring = env._draw_action_batches(self._hip_ring_steps)
            pos = 0
        self._hip_pos = pos + 1
        return ring[pos]

    def __getstate__(self):
        self._hip_to_host()
        return {k: v for k, v in self.__dict__.items() if not k.startswith("_hip_")}

    def __deepcopy__(self, memo):
        import copy

        self._hip_to_host()
        new = self.__class__.__new__(self.__class__)
        for k, v in self.__dict__.items():
            if not k.startswith("_hip_"):
                setattr(new, k, copy.deepcopy(v, memo))
        return new


class HipMultiDiscrete(_DevicePolicyMixin, spaces.MultiDiscrete):
    pass


class HipBox(_DevicePolicyMixin, spaces.Box):
    pass


def attach(space, env, act_dim: int):
    """Turn the batched action space of ``env`` into its device-sampled subclass -- when the engine's sampler covers it: a MultiDiscrete of
    equal counts starting at 0 (a batched Discrete) or a fully bounded float32 Box (the classic and MuJoCo action spaces).  Returns the space."""
    if isinstance(space, spaces.MultiDiscrete):
        start = getattr(space, "start", None)
        ok = space.dtype == np.int64 and np.all(space.nvec == space.nvec.flat[0]) and (start is None or not np.any(start))
        cls = HipMultiDiscrete
    elif isinstance(space, spaces.Box):
        ok = space.dtype == np.float32 and bool(np.all(space.bounded_below) and np.all(space.bounded_above))
        cls = HipBox
    else:
        return space
    if not ok:
        return space
    new = cls.__new__(cls)  # the same space (nvec / bounds / dtype / generator) as an instance of the device-sampled subclass
    new.__dict__.update(space.__dict__)
    space = new
    space._hip_env = weakref.ref(env)
    space._hip_on_engine = False
    space._hip_ring, space._hip_pos = (), 0
    space._hip_batch_draws = env.num_envs * act_dim
    bytes_per_batch = env.num_envs * act_dim * (8 if cls is HipMultiDiscrete else 4)
    space._hip_ring_steps = int(max(1, min(RING_MAX, RING_BYTES // bytes_per_batch)))
    return space
