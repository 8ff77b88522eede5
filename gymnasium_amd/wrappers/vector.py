"""gymnasium.wrappers.vector.{NormalizeObservation, NormalizeReward, ClipReward} with the batch kept in HBM.

Same constructor arguments, attributes (`obs_rms`, `return_rms`, `epsilon`, `gamma`, `update_running_mean`) and errors as

  gymnasium/wrappers/vector/stateful_observation.py:27-160   NormalizeObservation
  gymnasium/wrappers/vector/stateful_reward.py:21-183        NormalizeReward
  gymnasium/wrappers/vector/vectorize_reward.py:115-151      ClipReward
  gymnasium/wrappers/utils.py:33-71                          RunningMeanStd

but the arithmetic runs in libmi355env.so on the GPU (the wrappers have no CPU implementation -- the NumPy restatement in
oracle/wrappers.py is test infrastructure), in one of two forms:

* FUSED (classic-control HipVectorEnv directly underneath, possibly through RecordEpisodeStatistics / NumpyToTorch): the wrapper registers
  itself with the env and its arithmetic becomes the output stage of the step kernel (mi_set_step_epilogue, csrc/engine.hip): the batch
  statistics are gathered by the step kernel's workgroups, one small second launch normalises in place -- NumPy callers get the wrapped
  batch with the step's one device-to-host copy.  The fusion is SCOPED to the call: a ``step()`` entered through fused wrapper k runs the
  arithmetic of the fused wrappers up to k; stepping the env itself (``w.env.step``, ``w.unwrapped.step``) or an inner wrapper returns that
  object's own values -- raw for the env -- and leaves the outer wrappers' statistics alone, as in the reference.  Settings (``gamma``,
  ``epsilon``, ``min_reward``, ``max_reward``, ``update_running_mean``) are read at every step.
* STAND-ALONE (any other env of this package, or a wrapper order the epilogue cannot express): passes of csrc/wrappers.hip over the arrays
  the engine produced; with ``output="torch"`` nothing leaves the GPU, NumPy batches are staged through the device.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _native
from ..gym_api import AutoresetMode, batch_space, error, spaces


def _torch():
    import torch

    return torch


class RunningMeanStd:
    """Device-resident mean / var / count (wrappers/utils.py:33-71); `.mean`, `.var`, `.count` read them back as NumPy."""

    def __init__(self, epsilon=1e-4, shape=(), dtype=np.float64, device=0):
        self._lib = _native.load_library()
        self.shape = tuple(shape)
        self.dim = int(np.prod(self.shape)) if self.shape else 1
        self.dtype = np.dtype(dtype)
        self.device = int(device)
        self._h = C.c_void_p()
        code = _native.MI_F32 if self.dtype == np.float32 else _native.MI_F64
        self._lib.check(self._lib.rms_create(self.device, self.dim, code, float(epsilon), C.byref(self._h)))

    def _get(self):
        mean, var, count = np.zeros(self.dim), np.zeros(self.dim), np.zeros(1)
        self._lib.check(self._lib.rms_get(self._h, _stream(), mean.ctypes.data, var.ctypes.data, count.ctypes.data))
        return mean, var, float(count[0])

    @property
    def mean(self):
        return self._get()[0].reshape(self.shape)

    @property
    def var(self):
        return self._get()[1].reshape(self.shape)

    @property
    def count(self):
        return self._get()[2]

    def set(self, mean=None, var=None, count=None):
        arrs = [None if a is None else np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), self.shape or (1,)).reshape(-1))
                for a in (mean, var)]
        cnt = None if count is None else np.array([float(count)])
        self._lib.check(self._lib.rms_set(self._h, _stream(), *[None if a is None else a.ctypes.data for a in arrs],
                                          None if cnt is None else cnt.ctypes.data))

    def close(self):
        if self._h:
            self._lib.rms_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _stream():
    torch = _torch()
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _base_env(env):
    while isinstance(env, VectorWrapper):
        env = env.env
    return env


def _fusion_base(env):
    """The HipVectorEnv under ``env`` if it fuses wrappers and everything in between passes observations and rewards through unchanged (or
    is itself part of the fused unit); None otherwise."""
    e = env
    while isinstance(e, VectorWrapper):
        if not (e._transparent or e._fused):
            return None
        e = e.env
    return e if getattr(e, "_can_fuse", None) is not None and e._can_fuse() else None


def _close_fusion(env):
    """A stand-alone wrapper now sits on top of ``env``: nothing above it may join the fused unit underneath."""
    base = _base_env(env)
    if getattr(base, "_fusion_state", None) is not None and getattr(base, "FUSES_WRAPPERS", False):
        base._fusion_state()["closed"] = True


class VectorWrapper:
    """Minimal gymnasium.vector.VectorWrapper: forwards everything to the wrapped vector env (vector_env.py:341-470)."""

    _transparent = False  # True: observations and rewards pass through unchanged (a fused unit may extend across this wrapper)
    _fused = False        # True: this wrapper's arithmetic runs inside the step kernel of the HipVectorEnv underneath

    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def reset(self, *, seed=None, options=None):
        return self.env.reset(seed=seed, options=options)

    def step(self, actions):
        return self.env.step(actions)

    def _step_fused(self, actions):
        """step() of a fused wrapper: mark this call as entered through `self` (the outermost fused wrapper of a call wins) and run the
        chain underneath -- the engine's step applies the epilogue of the members up to the entry (HipVectorEnv._sync_epilogue)."""
        st = self._base._fusion_state()
        outermost = st["entry"] is None
        if outermost:
            st["entry"] = self
        try:
            return self.env.step(actions)
        finally:
            if outermost:
                st["entry"] = None

    def close(self, **kwargs):
        return self.env.close(**kwargs)

    # staging helpers: device tensors pass through, NumPy batches go to the device and back
    def _dev(self):
        return getattr(self.env, "_device_index", 0)

    def _to_device(self, x, dtype=None):
        torch = _torch()
        if isinstance(x, torch.Tensor):
            return x.contiguous(), True
        t = torch.from_numpy(np.ascontiguousarray(x if dtype is None else np.asarray(x, dtype=dtype))).to(f"cuda:{self._dev()}")
        return t, False

    @staticmethod
    def _back(t, was_tensor):
        return t if was_tensor else t.cpu().numpy()


class RecordEpisodeStatistics(VectorWrapper):
    """gymnasium.wrappers.vector.RecordEpisodeStatistics (wrappers/vector/common.py:22-235) on the engine's own accounting: the step
    kernels accumulate each sub-environment's return and length on the device (NEXT_STEP: the autoreset step does not count; SAME_STEP:
    every step counts) and hand out the rows of the episodes that just ended; this class adds the reference's bookkeeping around them --
    ``infos[stats_key] = {"r", "l", "t"}`` + ``infos["_" + stats_key]``, ``episode_count`` and the three bounded queues."""

    _transparent = True

    def __init__(self, env, buffer_length: int = 100, stats_key: str = "episode"):
        from collections import deque

        super().__init__(env)
        if not hasattr(env.unwrapped, "enable_episode_statistics"):
            raise TypeError("RecordEpisodeStatistics of gymnasium_amd wraps a HipVectorEnv (use gymnasium's own wrapper for other vector envs)")
        env.unwrapped.enable_episode_statistics()
        self._stats_key = stats_key
        self.time_queue, self.return_queue, self.length_queue = deque(maxlen=buffer_length), deque(maxlen=buffer_length), deque(maxlen=buffer_length)

    @property
    def episode_count(self):
        return self.env.unwrapped.episode_count

    def step(self, actions):
        obs, rewards, terminations, truncations, infos = self.env.step(actions)
        if "_episode" in infos:
            stats, dones = infos.pop("episode"), infos.pop("_episode")
            if self._stats_key in infos or f"_{self._stats_key}" in infos:
                raise ValueError(f"Attempted to add episode stats with key '{self._stats_key}' but this key already exists in info: {list(infos.keys())}")
            infos[self._stats_key], infos[f"_{self._stats_key}"] = stats, dones
            # common.py:214-217: the queues take the finished episodes in sub-environment order.  Device-resident infos (output="torch"): the
            # queues live on the host, so this wrapper reads the done mask back every step (the env underneath does not synchronise by itself)
            host = (lambda x: x.cpu().numpy()) if hasattr(dones, "cpu") else (lambda x: x)
            idx = np.flatnonzero(host(dones))
            if idx.size:
                self.time_queue.extend(host(stats["t"])[idx]), self.return_queue.extend(host(stats["r"])[idx]), self.length_queue.extend(host(stats["l"])[idx])
        return obs, rewards, terminations, truncations, infos


class NumpyToTorch(VectorWrapper):
    """gymnasium.wrappers.vector.NumpyToTorch (wrappers/vector/numpy_to_torch.py:16-53) without the conversion: the wrapped HipVectorEnv is
    switched to ``output="torch"``, so observations, rewards and flags ARE torch tensors the engine wrote in HBM (zero copy; the reference
    wrapper converts NumPy batches with ``torch.from_numpy`` / DLPack after the fact) and actions may be device tensors.  ``device``: where
    the caller wants the tensors -- None or the env's own GPU costs nothing, anything else (e.g. "cpu") is one ``.to(device)`` per array.
    The arrays inside ``infos`` become tensors as well, like the reference's recursive conversion."""

    _transparent = True

    def __init__(self, env, device=None):
        super().__init__(env)
        if not hasattr(env.unwrapped, "set_output"):
            raise TypeError("NumpyToTorch of gymnasium_amd wraps a HipVectorEnv (use gymnasium's own wrapper for other vector envs)")
        env.unwrapped.set_output("torch")
        self.device = device

    def _move(self, x):
        torch = _torch()
        if isinstance(x, dict):
            return {k: self._move(v) for k, v in x.items()}
        if isinstance(x, np.ndarray):
            if x.dtype == object:  # final_obs under SAME_STEP: a ragged object array stays as it is
                return x
            x = torch.from_numpy(x)
            return x.to(self.device if self.device is not None else f"cuda:{self._dev()}")
        if isinstance(x, torch.Tensor) and self.device is not None and x.device != torch.device(self.device):
            return x.to(self.device)
        return x

    def reset(self, *, seed=None, options=None):
        obs, infos = self.env.reset(seed=seed, options=options)
        return self._move(obs), self._move(infos)

    def step(self, actions):
        obs, rewards, terminations, truncations, infos = self.env.step(actions)
        return self._move(obs), self._move(rewards), self._move(terminations), self._move(truncations), self._move(infos)


class NormalizeObservation(VectorWrapper):
    """stateful_observation.py:27-160."""

    def __init__(self, env, epsilon: float = 1e-8):
        if epsilon <= 0:
            raise error.InvalidBound(f"`epsilon` should be strictly positive. Received {epsilon}")
        super().__init__(env)
        if self.env.metadata.get("autoreset_mode", AutoresetMode.NEXT_STEP) not in {AutoresetMode.NEXT_STEP}:
            raise ValueError(f"Expected env.metadata['autoreset_mode'] to be AutoresetMode.NEXT_STEP, got {self.env.metadata['autoreset_mode']}")
        shape = self.env.single_observation_space.shape
        self.single_observation_space = spaces.Box(low=-np.inf, high=np.inf, shape=shape, dtype=np.float32)
        self.observation_space = batch_space(self.single_observation_space, self.env.num_envs)
        in_dtype = np.dtype(self.env.single_observation_space.dtype)
        if in_dtype not in (np.float32, np.float64):
            raise ValueError(f"NormalizeObservation needs float32 / float64 observations, got {in_dtype}")
        # RunningMeanStd(dtype=float32) in the reference; float64 observations promote the statistics to float64 on the first update
        self.obs_rms = RunningMeanStd(shape=shape, dtype=np.float32 if in_dtype == np.float32 else np.float64, device=self._dev())
        self._in_code = _native.MI_F32 if in_dtype == np.float32 else _native.MI_F64
        self.epsilon = epsilon
        self._update_running_mean = True
        base = _fusion_base(self.env)
        if base is not None and in_dtype == np.float32 and base._fusion_state()["obs"] is None:
            self._fused, self._base = True, base
            self._fuse_index = base._fuse(self, "obs")
        else:
            _close_fusion(self.env)

    @property
    def update_running_mean(self) -> bool:
        return self._update_running_mean

    @update_running_mean.setter
    def update_running_mean(self, setting: bool):
        self._update_running_mean = setting

    def observations(self, observations):
        torch = _torch()
        x, was_tensor = self._to_device(observations)
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        lib = self.obs_rms._lib
        lib.check(lib.normalize_observation(self.obs_rms._h, _stream(), C.c_void_p(x.data_ptr()), self._in_code, int(x.shape[0]),
                                            float(self.epsilon), int(self._update_running_mean), C.c_void_p(out.data_ptr())))
        return self._back(out, was_tensor)

    def reset(self, *, seed=None, options=None):
        if options is not None and "reset_mask" in options and not np.all(options["reset_mask"]):
            raise ValueError("NormalizeObservation does not support partial resets. The 'reset_mask' must contain all True values.")
        obs, info = self.env.reset(seed=seed, options=options)
        return self.observations(obs), info

    def step(self, actions):
        if self._fused:  # the step kernel's output stage normalises (and updates obs_rms)
            return self._step_fused(actions)
        obs, reward, terminated, truncated, info = self.env.step(actions)
        return self.observations(obs), reward, terminated, truncated, info


class NormalizeReward(VectorWrapper):
    """stateful_reward.py:21-183."""

    def __init__(self, env, gamma: float = 0.99, epsilon: float = 1e-8):
        if not 0 <= gamma <= 1:
            raise error.InvalidBound(f"`gamma` should be in the interval [0, 1]. Received {gamma}")
        if epsilon <= 0:
            raise error.InvalidBound(f"`epsilon` should be strictly positive. Received {epsilon}")
        super().__init__(env)
        torch = _torch()
        dev = f"cuda:{self._dev()}"
        self.return_rms = RunningMeanStd(shape=(), device=self._dev())
        self._acc = torch.zeros(self.env.num_envs, dtype=torch.float32, device=dev)
        self._prev = torch.zeros(self.env.num_envs, dtype=torch.uint8, device=dev)
        self.gamma, self.epsilon = gamma, epsilon
        self._update_running_mean = True
        self._autoreset_mode = self.env.metadata.get("autoreset_mode", AutoresetMode.NEXT_STEP)
        base = _fusion_base(self.env)
        st = None if base is None else base._fusion_state()
        if st is not None and st["ret"] is None and st["clip_post"] is None:  # order inside the epilogue: clip_pre -> normalise -> clip_post
            self._fused, self._base = True, base
            self._fuse_index = base._fuse(self, "ret")
        else:
            _close_fusion(self.env)

    @property
    def accumulated_reward(self):
        return self._acc.cpu().numpy()

    @property
    def update_running_mean(self) -> bool:
        return self._update_running_mean

    @update_running_mean.setter
    def update_running_mean(self, setting: bool):
        self._update_running_mean = setting

    def reset(self, *, seed=None, options=None):
        self._acc.zero_(), self._prev.zero_()
        return self.env.reset(seed=seed, options=options)

    def step(self, actions):
        if self._fused:
            return self._step_fused(actions)
        torch = _torch()
        obs, reward, terminated, truncated, info = self.env.step(actions)
        r, was_tensor = self._to_device(reward, np.float64)
        te, _ = self._to_device(terminated)
        tr, _ = self._to_device(truncated)
        te8, tr8 = te.view(torch.uint8) if te.dtype == torch.bool else te, tr.view(torch.uint8) if tr.dtype == torch.bool else tr
        out = torch.empty_like(r)
        lib = self.return_rms._lib
        lib.check(lib.normalize_reward(self.return_rms._h, _stream(), C.c_void_p(self._acc.data_ptr()), C.c_void_p(self._prev.data_ptr()),
                                       C.c_void_p(r.data_ptr()), C.c_void_p(te8.data_ptr()), C.c_void_p(tr8.data_ptr()), int(r.shape[0]),
                                       float(self.gamma), float(self.epsilon), int(self._autoreset_mode == AutoresetMode.SAME_STEP),
                                       int(self._update_running_mean), C.c_void_p(out.data_ptr())))
        return obs, self._back(out, was_tensor), terminated, truncated, info


class ClipReward(VectorWrapper):
    """vectorize_reward.py:115-151 (transform_reward.ClipReward: np.clip(reward, min_reward, max_reward))."""

    def __init__(self, env, min_reward=None, max_reward=None):
        if min_reward is None and max_reward is None:
            raise error.InvalidBound("Both `min_reward` and `max_reward` cannot be None")
        if min_reward is not None and max_reward is not None and np.any(max_reward - min_reward < 0):
            raise error.InvalidBound(f"Min reward ({min_reward}) must be smaller than max reward ({max_reward})")
        super().__init__(env)
        self.min_reward, self.max_reward = min_reward, max_reward
        self._lib = _native.load_library()
        base = _fusion_base(self.env)
        st = None if base is None else base._fusion_state()
        scalar = all(b is None or np.ndim(b) == 0 for b in (min_reward, max_reward))
        slot = None
        if st is not None and scalar:
            if st["ret"] is None and st["clip_pre"] is None and st["clip_post"] is None:
                slot = "clip_pre"
            elif st["clip_post"] is None:
                slot = "clip_post"
        if slot is not None:
            self._fused, self._base = True, base
            self._fuse_index = base._fuse(self, slot)
        else:
            _close_fusion(self.env)

    def step(self, actions):
        if self._fused:
            return self._step_fused(actions)
        torch = _torch()
        obs, reward, terminated, truncated, info = self.env.step(actions)
        r, was_tensor = self._to_device(reward, np.float64)
        out = torch.empty_like(r)
        lo = None if self.min_reward is None else C.c_double(float(self.min_reward))
        hi = None if self.max_reward is None else C.c_double(float(self.max_reward))
        self._lib.check(self._lib.clip_reward(self._dev(), _stream(), C.c_void_p(r.data_ptr()), int(r.shape[0]),
                                              None if lo is None else C.cast(C.byref(lo), C.c_void_p), None if hi is None else C.cast(C.byref(hi), C.c_void_p),
                                              C.c_void_p(out.data_ptr())))
        return obs, self._back(out, was_tensor), terminated, truncated, info
