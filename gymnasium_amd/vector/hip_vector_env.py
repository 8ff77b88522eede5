"""HipVectorEnv: a gymnasium.vector.VectorEnv whose sub-environments are lanes of an MI355X kernel.

Host-side mirror of the reference's vectoriser for this path.  It replaces, with ONE C-ABI call per method,

  gymnasium/vector/sync_vector_env.py:76-185   __init__  (spaces, metadata["autoreset_mode"], buffers)
  gymnasium/vector/sync_vector_env.py:187-264  reset     (seed fan-out seed+i, options["reset_mask"])
  gymnasium/vector/sync_vector_env.py:266-337  step      (NEXT_STEP / SAME_STEP / DISABLED autoreset)
  gymnasium/wrappers/common.py:116-150         TimeLimit (max_episode_steps, folded into the kernel)
  gymnasium/wrappers/vector/common.py:156-235  RecordEpisodeStatistics (optional, accumulated on device)

and is created through the reference's own plug-in boundary, ``make_vec(id, num_envs,
vectorization_mode="vector_entry_point", **kwargs)`` (envs/registration.py:933-963), which calls the subclass
as ``creator(num_envs=..., max_episode_steps=..., **kwargs)``.

Observations/rewards/terminations/truncations have the reference's shapes and dtypes ((N, obs_dim) float32,
(N,) float64, (N,) bool, (N,) bool).  With ``output="numpy"`` they are NumPy arrays (copies unless
``copy=False``, like SyncVectorEnv); with ``output="torch"`` they are tensors resident in HBM and nothing
crosses PCIe (VectorEnv is generic in its array type, vector_env.py:20,42).
"""
from __future__ import annotations

import time
from typing import Any

import numpy as np

from .. import _native
from ..gym_api import AutoresetMode, VectorEnv, batch_space, error, logger, seeding
from . import device_policy

_U64 = (1 << 64) - 1


def _verify_number_and_cast(x) -> float:
    """envs/classic_control/utils.py:9-15."""
    try:
        return float(x)
    except (ValueError, TypeError) as e:
        raise ValueError(f"An option ({x}) could not be converted to a float.") from e


def parse_low_high(options, default_low, default_high):
    """envs/classic_control/utils.py:17-46 maybe_parse_reset_bounds."""
    if options is None:
        return None
    low = _verify_number_and_cast(options.get("low") if "low" in options else default_low)
    high = _verify_number_and_cast(options.get("high") if "high" in options else default_high)
    if low > high:
        raise ValueError(f"Lower bound ({low}) must be lower than higher bound ({high}).")
    return (low, high)


def _pcg_words_for_seed(seed: int) -> np.ndarray:
    gen, _ = seeding.np_random(seed)
    return _native.pcg_words(gen)


class HipVectorEnv(VectorEnv):
    """Base class; subclasses set KIND, spaces, default reset bounds and constructor params."""

    KIND: str = ""
    DEFAULT_MAX_EPISODE_STEPS: int | None = None
    INFO_KEYS: tuple = ()        # names of the engine's scalar info columns (MuJoCo envs); the first N_RESET_INFO_KEYS are also
    N_RESET_INFO_KEYS: int = 0   # what the scalar env's reset() reports (_get_reset_info), i.e. valid on autoreset steps
    INFO_VECTOR_KEYS: tuple = () # (name, width) array-valued info entries stored after the scalar columns; reported by step AND reset
    INFO_DTYPES: dict = {}       # info key -> dtype of the reference's entry where it is not float64 (e.g. reward_ctrl of a float32 action: np.float32)
    HOST_INFOS = False           # True: the infos need host-side table look-ups (ToyText): built on the host also with output="torch"
    metadata: dict[str, Any] = {"render_modes": [], "autoreset_mode": AutoresetMode.NEXT_STEP}

    # -- to be provided by subclasses ------------------------------------------------------------------
    def _single_spaces(self):
        raise NotImplementedError

    def _engine_params(self) -> tuple:
        return ()

    def _engine_options(self) -> int:
        """MI_CFG_* option bits of mi_config.reserved[0]."""
        return 0

    # -- wrappers fused into the step kernel (gymnasium_amd/wrappers/vector.py, mi_set_step_epilogue) ----------------------------------
    FUSES_WRAPPERS = False  # classic control: NormalizeObservation / NormalizeReward / ClipReward run as the step kernel's output stage

    def _fusion_state(self):
        st = self.__dict__.get("_fused")
        if st is None:
            # chain: the fused wrappers in wrapping order (innermost first); entry: the OUTERMOST fused wrapper the running step() call came
            # through (None: the env is being stepped directly); attached: signature of the epilogue the engine currently holds
            st = self._fused = {"obs": None, "ret": None, "clip_pre": None, "clip_post": None, "closed": False, "chain": [], "entry": None, "attached": None}
        return st

    def _can_fuse(self) -> bool:
        return (self.FUSES_WRAPPERS and self._engine_factory is None and hasattr(self._engine.lib, "set_step_epilogue")
                and not self._fusion_state()["closed"])

    def _fuse(self, wrapper, slot: str) -> int:
        """Register ``wrapper`` as the next member of the fused unit; returns its position in the chain."""
        st = self._fusion_state()
        st[slot] = wrapper
        st["chain"].append((slot, wrapper))
        return len(st["chain"]) - 1

    def _sync_epilogue(self):
        """Attach exactly the epilogue of the step() call in progress.  The wrapped values belong to the WRAPPER they come out of
        (the reference's contract: the inner env returns raw values): a step entered through fused wrapper k runs the arithmetic of the
        chain's members 0..k as the step kernel's output stage, a step of the env itself -- or of a wrapper below k -- runs none / fewer.
        Settings are read from the wrappers at every step (a tuple compare), so a later ``w.gamma = ...`` / ``w.min_reward = ...`` /
        ``w.update_running_mean = False`` takes effect like in the reference; the engine is only re-configured when something changed."""
        st = self.__dict__.get("_fused")
        if st is None or not st["chain"]:
            return
        entry = st["entry"]
        level = -1 if entry is None else entry._fuse_index
        members = st["chain"][:level + 1]
        sig = [level]
        for slot, w in members:
            # (the statistics HANDLES and accumulator addresses are part of the signature: replacing `w.obs_rms` / `w.return_rms`, e.g. when a
            # checkpoint is restored, must re-attach the epilogue to the new object)
            if slot == "obs":
                sig += [float(w.epsilon), bool(w.update_running_mean), w.obs_rms._h.value]
            elif slot == "ret":
                sig += [float(w.gamma), float(w.epsilon), bool(w.update_running_mean),
                        w.return_rms._h.value, w._acc.data_ptr(), w._prev.data_ptr()]
            else:
                sig += [None if w.min_reward is None else float(w.min_reward), None if w.max_reward is None else float(w.max_reward)]
        sig = tuple(sig)
        if sig == st["attached"]:
            return
        if not members:
            self._epilogue_struct = None
            self._engine.set_step_epilogue(None)
        else:
            e = _native.MiStepEpilogue()
            for slot, w in members:
                if slot == "obs":
                    e.obs_rms, e.obs_epsilon, e.obs_update = w.obs_rms._h, float(w.epsilon), int(w.update_running_mean)
                elif slot == "ret":
                    e.return_rms, e.accumulated, e.prev_done = w.return_rms._h, w._acc.data_ptr(), w._prev.data_ptr()
                    e.gamma, e.reward_epsilon, e.reward_update = float(w.gamma), float(w.epsilon), int(w.update_running_mean)
                else:
                    lo, hi = w.min_reward, w.max_reward
                    setattr(e, slot, (1 if lo is not None else 0) | (2 if hi is not None else 0))
                    setattr(e, slot + "_min", 0.0 if lo is None else float(lo)), setattr(e, slot + "_max", 0.0 if hi is None else float(hi))
            self._epilogue_struct = e  # keep the ctypes object alive
            self._engine.set_step_epilogue(e)
        st["attached"] = sig

    def _parse_reset_options(self, options):
        """Return the env-specific (b0, b1) reset bounds or None for defaults; raise ValueError like the reference."""
        return None

    def _short_step_allowed(self) -> bool:
        """May step() take its short path (device tensor in, the env's own output tensors out)?  Subclasses whose step() post-processes say no."""
        return True

    # --------------------------------------------------------------------------------------------------
    def __init__(self, num_envs: int = 1, max_episode_steps: int | None = None, autoreset_mode=AutoresetMode.NEXT_STEP,
                 render_mode: str | None = None, device=None, output: str = "numpy", copy: bool = True,
                 env_index_offset: int = 0, record_episode_statistics: bool = False, strict_actions: bool = False, sample_output: str | None = None,
                 _engine_factory=None):
        if render_mode is not None:
            raise error.Error("gymnasium_amd sub-environments live on the GPU and cannot render; use render_mode=None")
        if output not in ("numpy", "torch"):
            raise ValueError(f"output must be 'numpy' or 'torch', got {output!r}")
        self.num_envs = int(num_envs)
        if self.num_envs < 1:
            raise ValueError(f"num_envs must be >= 1, got {num_envs}")
        self.autoreset_mode = autoreset_mode if isinstance(autoreset_mode, AutoresetMode) else AutoresetMode(autoreset_mode)
        self.metadata = dict(type(self).metadata)
        self.metadata["autoreset_mode"] = self.autoreset_mode
        self.render_mode = None
        self.copy = bool(copy)
        # output="torch": the engine runs asynchronously on torch's stream, so what the reference raises at once -- the AssertionError for an
        # action outside the space (cartpole.py:165-167), stepping a finished sub-environment under DISABLED autoreset -- is DEFERRED: the
        # kernel records a sticky error word in page-locked memory (the sub-environment with the invalid action is left untouched) and a later `step()` -- the first
        # one that finds the word set, usually the next -- or any synchronising call (`synchronize()`, `statistics()`) raises it.  strict_actions=True synchronises after every step and raises there, at the cost
        # of the asynchrony (debugging aid).  NumPy input is always validated on the host before anything is mutated.
        self.strict_actions = bool(strict_actions)
        self.output = output
        # What `action_space.sample()` hands out (vector/device_policy.py).  Either way the batch is drawn by the engine from the space's own
        # stream, bit-equal to the NumPy sampler.  "numpy" (default): NumPy arrays like the reference's, whatever `output` is -- scripts that
        # treat a sample as an ndarray keep working.  "torch" (needs output="torch"): device tensors, so that the metric's own loop
        # `env.step(env.action_space.sample())` (utils/performance.py:82-97) enqueues nothing but step kernels.
        sample_output = "numpy" if sample_output is None else sample_output
        if sample_output not in ("numpy", "torch") or (sample_output == "torch" and output != "torch"):
            raise ValueError(f"sample_output must be 'numpy' or (with output='torch') 'torch', got {sample_output!r} with output={output!r}")
        self.sample_output = sample_output
        self.max_episode_steps = self.DEFAULT_MAX_EPISODE_STEPS if max_episode_steps is None else max_episode_steps
        self.env_index_offset = int(env_index_offset)
        self.record_episode_statistics = bool(record_episode_statistics)

        self.single_observation_space, self.single_action_space = self._single_spaces()
        self.observation_space = batch_space(self.single_observation_space, self.num_envs)
        self.action_space = batch_space(self.single_action_space, self.num_envs)

        self._device_index = _resolve_device(device)
        self._engine_factory = _engine_factory
        if _engine_factory is None:
            lib = _native.load_library()  # raises ImportError if the HIP library was not built
            self._engine = _native.Engine(lib, self.KIND, self.num_envs, self.max_episode_steps, self.autoreset_mode.value,
                                          self._engine_params(), self._device_index, options=self._engine_options())
        else:  # test seam: tests drive this host class against a checker backend; never used by the package itself
            self._engine = _engine_factory(self.KIND, self.num_envs, self.max_episode_steps, self.autoreset_mode.value,
                                           self._engine_params(), self._device_index, options=self._engine_options())
        eng = self._engine
        self._discrete = eng.act_dtype is np.int64
        self._act_shape = (self.num_envs,) if self._discrete else (self.num_envs, eng.act_dim)
        # action_space.sample() from the engine's action stream (vector/device_policy.py): same draws as the NumPy sampler, no host sampling
        self._device_policy = hasattr(getattr(eng, "lib", None), "action_sample") and not getattr(self, "_host_policy_only", False)
        self.action_space = device_policy.attach(self.action_space, self, eng.act_dim)
        self.last_sampled_actions = None  # step(None): the batch the on-device policy drew in the last such step
        self._seeded = False
        self._has_reset = False
        self._async_pending = None
        self._was_done = np.zeros(self.num_envs, dtype=np.bool_)  # mirror of the device's needs-reset flags (SyncVectorEnv._autoreset_envs)
        self._alloc_buffers()
        if self.record_episode_statistics:
            self._episode_count = 0
            self._episode_start = np.zeros(self.num_envs)
            self._prev_dones = np.zeros(self.num_envs, dtype=np.bool_)

    @property
    def episode_count(self) -> int:
        """Episodes finished so far (RecordEpisodeStatistics.episode_count); with device-resident infos the count lives on the device and
        reading it synchronises."""
        dev = self.__dict__.get("_episode_count_t")
        return getattr(self, "_episode_count", 0) + (int(dev.item()) if dev is not None else 0)

    def set_output(self, output: str):
        """Switch between NumPy batches ("numpy": the engine's pinned host block, one H2D + one D2H per step) and device tensors ("torch": the engine writes
        straight into torch tensors in HBM, nothing crosses PCIe) after construction -- what wrappers.vector.NumpyToTorch(env) does.
        Call before reset()."""
        if output not in ("numpy", "torch"):
            raise ValueError(f"output must be 'numpy' or 'torch', got {output!r}")
        if output != self.output:
            if hasattr(self.action_space, "_hip_to_host"):
                self.action_space._hip_to_host()  # (batches drawn ahead have the old mode's array type)
            if output == "numpy":
                self.sample_output = "numpy"
            self.output = output
            self._alloc_buffers()
            self._has_reset = False

    def enable_episode_statistics(self):
        """Switch on the on-device episode accounting after construction (what wrappers.vector.RecordEpisodeStatistics(env) does):
        the step kernels start writing the finished episodes' return / length rows.  Call before reset()."""
        if not self.record_episode_statistics:
            self.record_episode_statistics = True
            self._episode_count = 0
            self._episode_start = np.zeros(self.num_envs)
            self._prev_dones = np.zeros(self.num_envs, dtype=np.bool_)
            self._alloc_buffers()
            self._has_reset = False

    # -- buffers ---------------------------------------------------------------------------------------
    def _alloc_buffers(self):
        N, eng = self.num_envs, self._engine
        old_count = self.__dict__.pop("_episode_count_t", None)
        if old_count is not None:  # episodes counted on the device so far survive a re-allocation / a switch of the output mode
            self._episode_count = getattr(self, "_episode_count", 0) + int(old_count.item())
        if self.output == "torch":
            import torch

            self._torch = torch
            # (the test seam's checker backend computes on the host: its "device" tensors are host tensors, same code path otherwise)
            dev = torch.device("cuda", self._device_index) if self._engine_factory is None else torch.device("cpu")
            self._tdev = dev
            self._obs_tdtype = {np.float64: torch.float64, np.float32: torch.float32, np.int64: torch.int64}[eng.obs_dtype]
            self._obs_shape = (N,) if (eng.obs_dtype is np.int64 and eng.obs_dim == 1) else (N, eng.obs_dim)  # Discrete states batch to MultiDiscrete: (N,)
            self._obs = torch.zeros(self._obs_shape, dtype=self._obs_tdtype, device=dev)
            self._rew = torch.zeros((N,), dtype=torch.float64, device=dev)
            self._term = torch.zeros((N,), dtype=torch.bool, device=dev)
            self._trunc = torch.zeros((N,), dtype=torch.bool, device=dev)
            self._final = torch.zeros(self._obs_shape, dtype=self._obs_tdtype, device=dev) if self.autoreset_mode == AutoresetMode.SAME_STEP else None
            self._info = torch.zeros((N, eng.info_dim), dtype=torch.float64, device=dev) if eng.info_dim else None
            self._final_info = torch.zeros((N, eng.info_dim), dtype=torch.float64, device=dev) if (eng.info_dim and self._final is not None) else None
            self._ep_r = torch.zeros((N,), dtype=torch.float64, device=dev) if self.record_episode_statistics else None
            self._ep_l = torch.zeros((N,), dtype=torch.int32, device=dev) if self.record_episode_statistics else None
            self._loc = _native.MI_DEVICE
            # device-resident infos (no read-back per step): the pending-autoreset set and the episode clocks live on the device too
            self._device_infos = not self.HOST_INFOS
            self._was_done_t = torch.zeros((N,), dtype=torch.bool, device=dev)
            self._all_true_t = torch.ones((N,), dtype=torch.bool, device=dev)
            if self.record_episode_statistics:
                self._episode_start_t = torch.zeros((N,), dtype=torch.float64, device=dev)
                self._prev_dones_t = torch.zeros((N,), dtype=torch.bool, device=dev)
                self._episode_count_t = torch.zeros((), dtype=torch.int64, device=dev)
            # the addresses step() writes to never change between two allocations: hand them to the binding once (Engine.bind_step)
            eng.bind_step(self._p(self._obs), self._p(self._rew), self._p(self._term), self._p(self._trunc), self._p(self._final), self._p(self._ep_r),
                          self._p(self._ep_l), self._loc, self._p(self._info), self._p(self._final_info))
            self._raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) if self._engine_factory is None else None
            self._stream_bound = None
            # step()'s short path (device tensors in, device tensors out, nothing for the host to assemble): see step()
            self._act_tdtype = torch.int64 if self._discrete else torch.float32
            self._short_step = (not self.INFO_KEYS and self.autoreset_mode != AutoresetMode.SAME_STEP and not self.record_episode_statistics
                                and not self.strict_actions and type(self).step in _SHORT_STEP_OWNERS and self._short_step_allowed())
        else:
            self._obs_shape = (N,) if (eng.obs_dtype is np.int64 and eng.obs_dim == 1) else (N, eng.obs_dim)
            same = self.autoreset_mode == AutoresetMode.SAME_STEP
            # The engine's own PINNED host arrays (mi_host_buffers): the step's single D2H lands directly in what step() returns
            # (copy=False) or copies from (copy=True, like SyncVectorEnv's deepcopy).  A backend without them (the checker): plain arrays.
            hb = eng.host_buffers() if hasattr(eng, "host_buffers") else None
            self._pinned = hb is not None
            self.action_buffer = hb["actions"] if hb else None  # sample / write actions here to skip the staging memcpy of step()
            self._obs = hb["obs"] if hb else np.zeros(self._obs_shape, dtype=eng.obs_dtype)
            self._rew = hb["reward"] if hb else np.zeros((N,), dtype=np.float64)
            self._term = hb["terminated"] if hb else np.zeros((N,), dtype=np.bool_)
            self._trunc = hb["truncated"] if hb else np.zeros((N,), dtype=np.bool_)
            self._final = (hb["final_obs"] if hb else np.zeros(self._obs_shape, dtype=eng.obs_dtype)) if same else None
            self._info = (hb["info"] if hb else np.zeros((N, eng.info_dim), dtype=np.float64)) if eng.info_dim else None
            self._final_info = (hb["final_info"] if hb else np.zeros((N, eng.info_dim), dtype=np.float64)) if (eng.info_dim and same) else None
            self._ep_r = (hb["episode_return"] if hb else np.zeros((N,), dtype=np.float64)) if self.record_episode_statistics else None
            self._ep_l = (hb["episode_length"] if hb else np.zeros((N,), dtype=np.int32)) if self.record_episode_statistics else None
            self._loc = _native.MI_HOST
            self._device_infos = False

    def _p(self, buf):
        if buf is None:
            return None
        return buf.data_ptr() if self.output == "torch" else buf

    def _out(self, buf):
        if not self.copy:
            return buf
        return buf.clone() if self.output == "torch" else buf.copy()

    def _bind_stream(self):
        """The engine enqueues on torch's CURRENT stream of its device (asked at every step; the engine is told only when it changed)."""
        if self.output == "torch" and self._engine_factory is None:
            raw = self._raw_stream
            st = raw(self._device_index) if raw is not None else self._torch.cuda.current_stream(self._tdev).cuda_stream
            if st != self._stream_bound:
                self._engine.set_stream(st)
                self._stream_bound = st

    # -- seeding ---------------------------------------------------------------------------------------
    def _seed_engines(self, seed, mask):
        """SyncVectorEnv seed fan-out (sync_vector_env.py:204-212): int -> seed+i, list -> per env, None -> keep."""
        N, off = self.num_envs, self.env_index_offset
        if seed is None:
            if not self._seeded:  # like Env.np_random's lazy OS-entropy seeding (core.py:227-236)
                base = int(np.random.SeedSequence().generate_state(1, np.uint64)[0]) >> 1
                self._engine.seed_sequence(base, off, None)
                self._seeded = True
            return
        if isinstance(seed, (int, np.integer)) and not isinstance(seed, bool):
            seed = int(seed)
            if seed < 0:
                raise error.Error(f"Seed must be greater or equal to zero, actual value: {seed}")
            if seed + off + N - 1 <= _U64:
                self._engine.seed_sequence(seed, off, mask)  # SeedSequence + PCG64 seeding evaluated on device
            else:  # beyond 64 bits: NumPy on the host
                words = np.stack([_pcg_words_for_seed(seed + off + i) for i in range(N)])
                self._engine.seed(words, mask)
            self._seeded = True
            return
        seeds = list(seed)
        if len(seeds) != N:
            raise ValueError(f"If seeds are passed as a list the length must match num_envs={N} but got length={len(seeds)}.")
        words = np.zeros((N, 4), dtype=np.uint64)
        smask = np.zeros(N, dtype=np.uint8)
        for i, s in enumerate(seeds):
            if s is None:
                continue
            words[i] = _pcg_words_for_seed(s)
            smask[i] = 1
        if not self._seeded and not smask.all():
            base = int(np.random.SeedSequence().generate_state(1, np.uint64)[0]) >> 1
            self._engine.seed_sequence(base, off, None)
        if mask is not None:
            smask &= np.asarray(mask, dtype=np.uint8)
        if smask.any():
            self._engine.seed(words, smask)
        self._seeded = True

    # -- API -------------------------------------------------------------------------------------------
    def reset(self, *, seed=None, options=None):
        """Reset the sub-environments (all, or options["reset_mask"]) and return (observations, infos)."""
        self._check_open()
        self._check_not_pending("reset")
        if isinstance(seed, (int, np.integer)) and not isinstance(seed, bool):
            super().reset(seed=int(seed))
        mask = None
        if options is not None and "reset_mask" in options:
            options = dict(options)
            reset_mask = options.pop("reset_mask")
            if not isinstance(reset_mask, np.ndarray):
                raise TypeError(f"`options['reset_mask']` must be a numpy array, got {type(reset_mask)}")
            if reset_mask.shape != (self.num_envs,):
                raise ValueError(f"`options['reset_mask']` must have shape `({self.num_envs},)`, got {reset_mask.shape}")
            if reset_mask.dtype != np.bool_:
                raise TypeError(f"`options['reset_mask']` must have `dtype=np.bool_`, got {reset_mask.dtype}")
            if not np.any(reset_mask):
                raise ValueError(f"`options['reset_mask']` must contain a boolean array with at least one True value, got reset_mask={reset_mask}")
            mask = np.ascontiguousarray(reset_mask).view(np.uint8)
        bounds = self._parse_reset_options(options if options else None)
        self._seed_engines(seed, mask)
        self._bind_stream()
        if self.output == "torch":
            dmask = None
            if mask is not None:
                dmask = self._torch.from_numpy(mask.copy()).to(self._tdev)
            self._engine.reset(None if dmask is None else dmask.data_ptr(), bounds, self._obs.data_ptr(), _native.MI_DEVICE)
        else:
            self._engine.reset(mask, bounds, self._obs, _native.MI_HOST)
        self._has_reset = True
        if mask is None:
            self._was_done[:] = False
        else:  # sync_vector_env.py:232-234: only the reset sub-envs leave the pending-autoreset set
            self._was_done[mask.view(np.bool_)] = False
        if self.record_episode_statistics:
            now = time.perf_counter()
            if mask is None:
                self._episode_start[:] = now
                self._prev_dones[:] = False
            else:
                self._episode_start[mask.view(np.bool_)] = now
                self._prev_dones[mask.view(np.bool_)] = False
        if self._device_infos:
            keep = None if mask is None else ~dmask.view(self._torch.bool)
            # (every update of the device-side bookkeeping is IN PLACE: a HIP graph captured around step() keeps reading and writing these tensors)
            self._was_done_t.logical_and_(keep) if keep is not None else self._was_done_t.zero_()
            if self.record_episode_statistics:
                if keep is None:
                    self._episode_start_t.fill_(now), self._prev_dones_t.zero_()
                else:
                    self._episode_start_t.copy_(self._torch.where(keep, self._episode_start_t, now))
                    self._prev_dones_t.logical_and_(keep)
        return self._out(self._obs), self._reset_infos(mask)

    def _reset_infos(self, mask) -> dict:
        """The scalar envs' reset info (``_get_reset_info`` / ``{"prob": 1}``) batched like VectorEnv._add_info does."""
        return {}

    def _coerce_actions(self, actions):
        """The action batch as the engine reads it: (object to keep alive, pointer, mi_dtype of a Box row).

        Box action spaces: the reference hands every sub-environment its row of the caller's array AS IT IS (sync_vector_env.py:274 iterate();
        pendulum.py:127-134, continuous_mountain_car.py:153, mujoco_env.py:148 `data.ctrl[:] = ctrl`) -- a float64 array is not rounded to the
        space's float32, and NumPy's promotion rules then make parts of the step float64 arithmetic.  So float64 rows (and integer arrays:
        exact in float64) go to the engine un-rounded with actions_dtype = MI_F64; float32 rows take the float32 path, which is also the
        on-device sampler's type.  A list of Python lists reaches the scalar envs as Python lists: `action[0]` is then a Python float, which
        NumPy 2 treats as WEAK (np.float32 + float stays float32) -- MI_F64_WEAK; only MountainCarContinuous tells it from MI_F64.  A row of
        NumPy scalars (`[[np.float64(x)], ...]`) is STRONG, like an ndarray row: only exact Python floats / ints are weak.
        NOT reproduced: float16 / bfloat16 rows.  The reference would then compute parts of the step in half precision; the engine widens
        them to float32 (one warning per env), so those trajectories are within tolerance of, not bit-equal to, the reference's."""
        eng = self._engine
        if self.output == "torch" and hasattr(actions, "data_ptr"):
            t = self._torch
            if self._discrete:
                want = t.int64
            else:
                want = t.float32 if actions.dtype in (t.float32, t.float16, t.bfloat16) else t.float64
                if actions.dtype in (t.float16, t.bfloat16):
                    self._warn_half_precision(actions.dtype)
            if actions.device != self._tdev or actions.dtype != want or not actions.is_contiguous():
                actions = actions.to(device=self._tdev, dtype=want).contiguous()
            if actions.numel() != self.num_envs * eng.act_dim:
                raise ValueError(f"actions must have {self.num_envs * eng.act_dim} elements, got shape {tuple(actions.shape)}")
            return actions, actions.data_ptr(), (_native.MI_F64 if want is t.float64 else _native.MI_F32)
        a = np.asarray(actions)
        dtype = _native.MI_F32
        if self._discrete:
            if not np.issubdtype(a.dtype, np.integer):
                raise AssertionError(f"{actions!r} ({type(actions)}) invalid")
            a = np.ascontiguousarray(a, dtype=np.int64)
        elif a.dtype in (np.float32, np.float16):
            if a.dtype == np.float16:
                self._warn_half_precision(a.dtype)
            a = np.ascontiguousarray(a, dtype=np.float32)
        else:
            # weak = every leaf is an exact Python float / int (NEP 50); np.generic scalars inside the lists are strong like an ndarray's elements
            weak = (isinstance(actions, (list, tuple)) and len(actions) > 0
                    and all(isinstance(r, (list, tuple)) and all(type(x) in (float, int) for x in r) for r in actions))
            a, dtype = np.ascontiguousarray(a, dtype=np.float64), (_native.MI_F64_WEAK if weak else _native.MI_F64)
        if a.size != self.num_envs * eng.act_dim:
            raise ValueError(f"actions must have shape {self._act_shape}, got {a.shape}")
        if self.output == "torch":
            ta = self._torch.from_numpy(a).to(self._tdev)
            return ta, ta.data_ptr(), dtype
        return a, a, dtype

    def _warn_half_precision(self, dtype):
        if not getattr(self, "_warned_half", False):
            self._warned_half = True
            logger.warn(f"{dtype} action rows are widened to float32: the reference computes parts of the step in half precision for such rows, "
                        "so this trajectory is within tolerance of the reference's, not bit-equal to it")

    def step(self, actions):
        """One lockstep step of every sub-environment: (obs, rewards, terminations, truncations, infos)."""
        if self.output == "torch" and self._short_step and self._has_reset and self._async_pending is None and not self.closed \
                and actions.__class__ is self._torch.Tensor and actions.dtype is self._act_tdtype and actions.device == self._tdev \
                and actions.is_contiguous() and actions.numel() == self.num_envs * self._engine.act_dim and not self.__dict__.get("_fused"):
            # The metric's loop (utils/performance.py:82-97) with device tensors is host-bound: the step kernel of 65 536 CartPoles runs 5 us, and the
            # general path below -- coercion of whatever the caller passed, the wrappers' epilogue, the infos -- costs about as much in Python.
            # An action tensor that already is what the engine reads, on an env with nothing to assemble on the host, goes straight to mi_step.
            self._bind_stream()
            self._act_f64 = False
            try:
                self._engine.step_bound(actions.data_ptr(), _native.MI_F32)
            except _native.NativeError as e:
                if e.code in (-1, -5):
                    raise AssertionError(e.message) from e
                raise
            if self.copy:
                return self._obs.clone(), self._rew.clone(), self._term.clone(), self._trunc.clone(), {}
            return self._obs, self._rew, self._term, self._trunc, {}
        self._check_open()
        self._check_not_pending("step")
        if not self._has_reset:
            raise AssertionError("Call reset before using step method.")
        aout = None
        if actions is None:
            # the on-device policy: `step(action_space.sample())` in ONE launch -- the step kernel draws the batch from the action stream itself
            # (mi_step with actions == NULL).  Device tensors only; with NumPy batches it is the two calls it stands for.
            eng_stream = self.action_space.hip_use_stream() if (self.output == "torch" and hasattr(self.action_space, "hip_use_stream")) else None
            if eng_stream is None:
                # (not `self.step(...)`: a subclass that reshapes what step() returns -- Blackjack's tuple of columns -- is already on the stack)
                actions = self.action_space.sample()
        if actions is None:
            if self.last_sampled_actions is None:
                t = self._torch
                self.last_sampled_actions = t.zeros(self._act_shape, dtype=t.int64 if self._discrete else t.float32, device=self._tdev)
            keep, aptr, adt, aout = None, None, _native.MI_F32, self.last_sampled_actions.data_ptr()
        else:
            keep, aptr, adt = self._coerce_actions(actions)
        self._act_f64 = adt != _native.MI_F32
        self._bind_stream()
        self._sync_epilogue()
        try:
            if self.output == "torch":
                self._engine.step_bound(aptr, adt, aout)
                if self.strict_actions:
                    self._engine.synchronize()  # raises the device error word of this very step
            else:
                self._engine.step(aptr, self._obs, self._rew, self._term, self._trunc, self._final, self._ep_r, self._ep_l, self._loc, self._info,
                                  self._final_info, actions_dtype=adt)
        except _native.NativeError as e:
            if e.code == -1:  # MI_ERR_INVALID_ARGUMENT: action outside the space (cartpole.py:165-167 asserts)
                raise AssertionError(e.message) from e
            if e.code == -5:
                raise AssertionError(e.message) from e
            raise
        del keep
        infos = self._build_infos()
        return self._out(self._obs), self._out(self._rew), self._out(self._term), self._out(self._trunc), infos

    def _info_columns(self):
        """[(name, first column, width, in_reset_info)] of the engine's info row."""
        cols = [(name, k, 0, k < self.N_RESET_INFO_KEYS) for k, name in enumerate(self.INFO_KEYS)]
        start = len(self.INFO_KEYS)
        for name, width in self.INFO_VECTOR_KEYS:
            cols.append((name, start, width, True))
            start += width
        return cols

    def _info_dict(self, rows, supplied, reset_rows) -> dict:
        """VectorEnv._add_info (vector_env.py:277-338) over an info matrix: one array per key plus the `_key` mask of the
        sub-envs that supplied it.  `supplied`: rows that have an info at all; `reset_rows`: rows whose info is the scalar
        env's RESET info, which carries only the reset keys.  A key no sub-env supplied does not appear."""
        out: dict[str, Any] = {}
        for name, start, width, in_reset in self._info_columns():
            mask = supplied if in_reset else (supplied & ~reset_rows)
            if not mask.any():
                continue
            col = rows[:, start] if width == 0 else rows[:, start:start + width]
            val = np.where(mask if width == 0 else mask[:, None], col, 0.0)
            dt = self._info_dtype(name)
            if dt is not None:  # (the engine's info row is float64; such entries hold float32 / integer values exactly)
                val = val.astype(dt)
            out[name], out["_" + name] = val, mask.copy()
        return out

    def _info_dtype(self, name):
        """dtype of the reference's info entry where it is not float64.  An np.float32 entry is one computed from the action row
        (`reward_ctrl`): float32 for a float32 row, float64 for a float64 row (see _coerce_actions)."""
        dt = self.INFO_DTYPES.get(name)
        if dt is np.float32 and getattr(self, "_act_f64", False):
            return None
        return dt

    def _host(self, buf):
        return buf.cpu().numpy() if self.output == "torch" else buf

    # -- asynchronous stepping (AsyncVectorEnv.step_async / step_wait, vector/async_vector_env.py:440-521) ----------------------------
    def step_async(self, actions):
        """Enqueue one step (actions H2D, kernel, one D2H into the pinned block) and return at once; ``step_wait()`` collects it.
        With device tensors (``output="torch"``) every step is already asynchronous: this is then ``step()`` with the result parked."""
        self._check_open()
        if not self._has_reset:
            raise AssertionError("Call reset before using step method.")
        self._check_not_pending("step_async")
        if self.output == "torch" or not self._pinned:
            self._async_pending = ("done", self.step(actions))
            return
        keep, aptr, adt = self._coerce_actions(actions)
        self._act_f64 = adt != _native.MI_F32
        self._bind_stream()
        self._sync_epilogue()
        try:
            self._engine.step_async(aptr, self._obs, self._rew, self._term, self._trunc, self._final, self._ep_r, self._ep_l, self._info, self._final_info,
                                    actions_dtype=adt)
        except _native.NativeError as e:
            if e.code in (-1, -5):
                raise AssertionError(e.message) from e
            raise
        self._async_pending = ("engine", keep)

    def step_wait(self, timeout=None):
        pending = getattr(self, "_async_pending", None)
        if pending is None:
            raise error.NoAsyncCallError("Calling `step_wait` without any prior call to `step_async`.", "step")  # async_vector_env.py:477-481
        self._async_pending = None
        if pending[0] == "done":
            return pending[1]
        try:
            self._engine.step_wait()
        except _native.NativeError as e:
            if e.code in (-1, -5):
                raise AssertionError(e.message) from e
            raise
        infos = self._build_infos()
        return self._out(self._obs), self._out(self._rew), self._out(self._term), self._out(self._trunc), infos

    def _info_dict_device(self, rows, supplied, reset_rows) -> dict:
        """_info_dict over a device info matrix without reading anything back.  supplied / reset_rows: device bool masks or None (= every
        sub-env / none).  The KEY SET IS STATIC: a key no sub-env supplied in this step is still present (its `_key` mask is all False) --
        deciding otherwise would need the masks on the host, i.e. a synchronisation per step."""
        t, out = self._torch, {}
        for name, start, width, in_reset in self._info_columns():
            col = rows[:, start] if width == 0 else rows[:, start:start + width]
            mask = supplied if (in_reset or reset_rows is None) else (~reset_rows if supplied is None else supplied & ~reset_rows)
            if mask is None:
                val, mask = (col.clone() if self.copy else col), self._all_true_t
            else:
                val = t.where(mask if width == 0 else mask[:, None], col, 0.0)
            dt = self._info_dtype(name)
            if dt is not None:  # same dtypes as the NumPy infos (_info_dict)
                val = val.to({np.float32: t.float32, np.int64: t.int64, np.int32: t.int32, np.bool_: t.bool}[dt])
            out[name], out["_" + name] = val, (mask.clone() if (self.copy and mask is self._all_true_t) else mask)
        return out

    def _build_infos_device(self) -> dict:
        """The infos of a step as DEVICE tensors (output="torch"): nothing is copied to the host and nothing synchronises -- the step stays
        asynchronous for the MuJoCo kinds (whose infos carry x_position, reward terms, ...), under SAME_STEP autoreset and with episode
        statistics.  Differences from the NumPy dict, all forced by not looking at the flags on the host: the key set is static (see
        _info_dict_device); under SAME_STEP `final_obs` is the batched tensor of final observations (rows valid where `_final_obs`), not an
        object array; `episode` is present every step with its `_episode` mask."""
        t, N = self._torch, self.num_envs
        infos: dict[str, Any] = {}
        same_step = self.autoreset_mode == AutoresetMode.SAME_STEP
        dones = self._term | self._trunc
        if self.INFO_KEYS and self._info is not None:
            reset_rows = self._was_done_t if self.autoreset_mode == AutoresetMode.NEXT_STEP else (dones if same_step else None)
            infos.update(self._info_dict_device(self._info, None, reset_rows))
        if same_step:
            infos["final_obs"], infos["_final_obs"] = self._out(self._final), dones
            finfo = self._info_dict_device(self._final_info, dones, None) if (self.INFO_KEYS and self._final_info is not None) else {}
            infos["final_info"], infos["_final_info"] = finfo, dones
        # (private copies: `dones` itself is handed to the caller as the `_episode` / `_final_obs` / `_final_info` masks, and an in-place edit of
        # those must not reach the next step's autoreset bookkeeping -- the NumPy path returns copies as well)
        self._was_done_t.copy_(dones) if not same_step else self._was_done_t.zero_()
        if self.record_episode_statistics:
            now = time.perf_counter()
            if not same_step:
                self._episode_start_t.copy_(t.where(self._prev_dones_t, now, self._episode_start_t))
            self._prev_dones_t.copy_(dones)
            infos["episode"] = {"r": self._out(self._ep_r), "l": self._ep_l.to(t.int64),
                                "t": t.where(dones, t.round((now - self._episode_start_t) * 1e6) / 1e6, 0.0)}
            infos["_episode"] = dones
            self._episode_count_t.add_(dones.sum())
            if same_step:
                self._episode_start_t.copy_(t.where(dones, now, self._episode_start_t))
        return infos

    def _build_infos(self) -> dict:
        infos: dict[str, Any] = {}
        N = self.num_envs
        same_step = self.autoreset_mode == AutoresetMode.SAME_STEP
        if not (self.INFO_KEYS or same_step or self.record_episode_statistics):
            return infos  # nothing to report: with device tensors the step stays asynchronous (no read-back of the flags)
        if self._device_infos:
            return self._build_infos_device()
        dones = np.logical_or(self._host(self._term), self._host(self._trunc))
        every = np.ones(N, dtype=np.bool_)
        if self.INFO_KEYS and self._info is not None:
            # an env in its NEXT_STEP autoreset step supplies only its reset info (sync_vector_env.py:279-284); under SAME_STEP a
            # finished env's top-level entries are its reset info and the finishing step's info goes to "final_info" (:309-319)
            if self.autoreset_mode == AutoresetMode.NEXT_STEP:
                reset_rows = self._was_done
            elif same_step:
                reset_rows = dones
            else:
                reset_rows = np.zeros(N, dtype=np.bool_)
            infos.update(self._info_dict(self._host(self._info), every, reset_rows))
        if same_step and dones.any():
            # sync_vector_env.py:309-317 via VectorEnv._add_info: object array of per-env observations + nested final_info
            final = self._host(self._final)
            arr = np.full(N, None, dtype=object)
            scalar_obs = final.ndim == 1  # Discrete observations: the scalar env returns a Python int (frozen_lake.py:343 `int(s)`)
            for i in np.flatnonzero(dones):
                arr[i] = int(final[i]) if scalar_obs else final[i].copy()
            infos["final_obs"], infos["_final_obs"] = arr, dones.copy()
            finfo = {}
            if self.INFO_KEYS and self._final_info is not None:
                finfo = self._info_dict(self._host(self._final_info), dones, np.zeros(N, dtype=np.bool_))
            infos["final_info"], infos["_final_info"] = finfo, dones.copy()
        self._was_done = dones if self.autoreset_mode != AutoresetMode.SAME_STEP else np.zeros(N, dtype=np.bool_)
        if self.record_episode_statistics:
            now = time.perf_counter()
            any_done = bool(dones.any())
            if self.autoreset_mode != AutoresetMode.SAME_STEP:
                self._episode_start[self._prev_dones] = now
            self._prev_dones = dones
            if any_done:
                r = self._ep_r.cpu().numpy() if self.output == "torch" else self._ep_r.copy()
                ln = self._ep_l.cpu().numpy() if self.output == "torch" else self._ep_l.copy()
                infos["episode"] = {"r": r, "l": ln.astype(np.int64),
                                    "t": np.where(dones, np.round(now - self._episode_start, 6), 0.0)}
                infos["_episode"] = dones.copy()
                self._episode_count += int(dones.sum())
                if self.autoreset_mode == AutoresetMode.SAME_STEP:
                    self._episode_start[dones] = now
        return infos

    # -- the policy `action_space.sample()` on the engine's action stream (vector/device_policy.py) -----------------------------------
    def _draw_action_batches(self, K: int):
        """The next K batches of ``action_space.sample()`` in one engine call (mi_action_sample); a tuple of K arrays / device tensors,
        views of ONE freshly allocated block (nothing a caller holds is ever overwritten)."""
        eng = self._engine
        if self.sample_output == "torch" and self.output == "torch":
            t = self._torch
            self._bind_stream()
            block = t.empty((K,) + self._act_shape, dtype=t.int64 if self._discrete else t.float32, device=self._tdev)
            eng.action_sample(K, block.data_ptr(), _native.MI_DEVICE)
            return block.unbind(0)
        dtype = np.int64 if self._discrete else np.float32
        block = None
        if self._engine_factory is None:  # page-locked, so that the device-to-host copy of the block runs at PCIe speed
            try:
                import torch

                block = torch.empty((K,) + self._act_shape, dtype=torch.int64 if self._discrete else torch.float32, pin_memory=True).numpy()
            except Exception:
                block = None
        if block is None:
            block = np.empty((K,) + self._act_shape, dtype=dtype)
        self._bind_stream()
        eng.action_sample(K, block, _native.MI_HOST)
        return tuple(block)

    # -- fused rollouts ---------------------------------------------------------------------------------
    def rollout(self, num_steps: int, actions=None, *, return_actions: bool = True):
        """``num_steps`` consecutive ``step()`` calls in ONE kernel launch; trajectories are time-major tensors in HBM.

        With ``actions=None`` the random policy ``action_space.sample()`` (spaces/multi_discrete.py:176-178,
        spaces/box.py:463-465) is evaluated on device from the action space's own PCG64 stream, which is then
        advanced on the host by the number of draws consumed -- so
        ``rollout(T)`` == ``[step(action_space.sample()) for _ in range(T)]`` bit for bit.
        Requires ``output="torch"``.  Returns dict(obs, rewards, terminations, truncations[, actions]).
        """
        self._check_open()
        if self.output != "torch":
            raise error.Error("rollout() returns device tensors; create the env with output='torch'")
        if not self._has_reset:
            raise AssertionError("Call reset before using rollout.")
        t, eng, N, T = self._torch, self._engine, self.num_envs, int(num_steps)
        dev = self._tdev
        self._bind_stream()
        act_dtype = t.int64 if self._discrete else t.float32
        act_shape = (T, N) if self._discrete else (T, N, eng.act_dim)
        a_in = a_out = None
        in_dtype = _native.MI_F32
        if actions is not None:
            if not self._discrete and actions.dtype == t.float64:  # float64 rows are taken un-rounded, like step() does (_coerce_actions)
                a_in, in_dtype = actions.to(device=dev).contiguous(), _native.MI_F64
            else:
                a_in = actions.to(device=dev, dtype=act_dtype).contiguous()
            if tuple(a_in.shape) != act_shape:
                raise ValueError(f"actions must have shape {act_shape}, got {tuple(a_in.shape)}")
        else:
            on_stream = self.action_space.hip_use_stream() if hasattr(self.action_space, "hip_use_stream") else None
            if on_stream is None:  # a space without the device sampler: its NumPy generator is the position
                eng.action_seed(_native.pcg_words(self.action_space.np_random))
            if return_actions:
                a_out = t.empty(act_shape, dtype=act_dtype, device=dev)
        obs = t.empty((T,) + tuple(self._obs_shape), dtype=self._obs_tdtype, device=dev)
        rew = t.empty((T, N), dtype=t.float64, device=dev)
        term = t.empty((T, N), dtype=t.bool, device=dev)
        trunc = t.empty((T, N), dtype=t.bool, device=dev)
        eng.rollout(T, None if a_in is None else a_in.data_ptr(), None if a_out is None else a_out.data_ptr(),
                    obs.data_ptr(), rew.data_ptr(), term.data_ptr(), trunc.data_ptr(), actions_in_dtype=in_dtype)
        self._act_f64 = in_dtype == _native.MI_F64
        if actions is None and on_stream is None:
            self.action_space.np_random.bit_generator.advance(T * N * eng.act_dim)
        out = {"obs": obs, "rewards": rew, "terminations": term, "truncations": trunc}
        if a_out is not None:
            out["actions"] = a_out
        elif a_in is not None and return_actions:
            out["actions"] = a_in
        # the vector env's "current" buffers follow the last step, as after T step() calls
        self._obs.copy_(obs[-1]); self._rew.copy_(rew[-1]); self._term.copy_(term[-1]); self._trunc.copy_(trunc[-1])
        if self.autoreset_mode == AutoresetMode.NEXT_STEP:  # the sub-envs that finished in the last step reset in the next one
            if self._device_infos:
                self._was_done_t.copy_(term[-1] | trunc[-1])
            else:
                self._was_done = (term[-1] | trunc[-1]).cpu().numpy()
        return out

    # -- HIP graphs ------------------------------------------------------------------------------------
    def capture_steps(self, actions=None, steps: int = 1, policy=None):
        """Capture ``steps`` consecutive ``step()`` calls -- and, with ``policy``, the policy between them -- into ONE HIP graph
        (``torch.cuda.CUDAGraph``, which on ROCm is a hipGraph) and return a :class:`GraphedSteps` whose ``replay()`` runs them with a
        single launch.  At 65 536 CartPoles a step kernel runs ~3 us and its launch from Python costs more than that: the per-step API is
        launch-bound, and a replayed graph takes the host out of the loop.

        ``actions``: a device tensor the captured steps READ at replay time (write the next actions into it with ``copy_`` before
        ``replay()``); or ``policy``: a callable ``obs -> actions`` of torch ops, captured with the steps (its first input is the current
        observation buffer), or the string ``"random"``: every captured step is ``step(action_space.sample())`` with the batch drawn inside the
        step kernel from the action stream, whose position lives on the device and moves with every replay (``step(None)``).  Capturing executes nothing: the sub-environments advance only when the graph is replayed.  What a replay does not
        do: the host-side checks of step() (an invalid action raises at the next eager call or ``synchronize()``), and the wall-clock ``t`` of
        the episode statistics (a host value, frozen at capture).  Requires output="torch" and one eager step()/reset() before (kernels load
        on first use, which a capture must not trigger).  The reference has no counterpart: its step is a Python loop (sync_vector_env.py:253-323)."""
        self._check_open()
        self._check_not_pending("capture_steps")
        if self.output != "torch" or self._engine_factory is not None:
            raise error.Error("capture_steps() needs device tensors on a GPU: create the env with output='torch'")
        if not self._has_reset:
            raise AssertionError("Call reset before using capture_steps.")
        if self.strict_actions:
            raise error.Error("strict_actions=True synchronises after every step, which a graph capture cannot contain")
        if (self.INFO_KEYS or self.autoreset_mode == AutoresetMode.SAME_STEP or self.record_episode_statistics) and not self._device_infos:
            raise error.Error("this environment assembles its infos on the host: its step() cannot be captured")
        st = self.__dict__.get("_fused")
        if st is not None and st["chain"]:
            raise error.Error("vector wrappers are fused into this env's step kernel (their running statistics double-buffer on the host side): "
                              "its step() cannot be captured")
        if (actions is None) == (policy is None):
            raise ValueError("capture_steps() takes either `actions` (a device tensor read at replay time) or `policy` (a callable, or \"random\")")
        if policy == "random" and not hasattr(self.action_space, "hip_use_stream"):
            raise error.Error("policy=\"random\" needs the device-sampled action space")
        return GraphedSteps(self, actions, int(steps), policy)

    # -- bookkeeping -----------------------------------------------------------------------------------
    def statistics(self) -> dict:
        """Running totals kept on device: env_steps, reset_steps, episodes, return_sum, length_sum."""
        return self._engine.stats()

    def reset_statistics(self):
        self._engine.reset_stats()

    def get_state(self):
        """(state[N, state_dim] float64, elapsed_steps[N] int32, flags[N] uint8) -- checkpoint of the sub-environments."""
        return self._engine.get_state()

    def set_state(self, state=None, elapsed_steps=None, flags=None):
        self._engine.set_state(state, elapsed_steps, flags)
        self._has_reset = True
        if flags is not None:  # keep the host mirror of the pending-autoreset set in step with the device flags
            self._was_done = (np.asarray(flags, dtype=np.uint8) & _native.FLAG_NEEDS_RESET) != 0
            if self._device_infos:
                self._was_done_t.copy_(self._torch.from_numpy(self._was_done.copy()))

    def get_rng_state(self) -> np.ndarray:
        """Per-env PCG64 words [N, 4] = {state_hi, state_lo, inc_hi, inc_lo}."""
        return self._engine.get_rng()

    def synchronize(self):
        self._engine.synchronize()

    def _check_not_pending(self, what: str):
        """AsyncVectorEnv's state machine (vector/async_vector_env.py:340-344, 440-452): one outstanding call at a time."""
        if getattr(self, "_async_pending", None) is not None:
            raise error.AlreadyPendingCallError(f"Calling `{what}` while waiting for a pending call to `step_async` to complete.", "step")

    def _check_open(self):
        if self.closed:
            raise error.ClosedEnvironmentError(f"Trying to operate on `{type(self).__name__}`, after a call to `close()`.")

    def close_extras(self, **kwargs):
        eng = getattr(self, "_engine", None)
        if eng is not None:
            if hasattr(self.action_space, "_hip_to_host"):  # the action space outlives the env: its NumPy generator takes the stream's position back
                try:
                    self.action_space._hip_to_host()
                except Exception:
                    pass
            if getattr(self, "_async_pending", None) is not None and self._async_pending[0] == "engine":
                try:  # a step is in flight on the pinned block: collect it before the block goes away
                    eng.step_wait()
                except Exception:
                    pass
                self._async_pending = None
            if self.output == "numpy" and getattr(self, "_pinned", False):
                # The NumPy arrays handed out with copy=False (and `action_buffer`) are VIEWS of the engine's page-locked block, which
                # mi_destroy frees: replace this object's references by ordinary copies so that reading `env._obs` / the last returned
                # batch's base after close() is not a use-after-free.  (Views a caller still holds are the caller's: documented in README.)
                for name in ("_obs", "_rew", "_term", "_trunc", "_final", "_info", "_final_info", "_ep_r", "_ep_l", "action_buffer"):
                    v = getattr(self, name, None)
                    if isinstance(v, np.ndarray):
                        setattr(self, name, v.copy())
                self._pinned = False
            eng.close()
            self._engine = None


_SHORT_STEP_OWNERS = {HipVectorEnv.step}  # step() implementations that add nothing to HipVectorEnv.step's result (subclasses register theirs)


class GraphedSteps:
    """``steps`` step() calls of a :class:`HipVectorEnv` (plus the policy between them) as one HIP graph; see ``HipVectorEnv.capture_steps``.

    ``results``: the (obs, rewards, terminations, truncations, infos) tuple of every captured step -- static device tensors that each
    ``replay()`` overwrites (with ``copy=False`` the observation / reward / flag tensors of all steps are the env's own buffers, i.e. they hold
    the LAST step's values; with ``copy=True`` every step has its own)."""

    def __init__(self, env: "HipVectorEnv", actions, steps: int, policy):
        t = env._torch
        if steps < 1:
            raise ValueError("steps must be >= 1")
        if actions is not None:
            keep, _, _ = env._coerce_actions(actions)
            if keep is not actions:
                raise ValueError("the captured steps read `actions` in place at replay time: pass a contiguous tensor on the env's device with "
                                 "the action space's dtype (int64, or float32 / float64 rows for Box spaces)")
        self.env, self.steps, self.actions = env, steps, actions
        self.graph = t.cuda.CUDAGraph()
        self.results = []
        if policy == "random":  # the per-lane states of the action stream must exist before the capture opens (mi_action_sample with T = 0)
            env._bind_stream()
            env.action_space.hip_use_stream().action_sample(0, None, _native.MI_DEVICE)
        env.synchronize()
        try:
            with t.cuda.graph(self.graph):
                obs = env._obs
                for _ in range(steps):
                    out = env.step(actions if policy is None else (None if policy == "random" else policy(obs)))
                    obs = out[0]
                    self.results.append(out)
        finally:
            env._stream_bound = None  # (the capture bound the engine to the capture stream)
            env._bind_stream()

    def replay(self):
        """Run the captured steps (one graph launch on torch's current stream, asynchronous like step() with device tensors); returns the
        last step's tuple."""
        env = self.env
        env._check_open()
        env._check_not_pending("replay")
        env._bind_stream()  # statistics() / synchronize() wait on the engine's stream: keep it the one the replay runs on
        self.graph.replay()
        return self.results[-1]


def _resolve_device(device) -> int:
    """None -> LOCAL_RANK (one process per GPU) or 0; 'cuda:3' / 3 / torch.device -> index."""
    import os

    if device is None:
        return int(os.environ.get("LOCAL_RANK", "0"))
    if isinstance(device, (int, np.integer)):
        return int(device)
    s = str(device)
    if ":" in s:
        return int(s.split(":")[1])
    if s in ("cuda", "hip"):
        return 0
    return int(s)
