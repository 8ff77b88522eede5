"""ctypes binding of the C ABI declared in include/mi355env.h (libmi355env.so).

There is no CPU fallback: if the HIP library has not been built, or no MI355X is visible, creating an
environment raises.  The binding is generic over (shared object, symbol prefix) only so that the test-suite can
drive the SAME host code against a checker library; the package itself only ever loads libmi355env.so.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

MI_OK = 0
MI_HOST, MI_DEVICE = 0, 1
MI_F32, MI_F64, MI_I64 = 0, 1, 2
MI_F64_WEAK = 3  # action rows only: float64 values that were Python floats on the caller's side (weak under NEP 50)
FLAG_NEEDS_RESET, FLAG_STATE_F32 = 1, 2
CFG_SOLVER_NEWTON = 1  # MI_CFG_SOLVER_NEWTON
CFG_FAST_MATH = 2  # MI_CFG_FAST_MATH (classic control: device sin / cos and x * x instead of the libm restatements)
CFG_SHARED_RNG = 4  # MI_CFG_SHARED_RNG (CartPole: the reference's CartPoleVectorEnv semantics -- one generator for all sub-environments)
ABI_VERSION = 7

ENV_KINDS = {"cartpole": 0, "pendulum": 1, "acrobot": 2, "mountain_car": 3, "mountain_car_continuous": 4,
             "half_cheetah": 5, "ant": 6, "humanoid": 7, "tabular": 8,
             "hopper": 9, "walker2d": 10, "inverted_pendulum": 11, "inverted_double_pendulum": 12, "blackjack": 13, "reacher": 14, "humanoid_standup": 15, "swimmer": 16, "pusher": 17}
AUTORESET = {"NextStep": 0, "SameStep": 1, "Disabled": 2}
NP_DTYPES = {MI_F32: np.float32, MI_F64: np.float64, MI_I64: np.int64}

# Every symbol include/mi355env.h declares (tests/test_abi.py checks the built library exports all of them).
SYMBOLS = [
    "abi_version", "last_error", "device_count", "create", "destroy", "get_layout", "set_stream", "synchronize",
    "seed", "seed_sequence", "reset", "step", "action_seed", "rollout", "get_stats", "reset_stats", "get_state",
    "set_state", "get_rng", "tabular_load", "action_sample", "action_get", "action_skip",
]


# The device-side vector wrappers (include/mi355env.h, gymnasium_amd/csrc/wrappers.hip); their checker is NumPy code
# (oracle/wrappers.py), so they are not part of the orc_-prefixed checker ABI.
# Host stepping through the engine's pinned staging block (mi_step_async / mi_step_wait / mi_host_buffers): product library only.
HOST_SYMBOLS = ["step_async", "step_wait", "host_buffers"]
WRAPPER_SYMBOLS = ["rms_create", "rms_destroy", "rms_get", "rms_set", "normalize_observation", "normalize_reward", "clip_reward", "set_step_epilogue"]


class MiConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("kind", C.c_int32), ("num_envs", C.c_int32), ("max_episode_steps", C.c_int32),
                ("autoreset_mode", C.c_int32), ("reserved", C.c_int32 * 3), ("params", C.c_double * 16)]


class MiLayout(C.Structure):
    _fields_ = [("obs_dim", C.c_int32), ("obs_dtype", C.c_int32), ("act_dim", C.c_int32), ("act_dtype", C.c_int32),
                ("state_dim", C.c_int32), ("info_dim", C.c_int32), ("reserved", C.c_int32 * 2)]


class MiStepIO(C.Structure):
    _fields_ = [("actions", C.c_void_p), ("obs", C.c_void_p), ("reward", C.c_void_p), ("terminated", C.c_void_p),
                ("truncated", C.c_void_p), ("final_obs", C.c_void_p), ("episode_return", C.c_void_p),
                ("episode_length", C.c_void_p), ("info", C.c_void_p), ("final_info", C.c_void_p),
                ("actions_dtype", C.c_int32), ("reserved", C.c_int32),  # actions_dtype: MI_F32 (default) / MI_F64 rows of a Box action space
                ("actions_out", C.c_void_p)]  # ABI 7: with actions == NULL (the on-device policy) where the drawn actions go, or NULL


class MiRolloutIO(C.Structure):
    _fields_ = [("actions_in", C.c_void_p), ("actions_out", C.c_void_p), ("obs", C.c_void_p), ("reward", C.c_void_p),
                ("terminated", C.c_void_p), ("truncated", C.c_void_p), ("actions_in_dtype", C.c_int32), ("reserved", C.c_int32)]


class MiTabularTable(C.Structure):
    _fields_ = [("num_states", C.c_int32), ("num_actions", C.c_int32), ("max_outcomes", C.c_int32), ("num_tables", C.c_int32),
                ("csprob", C.c_void_p), ("prob", C.c_void_p), ("next_state", C.c_void_p), ("reward", C.c_void_p),
                ("terminated", C.c_void_p), ("count", C.c_void_p), ("isd_csprob", C.c_void_p), ("env_table", C.c_void_p)]


class MiStepEpilogue(C.Structure):
    """mi_step_epilogue: the stateful vector wrappers as the output stage of the step kernel (include/mi355env.h)."""
    _fields_ = [("obs_rms", C.c_void_p), ("obs_epsilon", C.c_double), ("obs_update", C.c_int32), ("reward_update", C.c_int32),
                ("return_rms", C.c_void_p), ("accumulated", C.c_void_p), ("prev_done", C.c_void_p), ("gamma", C.c_double),
                ("reward_epsilon", C.c_double), ("clip_pre", C.c_int32), ("clip_post", C.c_int32), ("clip_pre_min", C.c_double),
                ("clip_pre_max", C.c_double), ("clip_post_min", C.c_double), ("clip_post_max", C.c_double)]


class MiStats(C.Structure):
    _fields_ = [("env_steps", C.c_uint64), ("reset_steps", C.c_uint64), ("episodes", C.c_uint64), ("return_sum", C.c_double),
                ("length_sum", C.c_uint64)]


class NativeError(RuntimeError):
    """A libmi355env call failed; ``code`` is the mi_status."""

    def __init__(self, code, message):
        super().__init__(f"[mi_status {code}] {message}")
        self.code = code
        self.message = message


def library_path() -> str:
    """The in-tree build; MI355ENV_LIBRARY points at another build of the same ABI (A/B measurements of two builds on one GPU box)."""
    return os.environ.get("MI355ENV_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libmi355env.so")


class NativeLib:
    """A loaded shared object exposing the mi355env ABI under ``prefix`` (product: ``mi_``)."""

    def __init__(self, path: str, prefix: str = "mi_"):
        self.path, self.prefix = path, prefix
        self.dll = C.CDLL(path)
        f = self._fn
        vp, i32, u64 = C.c_void_p, C.c_int, C.c_uint64
        self.abi_version = f("abi_version", [], i32)
        self.last_error = f("last_error", [], C.c_char_p)
        self.device_count = f("device_count", [], i32)
        self.create = f("create", [C.POINTER(MiConfig), i32, C.POINTER(vp)], i32)
        self.destroy = f("destroy", [vp], None)
        self.get_layout = f("get_layout", [vp, C.POINTER(MiLayout)], i32)
        self.set_stream = f("set_stream", [vp, vp], i32)
        self.synchronize = f("synchronize", [vp], i32)
        self.seed = f("seed", [vp, vp, vp], i32)
        self.seed_sequence = f("seed_sequence", [vp, u64, u64, vp], i32)
        self.reset = f("reset", [vp, vp, vp, vp, i32], i32)
        self.step = f("step", [vp, C.POINTER(MiStepIO), i32], i32)
        self.action_seed = f("action_seed", [vp, vp], i32)
        self.rollout = f("rollout", [vp, i32, C.POINTER(MiRolloutIO)], i32)
        self.get_stats = f("get_stats", [vp, C.POINTER(MiStats)], i32)
        self.reset_stats = f("reset_stats", [vp], i32)
        self.get_state = f("get_state", [vp, vp, vp, vp], i32)
        self.set_state = f("set_state", [vp, vp, vp, vp], i32)
        self.get_rng = f("get_rng", [vp, vp], i32)
        self.tabular_load = f("tabular_load", [vp, C.POINTER(MiTabularTable)], i32)
        self.action_sample = f("action_sample", [vp, i32, vp, i32], i32)
        self.action_get = f("action_get", [vp, vp], i32)
        self.action_skip = f("action_skip", [vp, C.c_int64], i32)
        if self.abi_version() != ABI_VERSION:
            raise ImportError(f"{path}: ABI version {self.abi_version()} != binding version {ABI_VERSION}")
        if prefix == "mi_":
            dbl = C.c_double
            self.rms_create = f("rms_create", [i32, i32, i32, dbl, C.POINTER(vp)], i32)
            self.rms_destroy = f("rms_destroy", [vp], None)
            self.rms_get = f("rms_get", [vp, vp, vp, vp, vp], i32)
            self.rms_set = f("rms_set", [vp, vp, vp, vp, vp], i32)
            self.normalize_observation = f("normalize_observation", [vp, vp, vp, i32, i32, dbl, i32, vp], i32)
            self.normalize_reward = f("normalize_reward", [vp, vp, vp, vp, vp, vp, vp, i32, dbl, dbl, i32, i32, vp], i32)
            self.clip_reward = f("clip_reward", [i32, vp, vp, i32, vp, vp, vp], i32)
            self.set_step_epilogue = f("set_step_epilogue", [vp, C.POINTER(MiStepEpilogue)], i32)
            self.step_async = f("step_async", [vp, C.POINTER(MiStepIO)], i32)
            self.step_wait = f("step_wait", [vp], i32)
            self.host_buffers = f("host_buffers", [vp, C.POINTER(MiStepIO)], i32)

    def _fn(self, name, argtypes, restype):
        fn = getattr(self.dll, self.prefix + name)
        fn.argtypes, fn.restype = argtypes, restype
        return fn

    def check(self, rc: int):
        if rc != MI_OK:
            msg = self.last_error()
            raise NativeError(rc, msg.decode() if msg else "unknown error")


_LIB = None


def load_library() -> NativeLib:
    """Load libmi355env.so (built by ``python -m gymnasium_amd.csrc.build`` / ``__graft_entry__.build()``)."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: the HIP engine has not been built. Run `python -m gymnasium_amd.csrc.build` "
                "(needs hipcc, --offload-arch=gfx950). gymnasium_amd has no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's).  If it is
        # going to be used at all it must be the copy that gets loaded first, so that libmi355env.so binds to it too;
        # two HIP runtimes in one process cannot both open the GPU ("No HIP GPUs are available").
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _LIB = NativeLib(path, "mi_")
    return _LIB


def _ptr(a):
    """Raw address of a NumPy array / int address / None."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    return a.ctypes.data


def pcg_words(gen: np.random.Generator) -> np.ndarray:
    """{state_hi, state_lo, inc_hi, inc_lo} of a NumPy PCG64 generator."""
    st = gen.bit_generator.state
    if st["bit_generator"] != "PCG64":
        raise ValueError(f"expected a PCG64 generator, got {st['bit_generator']}")
    s, i, m = st["state"]["state"], st["state"]["inc"], (1 << 64) - 1
    return np.array([s >> 64, s & m, i >> 64, i & m], dtype=np.uint64)


def set_pcg_words(gen: np.random.Generator, words) -> None:
    st = gen.bit_generator.state
    st["state"]["state"] = (int(words[0]) << 64) | int(words[1])
    st["state"]["inc"] = (int(words[2]) << 64) | int(words[3])
    st["has_uint32"], st["uinteger"] = 0, 0
    gen.bit_generator.state = st


class Engine:
    """One mi_vecenv handle.  Thin, allocation-free wrappers; pointers are NumPy arrays or raw device addresses."""

    def __init__(self, lib: NativeLib, kind: str, num_envs: int, max_episode_steps: int | None, autoreset_mode: str,
                 params=(), device: int = 0, options: int = 0):
        self.lib = lib
        cfg = MiConfig()
        cfg.reserved[0] = int(options)  # MI_CFG_* bits (include/mi355env.h)
        cfg.struct_size = C.sizeof(MiConfig)
        cfg.kind = ENV_KINDS[kind]
        cfg.num_envs = int(num_envs)
        cfg.max_episode_steps = int(max_episode_steps) if max_episode_steps else 0
        cfg.autoreset_mode = AUTORESET[autoreset_mode]
        for k, p in enumerate(params):
            cfg.params[k] = float(p)
        handle = C.c_void_p()
        lib.check(lib.create(C.byref(cfg), int(device), C.byref(handle)))
        self.handle = handle
        self.num_envs = int(num_envs)
        lay = MiLayout()
        lib.check(lib.get_layout(handle, C.byref(lay)))
        self.obs_dim, self.act_dim, self.state_dim, self.info_dim = lay.obs_dim, lay.act_dim, lay.state_dim, lay.info_dim
        self.obs_dtype, self.act_dtype = NP_DTYPES[lay.obs_dtype], NP_DTYPES[lay.act_dtype]
        self._step_io = MiStepIO()
        self._rollout_io = MiRolloutIO()

    def close(self):
        if self.handle is not None:
            self.lib.destroy(self.handle)
            self.handle = None

    def set_stream(self, stream_ptr):
        self.lib.check(self.lib.set_stream(self.handle, stream_ptr))

    def set_step_epilogue(self, epilogue: "MiStepEpilogue | None"):
        """Attach (or, with None, detach) the wrappers' arithmetic as the output stage of the step kernel (mi_set_step_epilogue)."""
        self.lib.check(self.lib.set_step_epilogue(self.handle, None if epilogue is None else C.byref(epilogue)))

    def synchronize(self):
        self.lib.check(self.lib.synchronize(self.handle))

    def seed(self, words: np.ndarray, mask=None):
        words = np.ascontiguousarray(words, dtype=np.uint64)
        assert words.shape == (self.num_envs, 4)
        self.lib.check(self.lib.seed(self.handle, _ptr(words), _ptr(mask)))

    def seed_sequence(self, base_seed: int, first_index: int = 0, mask=None):
        self.lib.check(self.lib.seed_sequence(self.handle, base_seed, first_index, _ptr(mask)))

    def reset(self, mask, bounds, obs, loc=MI_HOST):
        b = None if bounds is None else np.ascontiguousarray(bounds, dtype=np.float64)
        self.lib.check(self.lib.reset(self.handle, _ptr(mask), _ptr(b), _ptr(obs), loc))

    def step(self, actions, obs, reward, terminated, truncated, final_obs=None, episode_return=None,
             episode_length=None, loc=MI_HOST, info=None, final_info=None, actions_dtype=MI_F32, actions_out=None):
        io = self._step_io
        io.final_info, io.actions_dtype, io.actions_out = _ptr(final_info), int(actions_dtype), _ptr(actions_out)
        io.actions, io.obs, io.reward = _ptr(actions), _ptr(obs), _ptr(reward)
        io.terminated, io.truncated, io.final_obs = _ptr(terminated), _ptr(truncated), _ptr(final_obs)
        io.episode_return, io.episode_length, io.info = _ptr(episode_return), _ptr(episode_length), _ptr(info)
        self.lib.check(self.lib.step(self.handle, C.byref(io), loc))

    def bind_step(self, obs, reward, terminated, truncated, final_obs=None, episode_return=None, episode_length=None, loc=MI_HOST, info=None,
                  final_info=None):
        """Fix the output addresses of the step_bound() calls that follow: a vector env steps into the same buffers thousands of times, and
        at 65 536 CartPoles filling eleven struct fields from Python costs more than the kernel runs.  The caller keeps the buffers alive."""
        io = MiStepIO()
        io.obs, io.reward, io.terminated, io.truncated, io.final_obs = _ptr(obs), _ptr(reward), _ptr(terminated), _ptr(truncated), _ptr(final_obs)
        io.episode_return, io.episode_length, io.info, io.final_info = _ptr(episode_return), _ptr(episode_length), _ptr(info), _ptr(final_info)
        self._bound = (io, C.byref(io), int(loc), self.lib.step, self.handle)

    def step_bound(self, actions: "int | None", actions_dtype: int, actions_out: "int | None" = None):
        """mi_step into the buffers of bind_step(); `actions` is a raw address, or None: the on-device policy (the step kernel draws
        action_space.sample() from the action stream; `actions_out`: where the drawn batch goes, or None)."""
        io, ref, loc, fn, handle = self._bound
        io.actions, io.actions_dtype, io.actions_out = actions, actions_dtype, actions_out
        rc = fn(handle, ref, loc)
        if rc:
            self.lib.check(rc)

    def _fill_io(self, actions, obs, reward, terminated, truncated, final_obs, episode_return, episode_length, info, final_info, actions_dtype=MI_F32):
        io = self._step_io
        io.actions_dtype, io.actions_out = int(actions_dtype), None
        io.actions, io.obs, io.reward = _ptr(actions), _ptr(obs), _ptr(reward)
        io.terminated, io.truncated, io.final_obs = _ptr(terminated), _ptr(truncated), _ptr(final_obs)
        io.episode_return, io.episode_length, io.info, io.final_info = _ptr(episode_return), _ptr(episode_length), _ptr(info), _ptr(final_info)
        return io

    def step_async(self, actions, obs, reward, terminated, truncated, final_obs=None, episode_return=None, episode_length=None, info=None,
                   final_info=None, actions_dtype=MI_F32):
        """Enqueue a host step (H2D, kernel, one D2H) without waiting; the arrays are filled by step_wait()."""
        io = self._fill_io(actions, obs, reward, terminated, truncated, final_obs, episode_return, episode_length, info, final_info, actions_dtype)
        self.lib.check(self.lib.step_async(self.handle, C.byref(io)))

    def step_wait(self):
        self.lib.check(self.lib.step_wait(self.handle))

    def host_buffers(self):
        """The engine's pinned host arrays as NumPy views (valid until close()), or None for a backend without them (the checker)."""
        if not hasattr(self.lib, "host_buffers"):
            return None
        io = MiStepIO()
        self.lib.check(self.lib.host_buffers(self.handle, C.byref(io)))
        N = self.num_envs

        def view(ptr, ctype, dtype, count, shape):
            return np.ctypeslib.as_array((ctype * count).from_address(ptr)).view(dtype).reshape(shape)

        oct_, odt = {np.float32: (C.c_float, np.float32), np.float64: (C.c_double, np.float64), np.int64: (C.c_int64, np.int64)}[self.obs_dtype]
        act_ct, act_dt = (C.c_int64, np.int64) if self.act_dtype is np.int64 else (C.c_float, np.float32)
        obs_shape = (N,) if (self.obs_dtype is np.int64 and self.obs_dim == 1) else (N, self.obs_dim)
        act_shape = (N,) if self.act_dtype is np.int64 else (N, self.act_dim)
        out = {"actions": view(io.actions, act_ct, act_dt, N * self.act_dim, act_shape),
               "obs": view(io.obs, oct_, odt, N * self.obs_dim, obs_shape), "final_obs": view(io.final_obs, oct_, odt, N * self.obs_dim, obs_shape),
               "reward": view(io.reward, C.c_double, np.float64, N, (N,)), "episode_return": view(io.episode_return, C.c_double, np.float64, N, (N,)),
               "terminated": view(io.terminated, C.c_uint8, np.bool_, N, (N,)), "truncated": view(io.truncated, C.c_uint8, np.bool_, N, (N,)),
               "episode_length": view(io.episode_length, C.c_int32, np.int32, N, (N,))}
        if self.info_dim:
            out["info"] = view(io.info, C.c_double, np.float64, N * self.info_dim, (N, self.info_dim))
            out["final_info"] = view(io.final_info, C.c_double, np.float64, N * self.info_dim, (N, self.info_dim))
        return out

    def action_seed(self, words):
        w = np.ascontiguousarray(words, dtype=np.uint64)
        self.lib.check(self.lib.action_seed(self.handle, _ptr(w)))

    def action_sample(self, T: int, out, loc=MI_HOST):
        """The next T batches of action_space.sample() into out[T][N][act_dim] (mi_action_sample); T = 0 prepares the per-lane stream states."""
        self.lib.check(self.lib.action_sample(self.handle, int(T), _ptr(out), loc))

    def action_get(self) -> np.ndarray:
        """{state_hi, state_lo, inc_hi, inc_lo} of the generator that produces the action stream's next draw (mi_action_get)."""
        w = np.empty(4, dtype=np.uint64)
        self.lib.check(self.lib.action_get(self.handle, _ptr(w)))
        return w

    def action_skip(self, draws: int):
        self.lib.check(self.lib.action_skip(self.handle, int(draws)))

    def rollout(self, T, actions_in=None, actions_out=None, obs=None, reward=None, terminated=None, truncated=None, actions_in_dtype=MI_F32):
        io = self._rollout_io
        io.actions_in_dtype = int(actions_in_dtype)
        io.actions_in, io.actions_out, io.obs = _ptr(actions_in), _ptr(actions_out), _ptr(obs)
        io.reward, io.terminated, io.truncated = _ptr(reward), _ptr(terminated), _ptr(truncated)
        self.lib.check(self.lib.rollout(self.handle, int(T), C.byref(io)))

    def stats(self) -> dict:
        st = MiStats()
        self.lib.check(self.lib.get_stats(self.handle, C.byref(st)))
        return {k: getattr(st, k) for k, _ in MiStats._fields_}

    def reset_stats(self):
        self.lib.check(self.lib.reset_stats(self.handle))

    def get_state(self):
        state = np.empty((self.num_envs, self.state_dim), dtype=np.float64)
        elapsed = np.empty(self.num_envs, dtype=np.int32)
        flags = np.empty(self.num_envs, dtype=np.uint8)
        self.lib.check(self.lib.get_state(self.handle, _ptr(state), _ptr(elapsed), _ptr(flags)))
        return state, elapsed, flags

    def set_state(self, state=None, elapsed=None, flags=None):
        if state is not None:
            state = np.ascontiguousarray(state, dtype=np.float64)
            assert state.shape == (self.num_envs, self.state_dim)
        if elapsed is not None:
            elapsed = np.ascontiguousarray(elapsed, dtype=np.int32)
        if flags is not None:
            flags = np.ascontiguousarray(flags, dtype=np.uint8)
        self.lib.check(self.lib.set_state(self.handle, _ptr(state), _ptr(elapsed), _ptr(flags)))

    def load_table(self, csprob, prob, next_state, reward, terminated, count, isd_csprob, env_table=None):
        """Hand a finite MDP's transition table to the engine (mi_tabular_load): arrays [nS, nA, K], or -- with ``env_table`` [num_envs] naming each
        sub-environment's table -- [num_tables, nS, nA, K]."""
        t = MiTabularTable()
        if env_table is None:
            t.num_states, t.num_actions, t.max_outcomes = csprob.shape
            t.num_tables, t.env_table = 1, None
        else:
            t.num_tables, t.num_states, t.num_actions, t.max_outcomes = csprob.shape
            env_table = np.ascontiguousarray(env_table, np.int32)
            assert env_table.shape == (self.num_envs,) and t.num_tables > 1
            t.env_table = env_table.ctypes.data
        keep = [np.ascontiguousarray(csprob, np.float64), np.ascontiguousarray(prob, np.float64), np.ascontiguousarray(next_state, np.int32),
                np.ascontiguousarray(reward, np.float64), np.ascontiguousarray(terminated, np.uint8), np.ascontiguousarray(count, np.int32),
                np.ascontiguousarray(isd_csprob, np.float64)]
        t.csprob, t.prob, t.next_state, t.reward, t.terminated, t.count, t.isd_csprob = [a.ctypes.data for a in keep]
        self.lib.check(self.lib.tabular_load(self.handle, C.byref(t)))

    def get_rng(self) -> np.ndarray:
        words = np.empty((self.num_envs, 4), dtype=np.uint64)
        self.lib.check(self.lib.get_rng(self.handle, _ptr(words)))
        return words
