"""Loader for the CPU oracle (oracle/liboracle.so, built from oracle/classic_control.c).

TEST INFRASTRUCTURE: the oracle exports the same C ABI as libmi355env.so under the ``orc_`` prefix, so the
checker can be driven through the very same host code (gymnasium_amd._native.Engine / HipVectorEnv) as the
product.  The product never imports this module.
"""
import os
import subprocess

from gymnasium_amd import _native

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
_LIB = None


def build(force: bool = False) -> str:
    src = [os.path.join(HERE, f) for f in ("classic_control.c", "mujoco_core.c", "mujoco_core.h", "mujoco_api.c", "mujoco_envs.c", "mujoco_envs.h", "pcg64.h", "Makefile")] + [os.path.join(HERE, "..", "include", "mi355env.h")]
    stale = force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src if os.path.exists(s))
    if stale:
        subprocess.run(["make", "-C", HERE, "-s", "-B"], check=True)
    return LIB_PATH


def load() -> _native.NativeLib:
    global _LIB
    if _LIB is None:
        _LIB = _native.NativeLib(build(), "orc_")
        _register_mujoco_models(_LIB)
    return _LIB


def _register_mujoco_models(lib):
    """Hand the compiled robot models (gymnasium_amd/envs/mujoco/compiler.py) to the oracle's MuJoCo-pipeline restatement."""
    import ctypes

    from gymnasium_amd.envs.mujoco import compiler
    from oracle import mujoco as omj

    fn = lib.dll.orc_mj_register_model
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    for which, name in enumerate(("half_cheetah", "ant", "humanoid", "hopper", "walker2d", "inverted_pendulum", "inverted_double_pendulum", "reacher", "humanoid_standup", "swimmer", "pusher")):
        blob = omj.model_blob(compiler.compile_model(name))
        rc = fn(which, blob.ctypes.data, len(blob))
        assert rc == 0, f"oracle rejected the {name} model blob (rc={rc})"


def engine_factory(kind, num_envs, max_episode_steps, autoreset_mode, params, device, options=0):
    """Drop-in for HipVectorEnv(_engine_factory=...): the oracle behind the product's host class."""
    return _native.Engine(load(), kind, num_envs, max_episode_steps, autoreset_mode, params, device, options=options)
