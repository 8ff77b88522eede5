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
    src = [os.path.join(HERE, f) for f in ("classic_control.c", "pcg64.h", "Makefile")] + [os.path.join(HERE, "..", "include", "mi355env.h")]
    stale = force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src if os.path.exists(s))
    if stale:
        subprocess.run(["make", "-C", HERE, "-s", "-B"], check=True)
    return LIB_PATH


def load() -> _native.NativeLib:
    global _LIB
    if _LIB is None:
        _LIB = _native.NativeLib(build(), "orc_")
    return _LIB


def engine_factory(kind, num_envs, max_episode_steps, autoreset_mode, params, device):
    """Drop-in for HipVectorEnv(_engine_factory=...): the oracle behind the product's host class."""
    return _native.Engine(load(), kind, num_envs, max_episode_steps, autoreset_mode, params, device)
