"""-m gpu: the shipped cooperative physics kernels (physics16.hip / physics32.hip, compiled with LLVM's iterative GCN scheduler --
gymnasium_amd/csrc/build.py TU_FLAGS) against the SAME sources under hipcc's default scheduler (libmi355env_ref.so, built next to the
product by __graft_entry__.build()).

Why this is a test: the iterative schedulers were measured to MISCOMPILE the 16-lane instantiation when the RK4 stage update is inlined
(every environment differs after one sub-step); keeping `rk4_stage` out of line makes all four instantiations
bit-identical to the default scheduler's output.  A compiler update, a source change or a new flag can silently bring that back, and a
tolerance-level parity test might not notice -- so the two builds must agree on EVERY BIT of the state after 25 env-steps x 4096
sub-environments (Ant, HalfCheetah, Humanoid, HumanoidStandup: 100 / 25 / 100 / 100 forward passes each).
"""
import os

import numpy as np
import pytest

import gymnasium_amd
from gymnasium_amd import _native

pytestmark = pytest.mark.gpu
REF = os.path.join(os.path.dirname(os.path.abspath(_native.__file__)), "csrc", "libmi355env_ref.so")
_REF_LIB = None


def ref_factory(kind, num_envs, max_episode_steps, autoreset_mode, params, device, options=0):
    global _REF_LIB
    if _REF_LIB is None:
        _REF_LIB = _native.NativeLib(REF, "mi_")
    return _native.Engine(_REF_LIB, kind, num_envs, max_episode_steps, autoreset_mode, params, device, options=options)


@pytest.mark.parametrize("env_id,solver", [("Ant-v5", None), ("HalfCheetah-v5", None), ("Humanoid-v5", "PGS"), ("HumanoidStandup-v5", "PGS"),
                                           ("Humanoid-v5", "Newton"), ("HumanoidStandup-v5", "Newton")])
def test_iterative_scheduler_build_is_bit_identical_to_default_scheduler_build(env_id, solver):
    assert os.path.exists(REF), f"{REF} missing: run __graft_entry__.build() (python -m gymnasium_amd.csrc.build --ref)"
    n, T = 4096, 25
    kw = {} if env_id in ("HalfCheetah-v5", "HumanoidStandup-v5") else dict(terminate_when_unhealthy=False)  # keep every env stepping real physics
    if solver:
        kw["solver"] = solver  # both shipped instantiations of the 32-lane kernel: the MJCF's PGS / 50 and the opt-in Newton
    a = gymnasium_amd.make_vec(env_id, num_envs=n, **kw)
    b = gymnasium_amd.make_vec(env_id, num_envs=n, _engine_factory=ref_factory, **kw)
    assert a._engine.lib.path != b._engine.lib.path
    oa, _ = a.reset(seed=17)
    ob, _ = b.reset(seed=17)
    assert np.array_equal(oa, ob)
    a.action_space.seed(3)
    for t in range(T):
        act = a.action_space.sample()
        ra, rb = a.step(act), b.step(act)
        for x, y, what in zip(ra[:4], rb[:4], ("obs", "reward", "terminated", "truncated")):
            assert np.array_equal(x, y), f"{env_id}: {what} differs between the two builds at step {t} ({int((np.asarray(x) != np.asarray(y)).sum())} entries)"
    sa, sb = a.get_state(), b.get_state()
    assert all(np.array_equal(x, y) for x, y in zip(sa, sb)), f"{env_id}: final state differs"
    a.close(), b.close()
