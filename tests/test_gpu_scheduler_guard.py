"""-m gpu: the shipped cooperative physics kernels (physics16.hip / physics32.hip, compiled with LLVM's iterative GCN scheduler and the
MachineLICM settings of gymnasium_amd/csrc/build.py TU_FLAGS) -- and, since round 4, the classic-control unit (classic.hip, max-ILP scheduler) --
against the SAME sources under hipcc's defaults (libmi355env_ref.so, built next to the product by __graft_entry__.build()).

Why this is a test: the iterative schedulers were measured to MISCOMPILE the 16-lane instantiation when the RK4 stage update is inlined
(every environment differs after one sub-step); keeping `rk4_stage` out of line makes all instantiations
bit-identical to the default scheduler's output; in round 2 the MachineLICM sinking flag turned the 16-lane library kernels wrong (Ant: NaNs) while
the stand-alone harness stayed correct -- this test is what caught it.  A compiler update, a source change or a new flag can silently bring that back, and a
tolerance-level parity test might not notice -- so the two builds must agree on EVERY BIT of the trajectory and of the final state after
25 env-steps x 4096 sub-environments (Ant, HalfCheetah, Humanoid and HumanoidStandup with both of their solvers).

Each build runs in its own child process (MI355ENV_LIBRARY selects the library): two HIP libraries carrying the same kernels in one
long-lived process was measured (round 2) to abort the runtime.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gymnasium_amd", "csrc")
REF = os.path.join(CSRC, "libmi355env_ref.so")

CHILD = r"""
import sys
sys.path.insert(0, {root!r})
import numpy as np
import gymnasium_amd
from gymnasium_amd import _native
assert _native.load_library().path == {lib!r}
env = gymnasium_amd.make_vec({env_id!r}, num_envs=4096, **{kw!r})
obs, _ = env.reset(seed=17)
env.action_space.seed(3)
out = [obs]
for t in range(25):
    o, r, te, tr, _ = env.step(env.action_space.sample())
    out += [o, r, te, tr]
st, el, fl = env.get_state()
np.savez({path!r}, *out, st, el, fl)
env.close()
"""


def run_build(lib, env_id, kw, path):
    env = dict(os.environ, MI355ENV_LIBRARY=lib)
    p = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, lib=lib, env_id=env_id, kw=kw, path=path)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return np.load(path)


@pytest.mark.parametrize("env_id,solver", [("Ant-v5", None), ("HalfCheetah-v5", None), ("Hopper-v5", None), ("Walker2d-v5", None), ("Humanoid-v5", "PGS"), ("HumanoidStandup-v5", "PGS"),
                                           ("Humanoid-v5", "Newton"), ("HumanoidStandup-v5", "Newton"),
                                           # round 4: episodes that END inside the window -- the kernel's early exit for a resetting sub-environment
                                           # (half of a Humanoid wavefront retires while the other half keeps running) and the reset glue
                                           ("Humanoid-v5", "resets"), ("Ant-v5", "resets"), ("Walker2d-v5", "resets"),
                                           # round 4: the classic-control unit (classic.hip, max-ILP scheduler) against its default-scheduler twin
                                           ("CartPole-v1", "classic"), ("Pendulum-v1", "classic"), ("Acrobot-v1", "classic"), ("MountainCarContinuous-v0", "classic")])
def test_iterative_scheduler_build_is_bit_identical_to_default_scheduler_build(env_id, solver, tmp_path):
    assert os.path.exists(REF), f"{REF} missing: run __graft_entry__.build() (python -m gymnasium_amd.csrc.build --ref)"
    kw = {} if env_id in ("HalfCheetah-v5", "HumanoidStandup-v5") else dict(terminate_when_unhealthy=False)  # keep every env stepping real physics
    if solver == "classic":
        kw, solver = {}, None
    if solver == "resets":
        kw, solver = ({"max_episode_steps": 9} if env_id == "Ant-v5" else {}), None  # Humanoid / Walker2d fall within ~20 steps; the Ant is cut by the TimeLimit
    if solver:
        kw["solver"] = solver  # both shipped instantiations of the 32-lane kernel: the MJCF's PGS / 50 and the opt-in Newton
    a = run_build(os.path.join(CSRC, "libmi355env.so"), env_id, kw, str(tmp_path / "product.npz"))
    b = run_build(REF, env_id, kw, str(tmp_path / "reference.npz"))
    assert a.files == b.files and len(a.files) == 1 + 4 * 25 + 3
    for k in a.files:
        assert np.array_equal(a[k], b[k]), f"{env_id} {solver}: array {k} differs between the two builds ({int((a[k] != b[k]).sum())} entries)"
