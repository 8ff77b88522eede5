import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Where Farama gymnasium itself can be imported from the read-only reference tree (the build container), the WHOLE CPU suite runs under it: `HipVectorEnv`
# is then a gymnasium.vector.VectorEnv, the ids live in gymnasium's registry and tests/test_real_gymnasium.py compares against gymnasium's own
# SyncVectorEnv in this interpreter (no skips).  tests/test_mirror_configuration.py re-runs the interface tests in a child interpreter with the
# from-scratch mirror forced -- the configuration of the GPU box, where neither gymnasium nor the tree exists (every -m gpu test runs on the mirror).
REFERENCE = os.environ.get("GYMNASIUM_REFERENCE_TREE", "/root/reference")
if os.environ.get("GYMNASIUM_AMD_FORCE_MIRROR", "0") != "1" and os.path.isdir(os.path.join(REFERENCE, "gymnasium")) and REFERENCE not in sys.path:
    try:
        import gymnasium  # noqa: F401  (already installed: nothing to add)
    except ImportError:
        sys.path.append(REFERENCE)
        sys.dont_write_bytecode = True  # never write __pycache__ into the read-only tree
        os.environ["PYTHONPATH"] = os.pathsep.join(x for x in (os.environ.get("PYTHONPATH", ""), REFERENCE) if x)  # child interpreters (gloo workers, bench dry runs)
        os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

ENV_IDS = {
    "cartpole": "CartPole-v1",
    "pendulum": "Pendulum-v1",
    "acrobot": "Acrobot-v1",
    "mountaincar": "MountainCar-v0",
    "mountaincar_continuous": "MountainCarContinuous-v0",
}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no libmi355env.so (built artefacts are not tracked): build it once so that the CPU-side ABI
    tests can load it.  Only when it is MISSING -- never a rebuild on the GPU box, where the prebuilt library travels."""
    so = os.path.join(ROOT, "gymnasium_amd", "csrc", "libmi355env.so")
    if not os.path.exists(so) and os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        from gymnasium_amd.csrc import build

        build.build(verbose=False)


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def oracle_factory():
    from oracle import oracle

    oracle.load()
    return oracle.engine_factory


def has_gpu() -> bool:
    try:
        from gymnasium_amd import _native

        return _native.load_library().device_count() > 0
    except Exception:
        return False
