import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

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
