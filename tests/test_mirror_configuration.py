"""The GPU box's configuration on the CPU: no gymnasium, the from-scratch interface mirror (gymnasium_amd/mirror/).  tests/conftest.py runs the suite under the
real gymnasium wherever the reference tree exists; this re-runs the tests that exercise the plug-in interface (registration, spaces, VectorEnv base class,
info dicts, seeding) in a child interpreter with GYMNASIUM_AMD_FORCE_MIRROR=1 and requires that they RAN."""
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

FILES = ["tests/test_oracle_golden.py", "tests/test_device_infos.py", "tests/test_abi.py"]


@pytest.mark.skipif(os.environ.get("GYMNASIUM_AMD_FORCE_MIRROR", "0") == "1", reason="already on the mirror")
def test_interface_tests_on_the_mirror():
    env = dict(os.environ, GYMNASIUM_AMD_FORCE_MIRROR="1", PYTHONDONTWRITEBYTECODE="1")
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, "-m", "pytest", *FILES, "-q", "-x", "-m", "not gpu", "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 40, tail
    probe = subprocess.run([sys.executable, "-c", "import gymnasium_amd.gym_api as g; print(g.HAVE_GYMNASIUM, g.VectorEnv.__module__)"], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=300)
    assert probe.stdout.split() == ["False", "gymnasium_amd.mirror.vector_env"], probe.stdout + probe.stderr


def test_mirror_make_vec_contract():
    """The mirror's make_vec against the contract of SURVEY.md section 8(b) (envs/registration.py:829-988 is what a user has otherwise)."""
    from gymnasium_amd.mirror import error, registration as reg
    from gymnasium_amd.mirror.vector_env import AutoresetMode

    made = []

    class Dummy:
        metadata = {"autoreset_mode": AutoresetMode.NEXT_STEP}

        def __init__(self, num_envs, **kw):
            self.num_envs, self.kw, self.unwrapped, self.spec = num_envs, kw, self, None
            made.append(self)

    reg.register("Tests/Dummy-v3", vector_entry_point=Dummy, max_episode_steps=77, kwargs={"gravity": 9.8})
    try:
        env = reg.make_vec("Tests/Dummy-v3", num_envs=5, length=2.0)
        assert env.num_envs == 5 and env.kw == {"gravity": 9.8, "length": 2.0, "max_episode_steps": 77}
        assert env.spec.kwargs == {"gravity": 9.8, "length": 2.0, "max_episode_steps": 77, "num_envs": 5, "vectorization_mode": "vector_entry_point"}
        assert reg.registry["Tests/Dummy-v3"].kwargs == {"gravity": 9.8}  # the registry's spec is not touched
        again = reg.make_vec(env.spec, gravity=1.6)  # a recorded spec rebuilds the same env; call-site keywords still override
        assert again.num_envs == 5 and again.kw == {"gravity": 1.6, "length": 2.0, "max_episode_steps": 77}
        assert reg.make_vec("Tests/Dummy-v3", max_episode_steps=9).kw["max_episode_steps"] == 9 and made[-1].spec.kwargs.get("num_envs") is None
        for bad, exc, match in ((dict(vectorization_mode="sync"), error.Error, "not provided"), (dict(vectorization_mode="turbo"), ValueError, "Invalid vectorization mode"),
                                (dict(vector_kwargs={"copy": False}), error.Error, "vector_kwargs"), (dict(wrappers=[object]), error.Error, "wrappers")):
            with pytest.raises(exc, match=match):
                reg.make_vec("Tests/Dummy-v3", **bad)
        with pytest.raises(error.Error, match="Invalid id type"):
            reg.make_vec(3)
        with pytest.raises(error.VersionNotFound):
            reg.make_vec("Tests/Dummy-v4")
        with pytest.raises(error.NameNotFound):
            reg.make_vec("Tests/Nothing-v0")
    finally:
        reg.registry.pop("Tests/Dummy-v3", None)
