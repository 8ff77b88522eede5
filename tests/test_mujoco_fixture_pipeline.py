"""The `mujoco` fixture pipeline, executed end to end on a stand-in -- so that it works first time the day a real wheel appears.

tests/fake_mujoco/ is a `mujoco` look-alike backed by THIS repo's CPU oracle (MjModel.from_xml_path, MjData, mj_forward, mj_step,
mj_resetData, mj_rnePostConstraint, data.body(...).xpos, ...).  With it in front of sys.path, in child interpreters:

  1. tests/golden/make_mujoco_golden.py runs unchanged -- it drives the REFERENCE's own env classes (gymnasium/envs/mujoco/*_v5.py from
     /root/reference) -- and writes its eleven fixture files into a scratch directory (never into tests/golden/: the script refuses);
  2. tests/test_mujoco_fixtures.py consumes them: all 45 CPU cases must RUN (none skipped) and pass;
  3. the reference env classes (Python glue on the oracle's physics) and gymnasium_amd's vector env (the C glue of oracle/mujoco_envs.c, which
     the HIP glue is tested against) are stepped side by side: observations, rewards, termination flags and the info entries must be
     EQUAL -- the reference's reward / observation / health / reset-noise code pinned on the reference itself, for all eleven robots.

What this is NOT: a pin of the physics.  Both sides of every comparison here share oracle/mujoco_core.c; DESIGN.md section 7's "parity
unpinned" stands until step 1 runs on a real `mujoco`.  Skipped where the reference tree is absent (the GPU box)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get("GYM_REFERENCE", "/root/reference")
FAKE = os.path.join(ROOT, "tests", "fake_mujoco")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "gymnasium")), reason="needs the reference tree (its env classes drive the stand-in)")


def _env(**extra):
    e = dict(os.environ, PYTHONPATH=os.pathsep.join([FAKE, REFERENCE, ROOT]), PYTHONDONTWRITEBYTECODE="1", GYM_REFERENCE=REFERENCE)
    e.pop("GYMNASIUM_AMD_FORCE_MIRROR", None)
    e.update(extra)
    return e


def test_generator_and_all_consumer_cases_run_end_to_end(tmp_path):
    out = tmp_path / "fixtures"
    out.mkdir()
    g = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_mujoco_golden.py")], env=_env(MUJOCO_GOLDEN_OUT=str(out)),
                       cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert g.returncode == 0, g.stdout[-2000:] + g.stderr[-3000:]
    assert len(list(out.glob("mujoco_*.npz"))) == 11 and "oracle.shim" in g.stdout
    # ... and it must refuse to put stand-in fixtures where real ones belong
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_mujoco_golden.py")], env=_env(), cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 3 and "refusing" in r.stdout
    c = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_mujoco_fixtures.py"), "-q", "-m", "not gpu", "-p", "no:cacheprovider"],
                       env=dict(os.environ, MUJOCO_GOLDEN_DIR=str(out)), cwd=ROOT, capture_output=True, text=True, timeout=1200)
    tail = (c.stdout + c.stderr)[-3000:]
    assert c.returncode == 0, tail
    m = re.search(r"(\d+) passed", c.stdout)
    assert m and int(m.group(1)) >= 45 and "skipped" not in c.stdout.splitlines()[-1], tail


GLUE_CHECK = r'''
import sys
import numpy as np
import mujoco, gymnasium as gym          # the stand-in and the reference
import gymnasium_amd
from oracle import oracle
assert getattr(mujoco, "IS_ORACLE_SHIM", False)
IDS = ["HalfCheetah-v5", "Ant-v5", "Humanoid-v5", "HumanoidStandup-v5", "Hopper-v5", "Walker2d-v5", "InvertedPendulum-v5",
       "InvertedDoublePendulum-v5", "Reacher-v5", "Swimmer-v5", "Pusher-v5"]
for env_id in IDS:
    ref = gym.make(env_id, max_episode_steps=40)           # TimeLimit(OrderEnforcing(PassiveEnvChecker(<reference env class>)))
    ours = gym.make_vec("MI355X/" + env_id, num_envs=1, max_episode_steps=40, _engine_factory=oracle.engine_factory)
    o_ref, i_ref = ref.reset(seed=11)
    o, i = ours.reset(seed=11)
    assert np.array_equal(o[0], o_ref), (env_id, "reset observation", np.abs(o[0] - o_ref).max())
    for k, v in i_ref.items():
        assert np.allclose(np.asarray(i[k][0], dtype=np.float64), np.asarray(v, dtype=np.float64), rtol=0, atol=0), (env_id, "reset info", k)
    ref.action_space.seed(3)
    steps = episodes = 0
    for t in range(150):
        a = ref.action_space.sample()
        o_ref, r_ref, te_ref, tr_ref, i_ref = ref.step(a)
        o, r, te, tr, i = ours.step(a[None])
        assert np.array_equal(o[0], o_ref), (env_id, t, "observation", np.abs(o[0] - o_ref).max())
        assert r[0] == r_ref and bool(te[0]) == bool(te_ref) and bool(tr[0]) == bool(tr_ref), (env_id, t, r[0], r_ref, te, te_ref, tr, tr_ref)
        for k, v in i_ref.items():
            if isinstance(v, (int, float, np.floating, np.ndarray)):
                assert np.array_equal(np.asarray(i[k][0], dtype=np.float64), np.asarray(v, dtype=np.float64)), (env_id, t, "info", k, i[k][0], v)
        steps += 1
        if te_ref or tr_ref:                                 # NEXT_STEP autoreset on our side == an explicit reset of the scalar env
            episodes += 1
            o_ref, _ = ref.reset()
            o, *_ = ours.step(a[None])
            assert np.array_equal(o[0], o_ref), (env_id, t, "autoreset observation")
    print(env_id, "ok:", steps, "steps,", episodes, "episodes")
print("GLUE_OK")
'''


VECTOR_CHECK = r'''
import numpy as np
import mujoco, gymnasium as gym
from gymnasium.utils.env_checker import data_equivalence
import gymnasium_amd
from oracle import oracle
assert getattr(mujoco, "IS_ORACLE_SHIM", False)
for env_id in ["HalfCheetah-v5", "Ant-v5", "Humanoid-v5", "HumanoidStandup-v5", "Hopper-v5", "Walker2d-v5", "InvertedPendulum-v5",
               "InvertedDoublePendulum-v5", "Reacher-v5", "Swimmer-v5", "Pusher-v5"]:
    for mode in ["NextStep", "SameStep", "Disabled"]:
        n = 3
        ref = gym.make_vec(env_id, num_envs=n, vectorization_mode="sync", vector_kwargs={"autoreset_mode": mode}, max_episode_steps=25)
        ours = gym.make_vec("MI355X/" + env_id, num_envs=n, autoreset_mode=mode, max_episode_steps=25, _engine_factory=oracle.engine_factory)
        r1, r2 = ours.reset(seed=5), ref.reset(seed=5)
        assert data_equivalence(r1, r2, exact=True), (env_id, mode, "reset")
        ref.action_space.seed(2)
        dones = 0
        for t in range(70):
            a = ref.action_space.sample()
            if t % 3 == 2:  # a float64 action batch: handed to the sub-environments un-rounded (sync_vector_env.py:274), float64 control cost / info dtypes
                a = a.astype(np.float64) * 0.7
            s1, s2 = ours.step(a), ref.step(a)
            for k, what in enumerate(("obs", "reward", "terminated", "truncated")):
                assert data_equivalence(s1[k], s2[k], exact=True), (env_id, mode, t, what)
            assert data_equivalence(dict(s1[4]), dict(s2[4]), exact=True), (env_id, mode, t, "infos", sorted(s1[4]), sorted(s2[4]))
            d = s2[2] | s2[3]
            dones += int(d.sum())
            if mode == "Disabled" and d.any():
                assert data_equivalence(ours.reset(options={"reset_mask": d}), ref.reset(options={"reset_mask": d}), exact=True), (env_id, t, "masked reset")
        assert dones > 0
        ours.close(), ref.close()

    print(env_id, "vector ok")
print("VECTOR_OK")
'''


def test_reference_sync_vector_env_equals_ours_for_the_mujoco_kinds():
    """gymnasium's own SyncVectorEnv over its own env classes (stand-in physics) vs the engine's host class: every step's observations,
    rewards, flags and the WHOLE infos dict (keys, masks, dtypes, final_obs / final_info under SAME_STEP) under the reference's strict
    data_equivalence, in all three autoreset modes -- the vectorised info assembly of the MuJoCo kinds pinned on the reference."""
    p = subprocess.run([sys.executable, "-c", VECTOR_CHECK], env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0 and "VECTOR_OK" in p.stdout, p.stdout[-2500:] + p.stderr[-3500:]


def test_reference_env_classes_equal_our_glue_on_the_same_physics():
    p = subprocess.run([sys.executable, "-c", GLUE_CHECK], env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0 and "GLUE_OK" in p.stdout, p.stdout[-2500:] + p.stderr[-3500:]
    assert p.stdout.count(" ok:") == 11


KWARGS_CHECK = r'''
TESTS_DIR = %r
import numpy as np
import mujoco, gymnasium as gym
from gymnasium.utils.env_checker import data_equivalence
import gymnasium_amd
from oracle import oracle
assert getattr(mujoco, "IS_ORACLE_SHIM", False)
import sys
sys.path.insert(0, TESTS_DIR)
from mujoco_kwargs_cases import CASES
count = 0
for env_id, cases in CASES.items():
    for kw in cases:
        n = 3
        ref = gym.make_vec(env_id, num_envs=n, vectorization_mode="sync", max_episode_steps=20, **kw)
        ours = gym.make_vec("MI355X/" + env_id, num_envs=n, max_episode_steps=20, _engine_factory=oracle.engine_factory, **kw)
        assert ours.single_observation_space == ref.single_observation_space, (env_id, kw, ours.single_observation_space, ref.single_observation_space)
        assert data_equivalence(ours.reset(seed=8), ref.reset(seed=8), exact=True), (env_id, kw, "reset")
        ref.action_space.seed(4)
        for t in range(45):
            a = ref.action_space.sample()
            s1, s2 = ours.step(a), ref.step(a)
            for k, what in enumerate(("obs", "reward", "terminated", "truncated")):
                assert data_equivalence(s1[k], s2[k], exact=True), (env_id, kw, t, what, s1[k], s2[k])
            assert data_equivalence(dict(s1[4]), dict(s2[4]), exact=True), (env_id, kw, t, "infos")
        ours.close(), ref.close()
        count += 1
print("KWARGS_OK", count)
'''


def test_non_default_constructor_kwargs_equal_the_reference_env_classes():
    """Every documented constructor keyword of the eleven v5 classes at a non-default value (reward weights, healthy / contact / impact ranges, observation
    switches, reset-noise scales, frame_skip), through make_vec on both sides: the reference's env classes over the stand-in physics vs the engine's glue,
    strict data_equivalence of observations (their SHAPE changes with the switches), rewards, flags and infos."""
    p = subprocess.run([sys.executable, "-c", KWARGS_CHECK % os.path.join(ROOT, "tests")], env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0 and "KWARGS_OK 15" in p.stdout, p.stdout[-2500:] + p.stderr[-3500:]


DOCTEST_CHECK = r'''
import doctest
import mujoco, gymnasium as gym
import gymnasium.wrappers.transform_action as ta
import gymnasium.wrappers.vector.dict_info_to_list as di
assert getattr(mujoco, "IS_ORACLE_SHIM", False)
for mod, name in ((ta, "DiscretizeAction"), (di, "DictInfoToList")):
    obj = getattr(mod, name)
    finder, runner = doctest.DocTestFinder(recurse=False), doctest.DocTestRunner(optionflags=doctest.ELLIPSIS | doctest.NORMALIZE_WHITESPACE)
    globs = {"gym": gym, name: obj}
    tests = [t for t in finder.find(obj, name, globs=globs) if t.examples]
    assert len(tests) == 1, (name, len(tests))
    res = runner.run(tests[0])
    assert res.failed == 0 and res.attempted >= 15, (name, res)
    print(name, "doctest ok:", res.attempted, "examples")
print("DOCTEST_OK")
'''


def test_the_reference_doctests_that_print_mujoco_numbers_pass_on_the_oracle():
    """The two docstrings of the reference that print numbers a real `mujoco` produced -- DiscretizeAction (Reacher-v5: three 10-value
    observations, gymnasium/wrappers/transform_action.py:223-257) and DictInfoToList (HalfCheetah-v5 infos of a 2-env SyncVectorEnv,
    gymnasium/wrappers/vector/dict_info_to_list.py:49-62) -- run by `doctest` itself, with the reference's own env classes and wrappers on top
    of the oracle's physics: every printed digit matches.  (tests/test_mujoco_reference_pins.py holds the same numbers as constants for
    the boxes without the reference tree, and checks the HIP engine against them.)"""
    p = subprocess.run([sys.executable, "-c", DOCTEST_CHECK], env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "DOCTEST_OK" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]
