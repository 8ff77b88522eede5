"""CPU validation of the cooperative MuJoCo kernel (gymnasium_amd/csrc/mjx_coop.h).

The device source is compiled for the host (tests/coop_emu/emu.cpp: one fiber per lane, coop_sync() = round-robin yield)
and compared with the C oracle (oracle/mujoco_core.c) on the same states.  HIP-vs-oracle parity on the GPU is in
tests/test_gpu_mujoco.py; this file pins the ALGORITHM (level-parallel tree passes, row-per-lane Cholesky, per-contact
3x3 Hessian blocks, butterfly line search) without a GPU.  Physics parity with `mujoco` itself stays UNPINNED.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import mujoco as om

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_DIR = os.path.join(HERE, "coop_emu")
NAMES = ["half_cheetah", "ant", "humanoid"]
# emulator model id -> (robot, solver): the humanoid runs its MJCF's PGS / 50 (id 2) or the opt-in Newton solver (id 12)
VARIANTS = {0: ("half_cheetah", None), 1: ("ant", None), 3: ("hopper", None), 4: ("walker2d", None), 2: ("humanoid", "PGS"), 12: ("humanoid", "Newton"),
            8: ("humanoid_standup", "PGS"), 18: ("humanoid_standup", "Newton")}
_LIB = None


def oracle_model(model):
    from gymnasium_amd.envs.mujoco import compiler as cp

    name, solver = VARIANTS[model]
    return om.OracleModel(cp.compile_model(name, faithful_solver=(solver != "Newton")))


# (until round 6 the header kept the predecessors of the round-3 rewrites and the rejected PGS sweeps behind A/B switches, and this file compared builds
#  with them on and off; the switches and their dead branches are gone -- docs/rejected_experiments.md has the measurements, the git history the code)
_LIBS = {}


def build_emu(name, flags=()):
    so, src = os.path.join(EMU_DIR, name), os.path.join(EMU_DIR, "emu.cpp")
    csrc = os.path.join(HERE, "..", "gymnasium_amd", "csrc")
    deps = [src, __file__] + [os.path.join(csrc, f) for f in ("mjx_coop.h", "mjx_core.h", os.path.join("generated", "mjx_models.h"))]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        from gymnasium_amd.envs.mujoco import codegen

        codegen.generate()
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fvisibility=hidden", "-Wno-unknown-pragmas", *flags,
                        "-o", so, src], check=True, cwd=EMU_DIR)
    L = C.CDLL(so)
    L.coop_emu_step.restype = C.c_int
    L.coop_emu_board_bytes.restype = C.c_long
    return L


def lib():
    if "product" not in _LIBS:
        extra = os.environ.get("MJX_EMU_EXTRA_FLAGS", "").split()  # the same tests over an A/B build of the kernels (scripts/build_variant.py's host-side twin)
        _LIBS["product"] = build_emu("libcoop_emu_variant.so" if extra else "libcoop_emu.so", extra)
    return _LIBS["product"]


def emu(model, m, qpos, qvel, ctrl, nsub, warm=None):
    L = lib()
    qo, vo = np.zeros(m.nq), np.zeros(m.nv)
    ex, dbg = np.zeros(L.coop_emu_extras_dim(model % 10)), np.zeros(4 * m.nv + m.nv * m.nv)
    p = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(C.c_void_p)
    ncon = L.coop_emu_step(model, p(qpos), p(qvel), p(ctrl), nsub, p(qo), p(vo), p(ex), p(dbg), None if warm is None else warm.ctypes.data_as(C.c_void_p))
    return qo, vo, ex, dbg, ncon


@pytest.mark.parametrize("model", list(VARIANTS), ids=[f"{n}-{s}" if s else n for n, s in VARIANTS.values()])
def test_forward_matches_oracle(model):
    om_ = oracle_model(model)
    m, d = om_.m, om_.make_data()
    rng = np.random.default_rng(model)
    nv = m.nv
    for trial in range(6):
        qpos = m.qpos0 + rng.uniform(-0.3, 0.3, m.nq)
        qvel = rng.normal(size=nv)
        if model > 0:
            qpos[2] = m.qpos0[2] - (0.25 if trial % 2 else 0.0)  # push into the floor every other trial
            qpos[3:7] /= np.linalg.norm(qpos[3:7])
        ctrl = rng.uniform(-1, 1, m.nu)
        d.reset(), d.set_state(qpos, qvel, ctrl), d.forward()
        _, _, _, dbg, ncon = emu(model, m, qpos, qvel, ctrl, 0)
        assert ncon == d.get("ncon")
        scale = max(1.0, np.abs(d.get("qacc")).max())
        pgs = VARIANTS[model][1] == "PGS"
        if not pgs:  # the PGS path reuses the blackboard storage of M for M^-1 once M is factorised
            np.testing.assert_allclose(dbg[4 * nv:].reshape(nv, nv), d.get("qM"), rtol=0, atol=1e-13)
        np.testing.assert_allclose(dbg[2 * nv:3 * nv], d.get("qfrc_bias"), rtol=0, atol=1e-11)
        if d.get("nefc") == 0 or pgs:  # Newton with constraint rows starts from the warm start and never forms qacc_smooth
            np.testing.assert_allclose(dbg[nv:2 * nv], d.get("qacc_smooth"), rtol=0, atol=1e-12 * scale)
        np.testing.assert_allclose(dbg[:nv], d.get("qacc"), rtol=0, atol=1e-11 * scale)
        if not pgs:
            np.testing.assert_allclose(dbg[3 * nv:4 * nv], d.get("qfrc_constraint"), rtol=0, atol=1e-9 * scale)


@pytest.mark.parametrize("model", list(VARIANTS), ids=[f"{n}-{s}" if s else n for n, s in VARIANTS.values()])
def test_env_steps_match_oracle_with_contacts(model):
    """frame_skip sub-steps (Euler with implicit damping / RK4) from states the oracle reached under a random policy."""
    om_ = oracle_model(model)
    m, d = om_.m, om_.make_data()
    nb, amp = m.nbody, (0.4 if model >= 2 else 1.0)
    seen_contacts = most_contacts = 0
    for trial in range(2):
        rng = np.random.default_rng(100 + trial)
        qpos = m.qpos0 + rng.uniform(-0.1, 0.1, m.nq)
        if model > 0:
            qpos[3:7] /= np.linalg.norm(qpos[3:7])
        d.reset(), d.set_state(qpos, 0.1 * rng.normal(size=m.nv), np.zeros(m.nu))
        for _ in range(120 if model in (2, 12) else (20 if model in (8, 18) else 60)):
            d.set_state(None, None, amp * rng.uniform(-1, 1, m.nu)), d.step(5)
        q, v = d.get("qpos"), d.get("qvel")
        warm = d.get("qacc_warmstart").copy()  # carried across env steps like the kernel's qacc_warmstart slot (PGS results depend on it)
        for _ in range(4):
            ctrl = amp * rng.uniform(-1, 1, m.nu)
            d.set_state(q, v, ctrl), d.step(5), d.rne_post_constraint()
            qo, vo, ex, _, ncon = emu(model, m, q, v, ctrl, 5, warm)
            assert ncon == d.get("ncon")
            seen_contacts += ncon
            most_contacts = max(most_contacts, ncon)
            np.testing.assert_allclose(qo, d.get("qpos"), rtol=0, atol=1e-11)
            np.testing.assert_allclose(vo, d.get("qvel"), rtol=0, atol=1e-10)
            cf = ex[4:4 + 6 * nb].reshape(nb, 6)
            np.testing.assert_allclose(cf, d.get("cfrc_ext"), rtol=0, atol=1e-9 * max(1.0, np.abs(cf).max()))
            np.testing.assert_allclose(ex[0:2], d.get("xpos")[1][:2], rtol=0, atol=1e-13)
            np.testing.assert_allclose(ex[4 + 6 * nb:4 + 16 * nb].reshape(nb, 10), d.get("cinert"), rtol=0, atol=1e-12)
            np.testing.assert_allclose(ex[4 + 16 * nb:4 + 22 * nb].reshape(nb, 6), d.get("cvel"), rtol=0, atol=1e-10)
            q, v = d.get("qpos"), d.get("qvel")
    assert seen_contacts > 0 and most_contacts > 0


def test_pgs_with_ten_and_more_contacts():
    """The cooperative PGS keeps M^-1 J_c^T of the first 15 contacts on the blackboard (the storage of M, of the Cholesky factor, of the RNE pass and two
    blocks of their own) and applies M^-1 directly for the rest: a humanoid pressed flat into the floor (>= 10 contacts) must equal the oracle, forward
    pass and sub-steps."""
    om_ = oracle_model(8)
    m, d = om_.m, om_.make_data()
    rng = np.random.default_rng(5)
    for trial in range(3):
        qpos = m.qpos0 + rng.uniform(-0.05, 0.05, m.nq)
        qpos[3:7] = m.qpos0[3:7]
        qpos[2] = m.qpos0[2] - 0.03 - 0.02 * trial  # lying on its back, pushed into the plane
        qvel, ctrl = 0.3 * rng.normal(size=m.nv), 0.4 * rng.uniform(-1, 1, m.nu)
        d.reset(), d.set_state(qpos, qvel, ctrl), d.forward()
        assert d.get("ncon") >= 10, d.get("ncon")
        _, _, _, dbg, ncon = emu(8, m, qpos, qvel, ctrl, 0)
        assert ncon == d.get("ncon")
        np.testing.assert_allclose(dbg[:m.nv], d.get("qacc"), rtol=0, atol=1e-10 * max(1.0, np.abs(d.get("qacc")).max()))
        d.reset(), d.set_state(qpos, qvel, ctrl), d.step(2), d.rne_post_constraint()
        qo, vo, ex, _, _ = emu(8, m, qpos, qvel, ctrl, 2, np.zeros(m.nv))
        np.testing.assert_allclose(qo, d.get("qpos"), rtol=0, atol=1e-10)
        np.testing.assert_allclose(vo, d.get("qvel"), rtol=0, atol=1e-8 * max(1.0, np.abs(vo).max()))
        cf = ex[4:4 + 6 * m.nbody].reshape(m.nbody, 6)
        np.testing.assert_allclose(cf, d.get("cfrc_ext"), rtol=0, atol=1e-8 * max(1.0, np.abs(cf).max()))


# ---- the ONE-LANE simulator (mjx_core.h, the product kernel of the nine smaller robots) compiled for the host ------------------------
CORE_NAMES = ["half_cheetah", "ant", "humanoid", "hopper", "walker2d", "inverted_pendulum", "inverted_double_pendulum", "reacher",
              "humanoid_standup", "swimmer", "pusher"]  # ids as in oracle/mujoco_envs.h


def core(model, m, qpos, qvel, ctrl, nsub, warm=None):
    qo, vo, ao = np.zeros(m.nq), np.zeros(m.nv), np.zeros(m.nv)
    p = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(C.c_void_p)
    fn = lib().core_emu_step
    fn.restype = C.c_int
    ncon = fn(model, p(qpos), p(qvel), p(ctrl), nsub, p(qo), p(vo), p(ao), None if warm is None else warm.ctypes.data_as(C.c_void_p), None)
    return qo, vo, ao, ncon


@pytest.mark.parametrize("model", range(len(CORE_NAMES)), ids=CORE_NAMES)
def test_one_lane_simulator_matches_oracle(model):
    """States a random policy reaches on the oracle, then env steps (frame_skip sub-steps, warm start carried) on both implementations."""
    name = CORE_NAMES[model]
    om_ = om.OracleModel(name)
    m, d = om_.m, om_.make_data()
    free_root = name in ("ant", "humanoid", "humanoid_standup")
    amp = 0.4 if name.startswith("humanoid") else (2.0 if name == "pusher" else 1.0)
    nsub = {"inverted_pendulum": 2, "reacher": 2, "swimmer": 4, "hopper": 4, "walker2d": 4}.get(name, 5)
    seen = 0
    for trial in range(2):
        rng = np.random.default_rng(10 * model + trial)
        qpos = m.qpos0 + rng.uniform(-0.1, 0.1, m.nq)
        if free_root:
            qpos[3:7] /= np.linalg.norm(qpos[3:7])
        if name == "pusher":  # lower the arm towards the table and put the object in front of the wrist (tests/test_mujoco_oracle.py)
            qpos[:] = 0
            qpos[1], qpos[3], qpos[6] = rng.uniform(0.45, 0.62), rng.uniform(-0.45, -0.05), rng.uniform(-0.5, 0.5)
            qpos[7], qpos[8] = -0.6 + rng.uniform(-0.12, 0.12) + 0.05, 0.80 + rng.uniform(-0.08, 0.06) - 0.45
        d.reset(), d.set_state(qpos, 0.1 * rng.normal(size=m.nv), np.zeros(m.nu))
        for _ in range(20 if name in ("pusher", "inverted_pendulum", "inverted_double_pendulum") else 60):
            d.set_state(None, None, amp * rng.uniform(-1, 1, m.nu)), d.step(nsub)
        q, v = d.get("qpos"), d.get("qvel")
        warm = d.get("qacc_warmstart").copy()
        for _ in range(4):
            ctrl = amp * rng.uniform(-1, 1, m.nu)
            d.set_state(q, v, ctrl), d.step(nsub)
            qo, vo, _, ncon = core(model, m, q, v, ctrl, nsub, warm)
            assert ncon == d.get("ncon")
            seen += ncon + d.get("nefc")
            np.testing.assert_allclose(qo, d.get("qpos"), rtol=0, atol=1e-10 * max(1.0, np.abs(qo).max()))
            np.testing.assert_allclose(vo, d.get("qvel"), rtol=0, atol=1e-9 * max(1.0, np.abs(vo).max()))
            q, v = d.get("qpos"), d.get("qvel")
    if name not in ("swimmer", "reacher", "inverted_pendulum", "inverted_double_pendulum"):
        assert seen > 0, "these robots must have met their constraints (contacts / joint limits)"


# ---- race detector: the result must not depend on the order in which the lanes of a group run between two coop_sync() calls -----------------
@pytest.mark.parametrize("model", list(VARIANTS), ids=[f"{n}-{s}" if s else n for n, s in VARIANTS.values()])
def test_no_cross_lane_dependency_inside_a_sync_interval(model):
    """The cooperative kernel's synchronisation discipline (mjx_coop.h coop_sync): between two fences no blackboard word is written by one
    lane and read or written by another.  On the GPU the lanes of a group run in lockstep, so a violation would read old or new data
    depending on how the COMPILER interleaved the two accesses -- results that change with the instruction scheduler (build.py keeps three
    such stories about the 16-lane unit).  The emulator runs the lanes of an interval one after the other; a violation makes the result
    depend on that order.  So: ascending, descending and per-interval random lane orders must give bit-identical states, extras and
    warm starts, over env steps with active contacts and joint limits."""
    om_ = oracle_model(model)
    m, d = om_.m, om_.make_data()
    amp = 0.4 if model >= 2 else 1.0
    rng = np.random.default_rng(7 + model)
    qpos = m.qpos0 + rng.uniform(-0.1, 0.1, m.nq)
    if model > 0:
        qpos[3:7] /= np.linalg.norm(qpos[3:7])
    d.reset(), d.set_state(qpos, 0.1 * rng.normal(size=m.nv), np.zeros(m.nu))
    states = []
    for t in range(90 if model in (2, 12) else (24 if model in (8, 18) else 60)):
        d.set_state(None, None, amp * rng.uniform(-1, 1, m.nu)), d.step(5)
        if t % 24 == 23:
            states.append((d.get("qpos").copy(), d.get("qvel").copy(), d.get("qacc_warmstart").copy()))
            if model > 0:  # ... and the same state pushed into the floor: many active contact rows
                q2 = states[-1][0].copy()
                q2[2] -= 0.2
                states.append((q2, states[-1][1], states[-1][2]))
    set_order = lib().coop_emu_set_order
    set_order.argtypes = [C.c_int, C.c_ulonglong]
    contacts = 0
    try:
        for q, v, w in states:
            ctrl = amp * rng.uniform(-1, 1, m.nu)
            runs = []
            for mode, seed in ((0, 1), (1, 1), (2, 12345 + 7 * len(runs))):
                set_order(mode, seed)
                warm = w.copy()
                qo, vo, ex, dbg, ncon = emu(model, m, q, v, ctrl, 3, warm)
                runs.append((qo, vo, ex, warm, ncon))
            contacts += runs[0][4]
            for r in runs[1:]:
                assert r[4] == runs[0][4]
                for a, b, what in zip(r[:4], runs[0][:4], ("qpos", "qvel", "extras", "warm start")):
                    assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), f"{what} depends on the lane order inside a sync interval: a blackboard race"
    finally:
        set_order(0, 1)
    assert contacts > 0
