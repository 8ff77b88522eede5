"""TEST INFRASTRUCTURE -- a stand-in for the `mujoco` Python package, backed by THIS repo's CPU oracle (oracle/mujoco_core.c).

Why it exists: the MuJoCo half of the path is "parity unpinned" because no `mujoco` wheel is obtainable in the build container
(DESIGN.md section 7).  The pin is prepared -- tests/golden/make_mujoco_golden.py writes fixtures from a real `mujoco`, and
tests/test_mujoco_fixtures.py consumes them -- but neither had ever executed.  With this directory in front of sys.path the REFERENCE's
own env classes (gymnasium/envs/mujoco/*_v5.py, mujoco_env.py: MjModel.from_xml_path, MjData, mj_forward, mj_step, mj_resetData,
mj_rnePostConstraint, data.body(...).xpos, data.site_xpos, ...) run on the oracle, so the generator and all consumer cases run end to end
(tests/test_mujoco_fixture_pipeline.py) -- a SELF-CONSISTENCY check of the pipeline's plumbing (field names, shapes, indexing, the
reference glue driving our physics), NOT a pin: fixtures made this way say nothing about the real MuJoCo and are never committed.
"""
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from gymnasium_amd.envs.mujoco import compiler as _cp  # noqa: E402
from oracle import mujoco as _om  # noqa: E402

__version__ = "0.0.0+oracle.shim"
IS_ORACLE_SHIM = True

_XML = {"ant.xml": "ant", "half_cheetah.xml": "half_cheetah", "hopper.xml": "hopper", "humanoid.xml": "humanoid",
        "humanoidstandup.xml": "humanoid_standup", "inverted_double_pendulum.xml": "inverted_double_pendulum",
        "inverted_pendulum.xml": "inverted_pendulum", "pusher_v5.xml": "pusher", "reacher.xml": "reacher", "swimmer.xml": "swimmer",
        "walker2d_v5.xml": "walker2d"}


class mjtObj:
    mjOBJ_BODY, mjOBJ_CAMERA = 1, 7


def mj_name2id(model, type, name):  # no cameras in the shim: -1 = "not found" (mujoco_rendering.py:740-752 falls back to the free camera)
    if type == mjtObj.mjOBJ_BODY and name in model._m.body_names:
        return model._m.body_names.index(name)
    return -1


class _Namespace:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class MjModel:
    @classmethod
    def from_xml_path(cls, path):
        name = _XML.get(os.path.basename(path))
        if name is None:
            raise ValueError(f"the oracle shim knows the eleven v5 assets of the reference only, not {path}")
        return cls(name)

    def __init__(self, name):
        m = self._m = _cp.compile_model(name, faithful_solver=True)
        self._oracle = _om.OracleModel(m)
        self.nq, self.nv, self.nu, self.na, self.nbody, self.ngeom = m.nq, m.nv, m.nu, 0, m.nbody, m.ngeom
        self.opt = _Namespace(timestep=m.timestep, gravity=np.asarray(m.gravity, dtype=np.float64), integrator=1 if m.integrator == "RK4" else 0,
                              solver={"PGS": 0, "CG": 1, "Newton": 2}[m.reference_solver], iterations=m.iterations, tolerance=1e-8, cone=0,
                              density=m.density, viscosity=m.viscosity)
        self.stat = _Namespace(meaninertia=m.meaninertia)
        self.vis = _Namespace(global_=_Namespace(offwidth=640, offheight=480))
        for f in ("body_parentid", "body_mass", "body_ipos", "body_pos", "body_quat", "body_invweight0", "jnt_type", "jnt_qposadr",
                  "jnt_dofadr", "jnt_bodyid", "jnt_limited", "jnt_pos", "jnt_axis", "jnt_range", "jnt_stiffness", "jnt_margin", "jnt_solref",
                  "jnt_solimp", "dof_bodyid", "dof_jntid", "dof_parentid", "dof_armature", "dof_damping", "dof_invweight0", "qpos0",
                  "qpos_spring", "geom_type", "geom_bodyid", "geom_size", "geom_pos", "actuator_ctrlrange"):
            if hasattr(m, f):
                setattr(self, f, np.array(getattr(m, f)))
        # mjModel stores the principal moments (+ body_iquat); the compiled model keeps the full tensor in body axes
        self.body_inertia = np.array([np.sort(np.linalg.eigvalsh(np.asarray(t, dtype=np.float64).reshape(3, 3)))[::-1] for t in m.body_inertia])
        self.actuator_gear = np.concatenate([np.asarray(m.actuator_gear, dtype=np.float64)[:, None], np.zeros((m.nu, 5))], axis=1)


class _Body:
    def __init__(self, data, idx):
        self._d, self._i = data, idx

    @property
    def xpos(self):
        return self._d.xpos[self._i]


class _Contact:
    def __init__(self, row, mu, margin):
        self.dist, self.pos, self.frame = float(row[0]), row[1:4].copy(), row[4:13].copy()
        self.geom = np.array([int(row[13]), int(row[14])])
        self.dim, self.efc_address = int(row[15]), int(row[16])
        self.friction, self.includemargin = np.array([mu, mu, 0.005, 0.0001, 0.0001]), margin


_PULL = ("qpos", "qvel", "xpos", "xquat", "xmat", "xipos", "subtree_com", "cinert", "cdof", "cvel", "qfrc_bias", "qfrc_passive", "qfrc_actuator",
         "qfrc_smooth", "qacc_smooth", "qfrc_constraint", "qacc", "qacc_warmstart", "cfrc_ext", "ten_length", "ten_velocity")


class MjData:
    def __init__(self, model):
        self._model, self._d = model, model._oracle.make_data()
        m = model._m
        self.qpos, self.qvel, self.ctrl, self.act = np.array(m.qpos0, dtype=np.float64), np.zeros(m.nv), np.zeros(m.nu), np.zeros(0)
        self.ncon = self.nefc = 0
        self.contact = []
        self._d.reset()
        self._pull(state=True)

    def body(self, key):
        return _Body(self, self._model._m.body_names.index(key) if isinstance(key, str) else int(key))

    def _push(self):
        self._d.set_state(self.qpos, self.qvel, self.ctrl)

    def _pull(self, state=False):
        m, d = self._model._m, self._d
        for f in _PULL:
            v = d.get(f)
            if f in ("qpos", "qvel"):
                if state:
                    getattr(self, f)[:] = v  # the env holds references to these arrays: update in place
                continue
            if f in ("ten_length", "ten_velocity"):
                v = v[:m.ntendon]
            setattr(self, f, v)
        self.ximat = self.xmat  # (not modelled separately: nothing in the reference path reads it)
        self.qM = d.get("qM")   # dense; mj_fullM copies it
        self.site_xpos = np.array([self.xpos[b] + self.xmat[b].reshape(3, 3) @ np.asarray(p) for b, p in m.sites]).reshape(-1, 3)
        self.ncon, self.nefc = d.get("ncon"), d.get("nefc")
        rows = d.get("contact") if self.ncon else np.zeros((0, 17))
        pair_of = {(int(g1), int(g2)): k for k, (g1, g2) in enumerate(zip(m.pair_geom1, m.pair_geom2))}
        self.contact = []
        for r in rows:
            k = pair_of.get((int(r[13]), int(r[14])), pair_of.get((int(r[14]), int(r[13])), 0))
            self.contact.append(_Contact(r, float(np.ravel(m.pair_friction[k])[0]), float(m.pair_margin[k])))
        if self.nefc:
            self.efc_J = d.get("efc_J")
            for f in ("efc_pos", "efc_margin", "efc_D", "efc_R", "efc_aref", "efc_force"):
                setattr(self, f, d.get(f))
            self.efc_type = np.full(self.nefc, 5)  # (constraint type codes are not compared by the consumer)


def mj_resetData(model, data):
    data._d.reset()
    data.ctrl[:] = 0
    data._pull(state=True)


def mj_forward(model, data):
    data._push()
    data._d.forward()
    data._pull()


def mj_step(model, data, nstep=1):
    data._push()
    data._d.step(int(nstep))
    data._pull(state=True)


def mj_rnePostConstraint(model, data):
    data._d.rne_post_constraint()
    data.cfrc_ext = data._d.get("cfrc_ext")


def mj_fullM(model, dst, qM):
    dst[...] = np.asarray(qM).reshape(dst.shape)
