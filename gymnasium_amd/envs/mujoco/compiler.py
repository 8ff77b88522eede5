"""Model compiler: MJCF-equivalent description (models.py) -> flat arrays for the engine.

This restates what MuJoCo's compiler does for the features the three reference robots use
(gymnasium/envs/mujoco/assets/*.xml) -- the `mujoco` library is a third-party dependency that is NOT in the reference
tree (pyproject.toml:45, `mujoco >= 2.1.5`), so the algorithm follows MuJoCo's published modelling documentation
(XML reference: defaults, `inertiafromgeom`, `settotalmass`, `fromto`, contact parameter mixing; computation chapter:
`invweight0`) and is **parity-unpinned** until a `mujoco` build can produce fixtures:

  * defaults (joint / geom / option) and per-element overrides; `angle="degree"` conversion of ranges,
  * geoms: `fromto` capsules (centre, half-length, orientation taking +z to the segment), `axisangle`, capsule and sphere
    volumes / inertias at the geom density (default 1000), body mass / centre of mass / inertia tensor from its geoms
    (`inertiafromgeom`), global rescale by `settotalmass`,
  * kinematic tree arrays (MuJoCo's depth-first body order; joints, dofs, qpos addresses; free joint = 7 qpos / 6 dofs),
  * contact candidates: contype/conaffinity filter, same-body and parent-child exclusion (the world body is exempt),
    per-pair condim = max, friction = element-wise max, margin = max, solref / solimp mixed with equal solmix weights,
  * `dof_invweight0` / `body_invweight0` at qpos0 (diagonal of M^-1 and of J M^-1 J^T, averaged per joint / per body).

The numpy kinematics / Jacobian / mass-matrix code here is deliberately a different formulation from the engine's
composite-rigid-body recursion (M = sum_b m Jv'Jv + Jw' I Jw), so tests can cross-check the two.
"""
from __future__ import annotations

import math

import numpy as np

from . import models as _models

FREE, BALL, SLIDE, HINGE = 0, 1, 2, 3          # mjtJoint
PLANE, SPHERE, CAPSULE, CYLINDER = 0, 2, 3, 5  # mjtGeom values
MINVAL = 1e-15

GEOM_DEFAULTS = dict(contype=1, conaffinity=1, condim=3, density=1000.0, friction=(1.0, 0.005, 0.0001), margin=0.0, gap=0.0,
                     solref=(0.02, 1.0), solimp=(0.9, 0.95, 0.001, 0.5, 2.0), solmix=1.0)
JOINT_DEFAULTS = dict(armature=0.0, damping=0.0, stiffness=0.0, limited=False, margin=0.0, solreflimit=(0.02, 1.0),
                      solimplimit=(0.9, 0.95, 0.001, 0.5, 2.0), springref=0.0, ref=0.0)


# ---- small quaternion / rotation helpers (MuJoCo convention: w, x, y, z) -----------------------------------------
def quat_mul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                     a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
                     a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def axisangle_to_quat(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    n = np.linalg.norm(axis)
    if n < MINVAL:
        return np.array([1.0, 0, 0, 0])
    s = math.sin(angle * 0.5)
    return np.concatenate([[math.cos(angle * 0.5)], axis / n * s])


def z_to_quat(vec):
    """Quaternion rotating +z onto `vec` (MuJoCo mjuu_z2quat)."""
    vec = np.asarray(vec, dtype=np.float64)
    vec = vec / np.linalg.norm(vec)
    z = np.array([0.0, 0, 1])
    axis = np.cross(z, vec)
    s = np.linalg.norm(axis)
    if s < 1e-10:
        return np.array([1.0, 0, 0, 0]) if vec[2] > 0 else np.array([0.0, 1, 0, 0])
    ang = math.atan2(s, vec[2])
    axis = axis / s
    return np.concatenate([[math.cos(ang / 2)], axis * math.sin(ang / 2)])


def _complete_solimp(v):
    full = list(GEOM_DEFAULTS["solimp"])
    for k, x in enumerate(v):
        full[k] = float(x)
    return tuple(full)


class CompiledModel:
    """Flat arrays of one robot; attribute names follow mjModel where a counterpart exists."""

    def __repr__(self):
        return f"<CompiledModel {self.name} nq={self.nq} nv={self.nv} nu={self.nu} nbody={self.nbody} ngeom={self.ngeom} npair={len(self.pair_geom1)}>"


def compile_model(desc, faithful_solver: bool = True) -> CompiledModel:
    """``faithful_solver=True`` (default): the constraint solver is the one the MJCF asks for -- ``solver="PGS" iterations="50"`` for
    the two humanoids (humanoid.xml:8), MuJoCo's default Newton for everything else.  ``faithful_solver=False`` solves every model
    to convergence with the primal Newton method (the opt-in ``solver="Newton"`` of the humanoid envs): both minimise the same convex
    cost, PGS truncated at 50 sweeps leaves a ~1e-5 relative residual in qacc that belongs to that solver run, not to the model."""
    if isinstance(desc, str):
        desc = _models.MODELS[desc]()
    m = CompiledModel()
    m.name = desc["name"]
    deg = desc["angle"] == "degree"
    opt = desc["option"]
    m.timestep = float(opt["timestep"])
    m.gravity = np.array(opt["gravity"], dtype=np.float64)
    m.integrator = opt["integrator"]
    m.reference_solver = opt["solver"]
    m.solver = opt["solver"] if faithful_solver else "Newton"
    m.iterations = int(opt["iterations"])
    jdef = dict(JOINT_DEFAULTS, **desc.get("joint_default", {}))
    gdef = dict(GEOM_DEFAULTS, **desc.get("geom_default", {}))

    # ---- flatten the tree in MuJoCo's order (depth first, document order); body 0 is the world ------------------
    bodies = [dict(name="world", pos=(0, 0, 0), quat=None, joints=[], geoms=[], parent=-1)]

    def visit(b, parent):
        idx = len(bodies)
        bodies.append(dict(b, parent=parent))
        for c in b["children"]:
            visit(c, idx)

    for b in desc["bodies"]:
        visit(b, 0)
    nbody = len(bodies)
    m.nbody = nbody
    m.body_names = [b["name"] for b in bodies]
    m.body_parentid = np.array([max(b["parent"], 0) for b in bodies], dtype=np.int32)
    m.body_pos = np.array([b["pos"] for b in bodies], dtype=np.float64)
    m.body_quat = np.array([(b["quat"] if b["quat"] is not None else (1, 0, 0, 0)) for b in bodies], dtype=np.float64)
    m.body_quat /= np.linalg.norm(m.body_quat, axis=1, keepdims=True)
    m.body_rootid = np.zeros(nbody, dtype=np.int32)
    for i in range(1, nbody):
        p = m.body_parentid[i]
        m.body_rootid[i] = i if p == 0 else m.body_rootid[p]

    # ---- geoms: world plane first (geom 0), then the bodies' geoms in order --------------------------------------
    geoms = []
    # a model without a ground plane still gets geom 0 (the engine's tables are never empty), with collisions switched off
    fl = dict(gdef, **(desc["floor"] if desc["floor"] is not None else dict(contype=0, conaffinity=0)))
    geoms.append(dict(name="floor", type=PLANE, body=0, size=(0, 0, 0), pos=np.array(fl.get("pos", (0, 0, 0)), dtype=np.float64), quat=np.array([1.0, 0, 0, 0]), mass=0.0,
                      inertia=np.zeros(3), **{k: fl[k] for k in ("contype", "conaffinity", "condim", "friction", "margin", "gap", "solref", "solimp", "solmix")}))
    for bi in range(1, nbody):
        for g in bodies[bi]["geoms"]:
            p = dict(gdef)
            p.update({k: v for k, v in g.items() if k in GEOM_DEFAULTS})
            if len(p["friction"]) < 3:  # friction="0.9": only the sliding coefficient is overridden (XML attribute completion)
                p["friction"] = tuple(p["friction"]) + tuple(gdef["friction"][len(p["friction"]):])
            size = g["size"]
            radius = float(size[0] if isinstance(size, (tuple, list)) else size)
            if g["type"] == "capsule":
                if g["fromto"] is not None:
                    a, b = np.array(g["fromto"][:3], dtype=np.float64), np.array(g["fromto"][3:], dtype=np.float64)
                    pos, half = 0.5 * (a + b), 0.5 * np.linalg.norm(b - a)
                    quat = z_to_quat(a - b)  # MuJoCo: vec = from - to
                else:
                    half = float(size[1])
                    pos = np.array(g["pos"] if g["pos"] is not None else (0, 0, 0), dtype=np.float64)
                    aa = g["axisangle"]
                    if g.get("quat") is not None:
                        quat = np.array(g["quat"], dtype=np.float64)
                        quat /= np.linalg.norm(quat)
                    elif aa is not None:
                        quat = axisangle_to_quat(aa[:3], math.radians(aa[3]) if deg else aa[3])
                    else:
                        quat = np.array([1.0, 0, 0, 0])
                height = 2 * half
                vol = math.pi * radius * radius * height + 4.0 / 3.0 * math.pi * radius ** 3
                mass = p["density"] * vol
                sphere_mass = mass * (4.0 / 3.0 * math.pi * radius ** 3) / vol
                cyl_mass = mass - sphere_mass
                ixx = cyl_mass * (3 * radius * radius + height * height) / 12.0
                izz = cyl_mass * radius * radius / 2.0
                sph_i = 2.0 * sphere_mass * radius * radius / 5.0
                ixx += sph_i + sphere_mass * height * (3 * radius + 2 * height) / 8.0
                izz += sph_i
                inertia = np.array([ixx, ixx, izz])
                gtype, gsize = CAPSULE, (radius, half, 0.0)
            elif g["type"] == "sphere":
                pos, quat = np.array(g["pos"], dtype=np.float64), np.array([1.0, 0, 0, 0])
                vol = 4.0 / 3.0 * math.pi * radius ** 3
                mass = p["density"] * vol
                inertia = np.full(3, 2.0 * mass * radius * radius / 5.0)
                gtype, gsize = SPHERE, (radius, 0.0, 0.0)
            elif g["type"] == "cylinder":  # size = (radius, half height), axis = local z; solid cylinder inertia
                half = float(size[1])
                pos, quat = np.array(g["pos"] if g["pos"] is not None else (0, 0, 0), dtype=np.float64), np.array([1.0, 0, 0, 0])
                height = 2 * half
                mass = p["density"] * math.pi * radius * radius * height
                ixx = mass * (3 * radius * radius + height * height) / 12.0
                inertia = np.array([ixx, ixx, mass * radius * radius / 2.0])
                gtype, gsize = CYLINDER, (radius, half, 0.0)
            else:
                raise ValueError(g["type"])
            geoms.append(dict(name=g["name"], type=gtype, body=bi, size=gsize, pos=pos, quat=quat, mass=mass, inertia=inertia,
                              **{k: p[k] for k in ("contype", "conaffinity", "condim", "friction", "margin", "gap", "solref", "solimp", "solmix")}))
    ngeom = len(geoms)
    m.ngeom = ngeom
    m.geom_names = [g["name"] for g in geoms]
    m.geom_type = np.array([g["type"] for g in geoms], dtype=np.int32)
    m.geom_bodyid = np.array([g["body"] for g in geoms], dtype=np.int32)
    m.geom_size = np.array([g["size"] for g in geoms], dtype=np.float64)
    m.geom_pos = np.array([g["pos"] for g in geoms], dtype=np.float64)
    m.geom_quat = np.array([g["quat"] for g in geoms], dtype=np.float64)
    m.geom_mat = np.array([quat_to_mat(g["quat"]) for g in geoms])

    # ---- body inertial properties from geoms (inertiafromgeom) --------------------------------------------------
    m.body_mass = np.zeros(nbody)
    m.body_ipos = np.zeros((nbody, 3))
    m.body_inertia = np.zeros((nbody, 3, 3))   # full tensor about the body's centre of mass, in body-frame axes
    for bi in range(1, nbody):
        gs = [g for g in geoms if g["body"] == bi]
        mass = sum(g["mass"] for g in gs)
        com = sum(g["mass"] * g["pos"] for g in gs) / mass
        I = np.zeros((3, 3))
        for g in gs:
            R = quat_to_mat(g["quat"])
            d = g["pos"] - com
            I += R @ np.diag(g["inertia"]) @ R.T + g["mass"] * (d @ d * np.eye(3) - np.outer(d, d))
        m.body_mass[bi], m.body_ipos[bi], m.body_inertia[bi] = mass, com, I
    if desc.get("settotalmass"):
        scale = desc["settotalmass"] / m.body_mass.sum()
        m.body_mass *= scale
        m.body_inertia *= scale

    # ---- joints / dofs ------------------------------------------------------------------------------------------
    J = dict(type=[], qposadr=[], dofadr=[], bodyid=[], pos=[], axis=[], limited=[], range=[], stiffness=[], margin=[], solref=[], solimp=[], name=[])
    D = dict(bodyid=[], jntid=[], parentid=[], armature=[], damping=[])
    qpos0 = []
    m.body_jntadr, m.body_jntnum = np.full(nbody, -1, dtype=np.int32), np.zeros(nbody, dtype=np.int32)
    m.body_dofadr, m.body_dofnum = np.full(nbody, -1, dtype=np.int32), np.zeros(nbody, dtype=np.int32)
    last_dof_of_body = np.full(nbody, -1, dtype=np.int64)
    for bi in range(1, nbody):
        b = bodies[bi]
        # the last dof of the nearest ancestor that has one
        p, parent_dof = m.body_parentid[bi], -1
        while p > 0 and last_dof_of_body[p] < 0:
            p = m.body_parentid[p]
        if p > 0:
            parent_dof = int(last_dof_of_body[p])
        if b["joints"]:
            m.body_jntadr[bi], m.body_dofadr[bi] = len(J["type"]), len(D["bodyid"])
        for jd in b["joints"]:
            p_ = dict(jdef)
            p_.update({k: v for k, v in jd.items() if k in JOINT_DEFAULTS})
            jt = {"free": FREE, "slide": SLIDE, "hinge": HINGE}[jd["type"]]
            jid = len(J["type"])
            J["name"].append(jd["name"]), J["type"].append(jt), J["qposadr"].append(len(qpos0)), J["dofadr"].append(len(D["bodyid"]))
            J["bodyid"].append(bi), J["pos"].append(jd["pos"])
            axis = np.array(jd["axis"] if jd["axis"] is not None else (0, 0, 1), dtype=np.float64)
            J["axis"].append(axis / np.linalg.norm(axis))
            rng = jd["range"]
            limited = bool(p_["limited"]) and rng is not None and jt != FREE
            if rng is None:
                rng = (0.0, 0.0)
            if deg and jt == HINGE:
                rng = (math.radians(rng[0]), math.radians(rng[1]))
            J["limited"].append(limited), J["range"].append(rng), J["stiffness"].append(float(p_["stiffness"])), J["margin"].append(float(p_["margin"]))
            J["solref"].append(tuple(float(x) for x in p_["solreflimit"])), J["solimp"].append(_complete_solimp(p_["solimplimit"]))
            ndof = 6 if jt == FREE else 1
            for k in range(ndof):
                D["bodyid"].append(bi), D["jntid"].append(jid), D["parentid"].append(parent_dof)
                D["armature"].append(float(p_["armature"])), D["damping"].append(float(p_["damping"]))
                parent_dof = len(D["bodyid"]) - 1
            if jt == FREE:
                qpos0.extend(list(m.body_pos[bi]) + list(m.body_quat[bi]))
            else:  # `ref`: the joint coordinate of the configuration the XML draws (degrees for hinges under angle="degree")
                ref = float(p_["ref"])
                qpos0.append(math.radians(ref) if (deg and jt == HINGE) else ref)
            m.body_jntnum[bi] += 1
            m.body_dofnum[bi] += ndof
        if b["joints"]:
            last_dof_of_body[bi] = len(D["bodyid"]) - 1
    m.njnt, m.nv, m.nq = len(J["type"]), len(D["bodyid"]), len(qpos0)
    m.jnt_names = J["name"]
    m.jnt_type = np.array(J["type"], dtype=np.int32)
    m.jnt_qposadr, m.jnt_dofadr, m.jnt_bodyid = (np.array(J[k], dtype=np.int32) for k in ("qposadr", "dofadr", "bodyid"))
    m.jnt_pos, m.jnt_axis = np.array(J["pos"], dtype=np.float64), np.array(J["axis"], dtype=np.float64)
    m.jnt_limited = np.array(J["limited"], dtype=np.int32)
    m.jnt_range = np.array(J["range"], dtype=np.float64)
    m.jnt_stiffness, m.jnt_margin = np.array(J["stiffness"]), np.array(J["margin"])
    m.jnt_solref, m.jnt_solimp = np.array(J["solref"]), np.array(J["solimp"])
    m.dof_bodyid, m.dof_jntid, m.dof_parentid = (np.array(D[k], dtype=np.int32) for k in ("bodyid", "jntid", "parentid"))
    m.dof_armature, m.dof_damping = np.array(D["armature"]), np.array(D["damping"])
    m.qpos0 = np.array(qpos0)
    m.qpos_spring = m.qpos0.copy()

    # ---- fluid (option density / viscosity): equivalent inertia box of every body, in its principal frame ----------------
    # MuJoCo's inertia-box fluid model (mj_passive): box[k] = sqrt(6 (I_i + I_j - I_k) / mass) from the principal moments;
    # body_imat = principal axes as columns, expressed in the body frame (ximat = xmat @ body_imat)
    m.density, m.viscosity = float(opt.get("density", 0.0)), float(opt.get("viscosity", 0.0))
    m.body_fluidbox, m.body_imat = np.zeros((nbody, 3)), np.tile(np.eye(3).reshape(-1), (nbody, 1))
    for bi in range(1, nbody):
        if m.body_mass[bi] <= 0:
            continue
        I = np.asarray(m.body_inertia[bi], dtype=np.float64).reshape(3, 3)
        if np.abs(I - np.diag(np.diag(I))).max() <= 1e-12 * np.trace(I):
            w, V = np.diag(I).copy(), np.eye(3)   # already principal: keep the body axes (a degenerate pair would let eigh pick any rotation)
        else:
            w, V = np.linalg.eigh(I)
            if np.linalg.det(V) < 0:
                V[:, 2] = -V[:, 2]
        I0, I1, I2 = w
        mass = m.body_mass[bi]
        m.body_fluidbox[bi] = [math.sqrt(max(MINVAL, I1 + I2 - I0) / mass * 6.0), math.sqrt(max(MINVAL, I0 + I2 - I1) / mass * 6.0),
                               math.sqrt(max(MINVAL, I0 + I1 - I2) / mass * 6.0)]
        m.body_imat[bi] = V.reshape(-1)
    # ---- fixed tendons: length = sum_k coef_k qpos[joint_k], velocity = sum_k coef_k qvel[joint_k] (mj_tendon / mj_fwdVelocity for
    # mjWRAP_JOINT wraps); humanoid.xml:91-100.  No stiffness / limits / actuators hang on them in any asset: info values only.
    m.tendon_names = [t[0] for t in desc.get("tendons", [])]
    m.ntendon = len(m.tendon_names)
    m.tendon_wraps = [[(int(m.jnt_qposadr[J["name"].index(j)]), int(m.jnt_dofadr[J["name"].index(j)]), float(c)) for j, c in t[1]]
                      for t in desc.get("tendons", [])]
    # ---- sites: (body index, position in the body frame) ------------------------------------------------------------
    m.sites = [(m.body_names.index(b), tuple(float(x) for x in pos)) for _, b, pos in desc.get("sites", [])]
    # ---- actuators (motors on joints) ---------------------------------------------------------------------------
    m.nu = len(desc["actuators"])
    m.actuator_dofadr = np.array([m.jnt_dofadr[J["name"].index(a[0])] for a in desc["actuators"]], dtype=np.int32)
    m.actuator_gear = np.array([a[1] for a in desc["actuators"]], dtype=np.float64)
    # (joint, gear) uses the model-wide ctrlrange, (joint, gear, (lo, hi)) its own
    m.actuator_ctrlrange = np.array([a[2] if len(a) > 2 else desc["ctrlrange"] for a in desc["actuators"]], dtype=np.float64).reshape(m.nu, 2)

    # ---- contact candidates -------------------------------------------------------------------------------------
    P = dict(g1=[], g2=[], condim=[], friction=[], margin=[], solref=[], solimp=[])
    excluded = set(tuple(e) for e in desc.get("exclude_pairs", ()))  # geom-name pairs left out on purpose (the model says why)
    for a in range(ngeom):
        for b_ in range(a + 1, ngeom):
            ga, gb = geoms[a], geoms[b_]
            if not ((ga["contype"] & gb["conaffinity"]) or (gb["contype"] & ga["conaffinity"])):
                continue
            ba, bb = ga["body"], gb["body"]
            if ba == bb:
                continue
            # parent-child filter; geoms of the world body stay collidable with everything
            if ba != 0 and bb != 0 and (m.body_parentid[ba] == bb or m.body_parentid[bb] == ba):
                continue
            if ga["type"] == PLANE and gb["type"] == PLANE:
                continue
            if (ga["name"], gb["name"]) in excluded or (gb["name"], ga["name"]) in excluded:
                continue
            mix = ga["solmix"] / (ga["solmix"] + gb["solmix"])
            P["g1"].append(a), P["g2"].append(b_)
            P["condim"].append(max(ga["condim"], gb["condim"]))
            P["friction"].append(np.maximum(np.array(ga["friction"], dtype=np.float64), np.array(gb["friction"], dtype=np.float64)))
            P["margin"].append(max(ga["margin"], gb["margin"]))
            P["solref"].append(mix * np.array(ga["solref"], dtype=np.float64) + (1 - mix) * np.array(gb["solref"], dtype=np.float64))
            P["solimp"].append(mix * np.array(_complete_solimp(ga["solimp"])) + (1 - mix) * np.array(_complete_solimp(gb["solimp"])))
    m.pair_geom1, m.pair_geom2 = np.array(P["g1"], dtype=np.int32), np.array(P["g2"], dtype=np.int32)
    m.pair_condim = np.array(P["condim"], dtype=np.int32)
    m.pair_friction, m.pair_margin = np.array(P["friction"]).reshape(-1, 3), np.array(P["margin"])
    m.pair_solref, m.pair_solimp = np.array(P["solref"]).reshape(-1, 2), np.array(P["solimp"]).reshape(-1, 5)

    # ---- invweight0 at qpos0 --------------------------------------------------------------------------------------
    kin = kinematics(m, m.qpos0)
    M = mass_matrix(m, kin)
    Minv = np.linalg.inv(M)
    m.dof_invweight0 = np.zeros(m.nv)
    for j in range(m.njnt):
        a = m.jnt_dofadr[j]
        if m.jnt_type[j] == FREE:
            m.dof_invweight0[a:a + 3] = np.mean(np.diag(Minv)[a:a + 3])
            m.dof_invweight0[a + 3:a + 6] = np.mean(np.diag(Minv)[a + 3:a + 6])
        else:
            m.dof_invweight0[a] = Minv[a, a]
    m.body_invweight0 = np.zeros((nbody, 2))
    for bi in range(1, nbody):
        Jp, Jr = jacobian(m, kin, bi, kin["xipos"][bi])
        A = np.vstack([Jp, Jr]) @ Minv @ np.vstack([Jp, Jr]).T
        m.body_invweight0[bi] = [max(MINVAL, np.trace(A[:3, :3]) / 3), max(MINVAL, np.trace(A[3:, 3:]) / 3)]
    m.meaninertia = float(np.mean(np.diag(M)))
    m.total_mass = float(m.body_mass.sum())
    return m


# ---- reference-style numpy kinematics (used for invweight0 and as a cross-check of the engine) -------------------
def kinematics(m, qpos):
    nb = m.nbody
    xpos, xquat, xmat = np.zeros((nb, 3)), np.zeros((nb, 4)), np.zeros((nb, 3, 3))
    xquat[0], xmat[0] = [1, 0, 0, 0], np.eye(3)
    xanchor, xaxis = np.zeros((m.njnt, 3)), np.zeros((m.njnt, 3))
    for b in range(1, nb):
        p = m.body_parentid[b]
        jn, ja = m.body_jntnum[b], m.body_jntadr[b]
        if jn == 1 and m.jnt_type[ja] == FREE:
            qa = m.jnt_qposadr[ja]
            pos, quat = qpos[qa:qa + 3].copy(), qpos[qa + 3:qa + 7] / np.linalg.norm(qpos[qa + 3:qa + 7])
            xanchor[ja], xaxis[ja] = pos, [0, 0, 1]
        else:
            pos = xpos[p] + xmat[p] @ m.body_pos[b]
            quat = quat_mul(xquat[p], m.body_quat[b])
            for j in range(ja, ja + jn):
                R = quat_to_mat(quat)
                xanchor[j] = pos + R @ m.jnt_pos[j]
                xaxis[j] = R @ m.jnt_axis[j]
                q = qpos[m.jnt_qposadr[j]] - m.qpos0[m.jnt_qposadr[j]]
                if m.jnt_type[j] == HINGE:
                    quat = quat_mul(quat, axisangle_to_quat(m.jnt_axis[j], q))
                    pos = xanchor[j] - quat_to_mat(quat) @ m.jnt_pos[j]
                else:
                    pos = pos + xaxis[j] * q
        quat = quat / np.linalg.norm(quat)
        xpos[b], xquat[b], xmat[b] = pos, quat, quat_to_mat(quat)
    xipos = np.array([xpos[b] + xmat[b] @ m.body_ipos[b] for b in range(nb)])
    return dict(xpos=xpos, xquat=xquat, xmat=xmat, xipos=xipos, xanchor=xanchor, xaxis=xaxis)


def jacobian(m, kin, body, point):
    """(3 x nv translational, 3 x nv rotational) Jacobian of `point` moving with `body` (mj_jac)."""
    Jp, Jr = np.zeros((3, m.nv)), np.zeros((3, m.nv))
    b = body
    while b > 0:
        for j in range(m.body_jntadr[b], m.body_jntadr[b] + m.body_jntnum[b]) if m.body_jntnum[b] else []:
            a = m.jnt_dofadr[j]
            t = m.jnt_type[j]
            if t == FREE:
                Jp[:, a:a + 3] = np.eye(3)
                R = kin["xmat"][b]
                for k in range(3):
                    Jr[:, a + 3 + k] = R[:, k]
                    Jp[:, a + 3 + k] = np.cross(R[:, k], point - kin["xpos"][b])
            elif t == HINGE:
                Jr[:, a] = kin["xaxis"][j]
                Jp[:, a] = np.cross(kin["xaxis"][j], point - kin["xanchor"][j])
            else:
                Jp[:, a] = kin["xaxis"][j]
        b = m.body_parentid[b]
    return Jp, Jr


def mass_matrix(m, kin):
    M = np.diag(m.dof_armature.copy())
    for b in range(1, m.nbody):
        Jp, Jr = jacobian(m, kin, b, kin["xipos"][b])
        Iw = kin["xmat"][b] @ m.body_inertia[b] @ kin["xmat"][b].T
        M += m.body_mass[b] * Jp.T @ Jp + Jr.T @ Iw @ Jr
    return M
