"""Emit the compiled robot models as C++ constexpr tables for the HIP engine (gymnasium_amd/csrc/generated/mjx_models.h).

The device code is specialised per robot at compile time: tree topology, joint axes, inertias, contact pairs and solver
parameters are constants the compiler can fold and unroll on, not data fetched per step.  `python -m
gymnasium_amd.envs.mujoco.codegen` regenerates the header (gymnasium_amd/csrc/build.py does it before compiling).
"""
from __future__ import annotations

import os

import numpy as np

from . import compiler

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "csrc", "generated", "mjx_models.h")
STRUCT = {"half_cheetah": "HalfCheetahModel", "ant": "AntModel", "humanoid": "HumanoidModel", "hopper": "HopperModel", "walker2d": "Walker2dModel",
          "inverted_pendulum": "InvertedPendulumModel", "inverted_double_pendulum": "InvertedDoublePendulumModel", "reacher": "ReacherModel", "humanoid_standup": "HumanoidStandupModel", "swimmer": "SwimmerModel", "pusher": "PusherModel"}


def _arr(name, ctype, values, shape):
    v = np.asarray(values).reshape(-1)
    if shape[0] == 0:  # C++ has no zero-length arrays: one zero row, the count constant (NPAIR, NSLOT) says it is unused
        shape = (1,) + tuple(shape[1:])
        v = np.zeros(int(np.prod(shape)))
    dims = "".join(f"[{d}]" for d in shape)
    if ctype == "double":
        body = ", ".join(float(x).hex() if np.isfinite(x) else ("INFINITY" if x > 0 else "-INFINITY") for x in v)
    else:
        body = ", ".join(str(int(x)) for x in v)
    return f"    static constexpr {ctype} {name}{dims} = {{{body}}};\n"


def emit_model(m) -> str:
    nb, nj, nv, ng, npair, nu = m.nbody, m.njnt, m.nv, m.ngeom, len(m.pair_geom1), m.nu
    maxdepth = 0
    for i in range(nv):
        d, j = 0, i
        while j >= 0:
            d, j = d + 1, m.dof_parentid[j]
        maxdepth = max(maxdepth, d)
    s = f"struct {STRUCT[m.name]} {{\n"
    s += (f"    static constexpr int NQ = {m.nq}, NV = {nv}, NU = {nu}, NBODY = {nb}, NJNT = {nj}, NGEOM = {ng}, NPAIR = {npair}, "
          f"MAXCHAIN = {maxdepth};\n")
    s += f"    static constexpr int INTEGRATOR = {1 if m.integrator == 'RK4' else 0}, SOLVER = {1 if m.reference_solver == 'PGS' else 0}, ITERATIONS = {m.iterations};\n"
    s += f"    static constexpr double TIMESTEP = {float(m.timestep).hex()}, MEANINERTIA = {float(m.meaninertia).hex()};\n"
    s += f"    static constexpr double DENSITY = {float(m.density).hex()}, VISCOSITY = {float(m.viscosity).hex()};  // fluid (inertia-box model)\n"
    s += _arr("gravity", "double", m.gravity, (3,))
    for name, ct, val, shape in [
        ("body_parentid", "int", m.body_parentid, (nb,)), ("body_rootid", "int", m.body_rootid, (nb,)),
        ("body_jntadr", "int", m.body_jntadr, (nb,)), ("body_jntnum", "int", m.body_jntnum, (nb,)),
        ("body_dofadr", "int", m.body_dofadr, (nb,)), ("body_dofnum", "int", m.body_dofnum, (nb,)),
        ("body_pos", "double", m.body_pos, (nb, 3)), ("body_quat", "double", m.body_quat, (nb, 4)),
        ("body_mass", "double", m.body_mass, (nb,)), ("body_ipos", "double", m.body_ipos, (nb, 3)),
        ("body_inertia", "double", m.body_inertia, (nb, 9)), ("body_invweight0", "double", m.body_invweight0, (nb, 2)),
        ("body_fluidbox", "double", m.body_fluidbox, (nb, 3)), ("body_imat", "double", m.body_imat, (nb, 9)),
        ("jnt_type", "int", m.jnt_type, (nj,)), ("jnt_qposadr", "int", m.jnt_qposadr, (nj,)), ("jnt_dofadr", "int", m.jnt_dofadr, (nj,)),
        ("jnt_bodyid", "int", m.jnt_bodyid, (nj,)), ("jnt_limited", "int", m.jnt_limited, (nj,)),
        ("jnt_pos", "double", m.jnt_pos, (nj, 3)), ("jnt_axis", "double", m.jnt_axis, (nj, 3)), ("jnt_range", "double", m.jnt_range, (nj, 2)),
        ("jnt_stiffness", "double", m.jnt_stiffness, (nj,)), ("jnt_margin", "double", m.jnt_margin, (nj,)),
        ("jnt_solref", "double", m.jnt_solref, (nj, 2)), ("jnt_solimp", "double", m.jnt_solimp, (nj, 5)),
        ("dof_bodyid", "int", m.dof_bodyid, (nv,)), ("dof_jntid", "int", m.dof_jntid, (nv,)), ("dof_parentid", "int", m.dof_parentid, (nv,)),
        ("dof_armature", "double", m.dof_armature, (nv,)), ("dof_damping", "double", m.dof_damping, (nv,)),
        ("dof_invweight0", "double", m.dof_invweight0, (nv,)), ("qpos0", "double", m.qpos0, (m.nq,)),
        ("geom_type", "int", m.geom_type, (ng,)), ("geom_bodyid", "int", m.geom_bodyid, (ng,)), ("geom_size", "double", m.geom_size, (ng, 3)),
        ("geom_pos", "double", m.geom_pos, (ng, 3)), ("geom_mat", "double", m.geom_mat, (ng, 9)),
        ("pair_geom1", "int", m.pair_geom1, (npair,)), ("pair_geom2", "int", m.pair_geom2, (npair,)), ("pair_condim", "int", m.pair_condim, (npair,)),
        ("pair_friction", "double", m.pair_friction[:, 0] if npair else np.zeros(0), (npair,)), ("pair_margin", "double", m.pair_margin, (npair,)),
        ("pair_solref", "double", m.pair_solref, (npair, 2)), ("pair_solimp", "double", m.pair_solimp, (npair, 5)),
        ("actuator_dofadr", "int", m.actuator_dofadr, (nu,)), ("actuator_gear", "double", m.actuator_gear, (nu,)),
        ("actuator_ctrlrange", "double", m.actuator_ctrlrange, (nu, 2)),
    ]:
        s += _arr(name, ct, val, shape)
    s += _emit_topology(m)
    sites = getattr(m, "sites", [])
    s += f"    static constexpr int NSITE = {len(sites)};\n"
    s += _arr("site_bodyid", "int", [b for b, _ in sites], (len(sites),))
    s += _arr("site_pos", "double", [p for _, p in sites], (len(sites), 3))
    nt = m.ntendon
    maxwrap = max([len(w) for w in m.tendon_wraps], default=0)
    s += f"    static constexpr int NTENDON = {nt}, MAXWRAP = {max(maxwrap, 1)};  // fixed tendons (joint wraps): info values only\n"
    pad = lambda rows, fill: [list(r) + [fill] * (max(maxwrap, 1) - len(r)) for r in rows]  # noqa: E731
    s += _arr("tendon_num", "int", [len(w) for w in m.tendon_wraps], (nt,))
    s += _arr("tendon_qposadr", "int", pad([[x[0] for x in w] for w in m.tendon_wraps], 0), (nt, max(maxwrap, 1)))
    s += _arr("tendon_dofadr", "int", pad([[x[1] for x in w] for w in m.tendon_wraps], 0), (nt, max(maxwrap, 1)))
    s += _arr("tendon_coef", "double", pad([[x[2] for x in w] for w in m.tendon_wraps], 0.0), (nt, max(maxwrap, 1)))
    s += "};\n"
    return s


def _emit_topology(m) -> str:
    """Tables for the cooperative kernel (mjx_coop.h): tree depth, ancestor / descendant masks, static contact slots."""
    nb, nv, nj, npair = m.nbody, m.nv, m.njnt, len(m.pair_geom1)
    assert nv <= 32 and nb <= 32
    depth = [0] * nb
    for b in range(1, nb):
        depth[b] = depth[m.body_parentid[b]] + 1
    # dofs on the path root -> body (bodies without joints inherit their parent's chain)
    body_dofmask = [0] * nb
    for b in range(1, nb):
        mask = body_dofmask[m.body_parentid[b]]
        for k in range(m.body_dofnum[b]):
            mask |= 1 << (m.body_dofadr[b] + k)
        body_dofmask[b] = mask
    body_descmask = [0] * nb
    for b in range(nb - 1, 0, -1):
        body_descmask[b] |= 1 << b
        p = m.body_parentid[b]
        if p > 0:
            body_descmask[p] |= body_descmask[b]
    dof_ancmask = [0] * nv
    for i in range(nv):
        j = i
        while j >= 0:
            dof_ancmask[i] |= 1 << j
            j = m.dof_parentid[j]
    dof_descbodies = [body_descmask[m.dof_bodyid[i]] for i in range(nv)]
    dof_actuator = [-1] * nv
    for u in range(m.nu):
        assert dof_actuator[m.actuator_dofadr[u]] == -1, "one actuator per dof"
        dof_actuator[m.actuator_dofadr[u]] = u
    # static contact slots in the order the serial code emits contacts: pair order; plane-capsule pairs give two (+h, -h)
    slot_pair, slot_sub = [], []
    for p in range(npair):
        g1, g2 = m.pair_geom1[p], m.pair_geom2[p]
        t1, t2 = sorted((m.geom_type[g1], m.geom_type[g2]))
        n = 2 if (t1 == compiler.PLANE and t2 == compiler.CAPSULE) else 1
        for k in range(n):
            slot_pair.append(p), slot_sub.append(k)
    nslot = len(slot_pair)
    s = (f"    static constexpr int MAXDEPTH = {max(depth)}, MAXJPB = {max(m.body_jntnum)}, NSLOT = {nslot}, "
         f"MAXCON = {min(nslot, 40)};\n")
    for name, val, n in [("body_depth", depth, nb), ("body_dofmask", body_dofmask, nb), ("body_descmask", body_descmask, nb),
                         ("dof_ancmask", dof_ancmask, nv), ("dof_descbodies", dof_descbodies, nv), ("dof_actuator", dof_actuator, nv),
                         ("slot_pair", slot_pair, nslot), ("slot_sub", slot_sub, nslot)]:
        s += _arr(name, "int", val, (n,))
    return s


def generate(path: str = OUT) -> str:
    text = ("// mjx_models.h -- GENERATED by gymnasium_amd/envs/mujoco/codegen.py from gymnasium_amd/envs/mujoco/models.py (a transcription of\n"
            "// the reference's MJCF assets, gymnasium/envs/mujoco/assets/{half_cheetah,ant,humanoid}.xml).  Do not edit.\n"
            "#pragma once\n#include <math.h>\nnamespace mjx {\n")
    for name in STRUCT:
        text += emit_model(compiler.compile_model(name))
    text += "}  // namespace mjx\n"
    os.makedirs(os.path.dirname(path), exist_ok=True)
    old = open(path).read() if os.path.exists(path) else None
    if old != text:
        with open(path, "w") as f:
            f.write(text)
    return os.path.normpath(path)


if __name__ == "__main__":
    print(generate())
