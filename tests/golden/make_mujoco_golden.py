#!/usr/bin/env python3
"""Generate MuJoCo fixtures FROM A REAL `mujoco` BUILD -- the missing pin of the MuJoCo half.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_mujoco_golden.py

Needs `import mujoco` (any version the reference accepts: pyproject.toml `mujoco >= 2.1.5`) and the reference gymnasium
(GYM_REFERENCE, default /root/reference).  The build container of rounds 1-2 has no `mujoco` wheel and no network, so the
fixtures do not exist yet; this script and its consumer (tests/test_mujoco_fixtures.py, which skips while
tests/golden/mujoco_<robot>.npz is absent) are committed so that the pin activates the moment a wheel is available.
Nothing here is read by the product.

Per robot (the eleven v5 ids) one file tests/golden/mujoco_<robot>.npz:

  model_*        compiled mjModel arrays (body_mass, body_inertia, body_ipos, body_iquat, body_pos, body_quat, jnt_*, dof_armature,
                 dof_damping, dof_invweight0, body_invweight0, geom_size, geom_pos, geom_quat, geom_friction, geom_margin,
                 actuator_gear, actuator_ctrlrange, qpos0, opt.*, stat.meaninertia) -> checks envs/mujoco/compiler.py field by field
  fwd_*          K seeded states (qpos around init_qpos incl. some pushed into the floor, qvel, ctrl) and mj_forward's
                 intermediates: xpos, xquat, xipos, cinert, cdof, cvel, full qM (mj_fullM), qfrc_bias, qfrc_passive,
                 qfrc_actuator, qacc_smooth, contacts (dist, pos, frame, geoms), efc_J / efc_D / efc_R / efc_aref / efc_force,
                 qfrc_constraint, qacc, ten_length, ten_velocity, and cfrc_ext after mj_rnePostConstraint
  traj_*         gym.make(id) scalar env: reset(seed=0), action_space.seed(0), 100 steps: actions, obs, reward, terminated,
                 truncated, the numeric info entries, and data.qpos / qvel / qacc_warmstart / cfrc_ext after every step
"""
import os
import sys

REF = os.environ.get("GYM_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import numpy as np  # noqa: E402

OUT = os.environ.get("MUJOCO_GOLDEN_OUT") or os.path.dirname(os.path.abspath(__file__))  # (the override is for tests/test_mujoco_fixture_pipeline.py)
IDS = {"half_cheetah": "HalfCheetah-v5", "ant": "Ant-v5", "humanoid": "Humanoid-v5", "humanoid_standup": "HumanoidStandup-v5",
       "hopper": "Hopper-v5", "walker2d": "Walker2d-v5", "inverted_pendulum": "InvertedPendulum-v5",
       "inverted_double_pendulum": "InvertedDoublePendulum-v5", "reacher": "Reacher-v5", "swimmer": "Swimmer-v5", "pusher": "Pusher-v5"}
MODEL_FIELDS = ["body_parentid", "body_mass", "body_inertia", "body_ipos", "body_iquat", "body_pos", "body_quat", "body_invweight0", "jnt_type",
                "jnt_qposadr", "jnt_dofadr", "jnt_bodyid", "jnt_limited", "jnt_pos", "jnt_axis", "jnt_range", "jnt_stiffness", "jnt_margin",
                "jnt_solref", "jnt_solimp", "dof_bodyid", "dof_jntid", "dof_parentid", "dof_armature", "dof_damping", "dof_invweight0",
                "qpos0", "qpos_spring", "geom_type", "geom_bodyid", "geom_size", "geom_pos", "geom_quat", "geom_friction", "geom_margin",
                "geom_contype", "geom_conaffinity", "geom_condim", "geom_solref", "geom_solimp", "actuator_gear", "actuator_ctrlrange",
                "actuator_trnid"]
FWD_FIELDS = ["xpos", "xquat", "xmat", "xipos", "ximat", "subtree_com", "cinert", "cdof", "cvel", "qfrc_bias", "qfrc_passive", "qfrc_actuator",
              "qfrc_smooth", "qacc_smooth", "qfrc_constraint", "qacc", "ten_length", "ten_velocity"]


def fixtures_for(name, env_id, mujoco, gym, K=24, T=100):
    env = gym.make(env_id).unwrapped
    model, data = env.model, env.data
    out = {"mujoco_version": np.array(mujoco.__version__), "nq": model.nq, "nv": model.nv, "nu": model.nu, "nbody": model.nbody, "ngeom": model.ngeom,
           "opt_timestep": model.opt.timestep, "opt_gravity": np.array(model.opt.gravity), "opt_integrator": int(model.opt.integrator),
           "opt_solver": int(model.opt.solver), "opt_iterations": int(model.opt.iterations), "opt_tolerance": float(model.opt.tolerance),
           "opt_cone": int(model.opt.cone), "opt_density": float(model.opt.density), "opt_viscosity": float(model.opt.viscosity),
           "stat_meaninertia": float(model.stat.meaninertia), "frame_skip": int(env.frame_skip)}
    for f in MODEL_FIELDS:
        if hasattr(model, f):
            out["model_" + f] = np.array(getattr(model, f))
    # ---- forward intermediates at seeded states -------------------------------------------------------------------------------
    rng = np.random.default_rng(12345)
    rec = {k: [] for k in ["qpos", "qvel", "ctrl", "qM", "ncon", "nefc", "cfrc_ext"] + FWD_FIELDS}
    con_rows, efc_rows = [], []
    for k in range(K):
        mujoco.mj_resetData(model, data)
        qpos = env.init_qpos + rng.uniform(-0.1, 0.1, model.nq) * (0.0 if k == 0 else 1.0)
        if model.nq > model.nv:  # a free joint: random orientation, and every third state is pushed towards the floor
            q = rng.normal(size=4)
            qpos[3:7] = q / np.linalg.norm(q) if k % 2 else env.init_qpos[3:7]
            if k % 3 == 2:
                qpos[2] -= rng.uniform(0.2, 0.6)
        qvel = rng.normal(size=model.nv) * (0.0 if k == 0 else 0.5)
        lo, hi = model.actuator_ctrlrange[:, 0], model.actuator_ctrlrange[:, 1]
        data.qpos[:], data.qvel[:], data.ctrl[:] = qpos, qvel, rng.uniform(lo, hi)
        mujoco.mj_forward(model, data)
        mujoco.mj_rnePostConstraint(model, data)
        M = np.zeros((model.nv, model.nv))
        mujoco.mj_fullM(model, M, data.qM)
        rec["qpos"].append(qpos), rec["qvel"].append(qvel), rec["ctrl"].append(np.array(data.ctrl)), rec["qM"].append(M)
        rec["ncon"].append(data.ncon), rec["nefc"].append(data.nefc), rec["cfrc_ext"].append(np.array(data.cfrc_ext))
        for f in FWD_FIELDS:
            rec[f].append(np.array(getattr(data, f)))
        for c in range(data.ncon):
            con = data.contact[c]
            g1, g2 = (con.geom[0], con.geom[1]) if hasattr(con, "geom") else (con.geom1, con.geom2)
            con_rows.append(np.concatenate([[k, con.dist], con.pos, con.frame, [g1, g2, con.dim, con.efc_address, con.friction[0], con.includemargin]]))
        if data.nefc:
            J = np.array(data.efc_J).reshape(data.nefc, -1)
            if J.shape[1] != model.nv:  # sparse storage: densify through mj's own helper where it exists
                J = np.zeros((data.nefc, model.nv))
                mujoco.mju_sparse2dense(J, data.efc_J, data.efc_J_rownnz, data.efc_J_rowadr, data.efc_J_colind)
            for r in range(data.nefc):
                efc_rows.append(np.concatenate([[k, r, data.efc_type[r], data.efc_pos[r], data.efc_margin[r], data.efc_D[r], data.efc_R[r], data.efc_aref[r],
                                                 data.efc_force[r]], J[r]]))
    for f, v in rec.items():
        out["fwd_" + f] = np.stack([np.asarray(x) for x in v])
    out["fwd_contacts"] = np.stack(con_rows) if con_rows else np.zeros((0, 20))     # state k, dist, pos3, frame9, geom1, geom2, dim, efc_address, mu, margin
    out["fwd_efc"] = np.stack(efc_rows) if efc_rows else np.zeros((0, 9 + model.nv))  # state k, row, type, pos, margin, D, R, aref, force, J[nv]
    # ---- a 100-step trajectory of the scalar env (TimeLimit-free: the unwrapped env) ---------------------------------------------
    env.action_space.seed(0)
    obs0, info0 = env.reset(seed=0)
    tr = {k: [] for k in ("actions", "obs", "reward", "terminated", "qpos", "qvel", "qacc_warmstart", "cfrc_ext", "ten_length", "ten_velocity")}
    infos = {}
    out["traj_obs0"], out["traj_qpos0"], out["traj_qvel0"] = np.asarray(obs0), np.array(data.qpos), np.array(data.qvel)
    for t in range(T):
        a = env.action_space.sample()
        o, r, te, _, info = env.step(a)
        tr["actions"].append(a), tr["obs"].append(o), tr["reward"].append(r), tr["terminated"].append(te)
        tr["qpos"].append(np.array(data.qpos)), tr["qvel"].append(np.array(data.qvel)), tr["qacc_warmstart"].append(np.array(data.qacc_warmstart))
        tr["cfrc_ext"].append(np.array(data.cfrc_ext)), tr["ten_length"].append(np.array(data.ten_length)), tr["ten_velocity"].append(np.array(data.ten_velocity))
        for key, val in info.items():
            if np.isscalar(val) or (isinstance(val, np.ndarray) and val.ndim <= 1):
                infos.setdefault(key, []).append(np.asarray(val, dtype=np.float64))
    for k, v in tr.items():
        out["traj_" + k] = np.stack([np.asarray(x) for x in v])
    for k, v in infos.items():
        if len(v) == T:
            out["traj_info_" + k] = np.stack(v)
    env.close()
    path = os.path.join(OUT, f"mujoco_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{os.path.basename(path)}: {os.path.getsize(path) / 1024:.1f} KiB (mujoco {mujoco.__version__})")


def main():
    try:
        import mujoco
    except ImportError as e:
        print(f"mujoco is not importable here ({e}): no fixtures written.  tests/test_mujoco_fixtures.py keeps skipping; "
              "DESIGN.md section 7 keeps saying 'parity unpinned'.")
        return 2
    import gymnasium as gym

    if getattr(mujoco, "IS_ORACLE_SHIM", False) and OUT == os.path.dirname(os.path.abspath(__file__)):
        print("refusing to write oracle-shim fixtures into tests/golden/: they would look like a pin and are none (set MUJOCO_GOLDEN_OUT)")
        return 3
    for name, env_id in IDS.items():
        fixtures_for(name, env_id, mujoco, gym)
    return 0


if __name__ == "__main__":
    sys.exit(main())
