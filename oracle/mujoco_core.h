/*
 * mujoco_core.h -- TEST INFRASTRUCTURE (CPU oracle).  Plain-C restatement of the MuJoCo computation pipeline for the
 * feature set used by the reference's HalfCheetah-v5 / Ant-v5 / Humanoid-v5 models.
 *
 * PARITY UNPINNED.  The physics of these environments is not in the reference tree: it is the third-party `mujoco`
 * C library (reference pyproject.toml:45 `mujoco >= 2.1.5`, no pinned version; call sites
 * gymnasium/envs/mujoco/mujoco_env.py:124-155,180: MjModel.from_xml_path, mj_forward, mj_step(nstep), mj_rnePostConstraint,
 * mj_resetData).  `mujoco` is not installed in the build container, there is no network, and no reference test holds a
 * numeric trajectory (tests/envs/mujoco/test_mujoco_v5.py checks counts / shapes / self-consistency only).  This file
 * therefore restates MuJoCo's PUBLISHED algorithm (documentation "Computation" chapter: kinematics in com-based spatial
 * coordinates, composite-rigid-body mass matrix, recursive Newton-Euler bias, soft constraints with solref / solimp
 * impedance, pyramidal friction cones, primal Newton solver, semi-implicit Euler with implicit joint damping, RK4) and is
 * anchored on (a) the reference's own call sites and Python glue, (b) the model counts the reference tests pin, and
 * (c) physical invariants (tests/test_mujoco_oracle.py).  It cannot claim bit- or tolerance-level parity with `mujoco`
 * until fixtures from a real build exist.
 */
#ifndef ORACLE_MUJOCO_CORE_H
#define ORACLE_MUJOCO_CORE_H

#include <stdint.h>

#define MJO_MAXB 16   /* bodies incl. world */
#define MJO_MAXV 24   /* dofs */
#define MJO_MAXQ 25
#define MJO_MAXJ 20
#define MJO_MAXG 20
#define MJO_MAXU 20
#define MJO_MAXPAIR 136
#define MJO_MAXCON 40
#define MJO_MAXEFC (4 * MJO_MAXCON + MJO_MAXJ)
#define MJO_MAXT 4    /* fixed tendons */
#define MJO_MAXWRAP 4 /* joints per fixed tendon */

enum { MJO_FREE = 0, MJO_BALL = 1, MJO_SLIDE = 2, MJO_HINGE = 3 };
enum { MJO_PLANE = 0, MJO_SPHERE = 2, MJO_CAPSULE = 3, MJO_CYLINDER = 5 };
enum { MJO_EULER = 0, MJO_RK4 = 1 };
enum { MJO_NEWTON = 0, MJO_PGS = 1 };

typedef struct mjo_model {
    int nq, nv, nu, nbody, njnt, ngeom, npair, integrator, solver, iterations;
    double timestep, gravity[3], meaninertia, density, viscosity;
    int body_parentid[MJO_MAXB], body_rootid[MJO_MAXB], body_jntadr[MJO_MAXB], body_jntnum[MJO_MAXB], body_dofadr[MJO_MAXB],
        body_dofnum[MJO_MAXB];
    double body_pos[MJO_MAXB][3], body_quat[MJO_MAXB][4], body_mass[MJO_MAXB], body_ipos[MJO_MAXB][3], body_inertia[MJO_MAXB][9],
        body_invweight0[MJO_MAXB][2], body_fluidbox[MJO_MAXB][3], body_imat[MJO_MAXB][9];
    int jnt_type[MJO_MAXJ], jnt_qposadr[MJO_MAXJ], jnt_dofadr[MJO_MAXJ], jnt_bodyid[MJO_MAXJ], jnt_limited[MJO_MAXJ];
    double jnt_pos[MJO_MAXJ][3], jnt_axis[MJO_MAXJ][3], jnt_range[MJO_MAXJ][2], jnt_stiffness[MJO_MAXJ], jnt_margin[MJO_MAXJ],
        jnt_solref[MJO_MAXJ][2], jnt_solimp[MJO_MAXJ][5];
    int dof_bodyid[MJO_MAXV], dof_jntid[MJO_MAXV], dof_parentid[MJO_MAXV];
    double dof_armature[MJO_MAXV], dof_damping[MJO_MAXV], dof_invweight0[MJO_MAXV];
    double qpos0[MJO_MAXQ], qpos_spring[MJO_MAXQ];
    int geom_type[MJO_MAXG], geom_bodyid[MJO_MAXG];
    double geom_size[MJO_MAXG][3], geom_pos[MJO_MAXG][3], geom_mat[MJO_MAXG][9];
    int pair_geom1[MJO_MAXPAIR], pair_geom2[MJO_MAXPAIR], pair_condim[MJO_MAXPAIR];
    double pair_friction[MJO_MAXPAIR][3], pair_margin[MJO_MAXPAIR], pair_solref[MJO_MAXPAIR][2], pair_solimp[MJO_MAXPAIR][5];
    int actuator_dofadr[MJO_MAXU];
    double actuator_gear[MJO_MAXU], actuator_ctrlrange[MJO_MAXU][2];
    /* fixed tendons (humanoid.xml:91-100): joint wraps only, no dynamics attached */
    int ntendon, tendon_num[MJO_MAXT], wrap_qposadr[MJO_MAXT][MJO_MAXWRAP], wrap_dofadr[MJO_MAXT][MJO_MAXWRAP];
    double wrap_coef[MJO_MAXT][MJO_MAXWRAP];
} mjo_model;

typedef struct mjo_contact {
    double dist, pos[3], frame[9], friction, margin, solref[2], solimp[5];
    int geom1, geom2, dim, efc_address;
} mjo_contact;

typedef struct mjo_data {
    /* state */
    double qpos[MJO_MAXQ], qvel[MJO_MAXV], ctrl[MJO_MAXU];
    /* position stage */
    double xpos[MJO_MAXB][3], xquat[MJO_MAXB][4], xmat[MJO_MAXB][9], xipos[MJO_MAXB][3], xanchor[MJO_MAXJ][3], xaxis[MJO_MAXJ][3];
    double geom_xpos[MJO_MAXG][3], geom_xmat[MJO_MAXG][9];
    double subtree_com[MJO_MAXB][3], cinert[MJO_MAXB][10], cdof[MJO_MAXV][6];
    double qM[MJO_MAXV][MJO_MAXV], qL[MJO_MAXV][MJO_MAXV]; /* dense mass matrix and its Cholesky factor */
    int ncon, nefc;
    mjo_contact contact[MJO_MAXCON];
    double efc_J[MJO_MAXEFC][MJO_MAXV], efc_pos[MJO_MAXEFC], efc_margin[MJO_MAXEFC], efc_D[MJO_MAXEFC], efc_R[MJO_MAXEFC],
        efc_vel[MJO_MAXEFC], efc_aref[MJO_MAXEFC], efc_force[MJO_MAXEFC], efc_KBIP[MJO_MAXEFC][4];
    /* velocity stage */
    double cvel[MJO_MAXB][6], cdof_dot[MJO_MAXV][6];
    double qfrc_passive[MJO_MAXV], qfrc_bias[MJO_MAXV], qfrc_actuator[MJO_MAXV], qfrc_smooth[MJO_MAXV], qacc_smooth[MJO_MAXV];
    double qfrc_constraint[MJO_MAXV], qacc[MJO_MAXV], qacc_warmstart[MJO_MAXV];
    double cfrc_ext[MJO_MAXB][6];
    double ten_length[MJO_MAXT], ten_velocity[MJO_MAXT]; /* mj_tendon / mj_fwdVelocity: fixed tendons, at this forward pass's state */
    int solver_iter;
} mjo_data;

/* model blob produced by oracle/mujoco.py (doubles; see mjo_model_from_blob) */
int mjo_model_from_blob(mjo_model *m, const double *blob, int n);
void mjo_reset_data(const mjo_model *m, mjo_data *d);              /* mj_resetData: qpos = qpos0, qvel = 0, ctrl = 0 */
void mjo_forward(const mjo_model *m, mjo_data *d);                 /* mj_forward */
void mjo_step(const mjo_model *m, mjo_data *d, int nstep);         /* mj_step(nstep) */
void mjo_rne_post_constraint(const mjo_model *m, mjo_data *d);     /* mj_rnePostConstraint: fills cfrc_ext */

#endif
