// mjx_core.h -- per-lane articulated-body simulator for gfx950: the MuJoCo computation pipeline specialised at compile
// time for one robot (template parameter M = a generated model table, generated/mjx_models.h).
//
// What it replaces: the `mujoco` C library calls the reference makes on the step path
// (gymnasium/envs/mujoco/mujoco_env.py:142 mj_forward, :150 mj_step(nstep), :155 mj_rnePostConstraint, :180 mj_resetData)
// for the three MJCF models of BASELINE.json's configs (assets/half_cheetah.xml, ant.xml, humanoid.xml).  `mujoco` is a
// third-party dependency that is not in the reference tree; the pipeline implemented here is MuJoCo's published one
//  and its parity is UNPINNED until fixtures from a real `mujoco` build exist.
//
// Execution model (round 1): one sub-environment per lane; every array below is per-lane private state.  Tree topology,
// joint axes, inertias, contact pairs are compile-time constants, so the tree recursions unroll into straight-line code
// with constant operands.  The constraint Jacobian is never stored: a contact row is regenerated from the contact point
// and the motion subspaces (cdof) each time it is needed (flops are cheap here, private memory is not).
//
// Formulation notes (where this differs from a textbook / the CPU oracle, on purpose):
//   * spatial quantities are referred to the subtree centre of mass of the root body ("com-based", as MuJoCo),
//   * the mass matrix and the Newton Hessian are stored packed (lower triangle) and factorised as L L^T,
//   * the primal Newton solver uses a bracketed 1-D Newton line search on the exact piecewise-quadratic cost.
#pragma once
// MJX_HOST_EMU: test-only host build of the same source (tests/coop_emu), never part of the product library.
#if defined(MJX_HOST_EMU)
#include <cmath>
#else
#include <hip/hip_runtime.h>
#endif
#include <math.h>
#include <stdint.h>

#include "generated/mjx_models.h"

// The physics may fuse a*b+c (it is compared with the oracle at 1e-8 .. 1e-10, not bit for bit); the NumPy-exact env glue in
// mjx_kernels.h and the classic-control envs stay unfused: contraction is switched back off at the end of this header.
#if !defined(MJX_HOST_EMU)
#pragma clang fp contract(fast)
#endif

namespace mjx {

#if defined(MJX_HOST_EMU)
#define MJX_DEV inline
#define MJX_DEVN static
#else
#define MJX_DEV __device__ __forceinline__
#define MJX_DEVN __device__ __noinline__
#endif

enum { FREE = 0, BALL = 1, SLIDE = 2, HINGE = 3 };
enum { PLANE = 0, SPHERE = 2, CAPSULE = 3, CYLINDER = 5 };
constexpr double kMinVal = 1e-15, kMinImp = 0.0001, kMaxImp = 0.9999;

// ---- tiny vector algebra ---------------------------------------------------------------------------------------
MJX_DEV double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
MJX_DEV void cross3(double *r, const double *a, const double *b) {
    const double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    r[0] = x, r[1] = y, r[2] = z;
}
MJX_DEV double normalize3(double *a) {
    const double n = sqrt(dot3(a, a));
    if (n < kMinVal) {
        a[0] = 1, a[1] = 0, a[2] = 0;
        return n;
    }
    const double inv = 1.0 / n;
    a[0] *= inv, a[1] *= inv, a[2] *= inv;
    return n;
}
MJX_DEV void rot_vec(double *r, const double *R, const double *v) {  // R row-major 3x3
    const double x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2], y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2],
                 z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
    r[0] = x, r[1] = y, r[2] = z;
}
MJX_DEV void quat_mul(double *r, const double *a, const double *b) {
    const double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                 y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    r[0] = w, r[1] = x, r[2] = y, r[3] = z;
}
MJX_DEV void quat_normalize(double *q) {
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n < kMinVal) {
        q[0] = 1, q[1] = q[2] = q[3] = 0;
        return;
    }
    const double inv = 1.0 / n;
    q[0] *= inv, q[1] *= inv, q[2] *= inv, q[3] *= inv;
}
MJX_DEV void quat_to_mat(double *m, const double *q) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    m[0] = w * w + x * x - y * y - z * z, m[1] = 2 * (x * y - w * z), m[2] = 2 * (x * z + w * y);
    m[3] = 2 * (x * y + w * z), m[4] = w * w - x * x + y * y - z * z, m[5] = 2 * (y * z - w * x);
    m[6] = 2 * (x * z - w * y), m[7] = 2 * (y * z + w * x), m[8] = w * w - x * x - y * y + z * z;
}
// sin and cos of one argument: Cody-Waite reduction by multiples of pi/2 (three-part constant, exact products through FMA) and
// the fdlibm minimax kernels on [-pi/4, pi/4]; < 1 ulp for |x| < 1e5 (joint half-angles and h |omega| live far inside that),
// libm's sincos beyond.  ~45 instructions on the hot path instead of ocml's ~200 (no Payne-Hanek code is ever fetched).
MJX_DEVN void sincos_slow(double x, double *sn, double *cs) { sincos(x, sn, cs); }  // cold: kept out of line
MJX_DEV void sincos_fast(double x, double *sn, double *cs) {
    if (__builtin_expect(!(fabs(x) < 1.0e5), 0)) {
        sincos_slow(x, sn, cs);
        return;
    }
    const double kq = rint(x * 6.36619772367581382433e-01);  // x * 2/pi
    const int q = (int)kq;
    double r = fma(-kq, 1.57079632673412561417e+00, x);      // pi/2 split in three parts (fdlibm pio2_1, pio2_2, pio2_3)
    r = fma(-kq, 6.07710050650619224932e-11, r);
    r = fma(-kq, 2.02226624879595063154e-21, r);
    const double z = r * r;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double ps = fma(z, S6, S5);
    ps = fma(z, ps, S4), ps = fma(z, ps, S3), ps = fma(z, ps, S2), ps = fma(z, ps, S1);
    const double s0 = fma(z * r, ps, r);
    double pc = fma(z, C6, C5);
    pc = fma(z, pc, C4), pc = fma(z, pc, C3), pc = fma(z, pc, C2), pc = fma(z, pc, C1);
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double c0 = w + (((1.0 - w) - hz) + z * (z * pc));
    const bool swap = (q & 1) != 0;
    const double sv = swap ? c0 : s0, cv = swap ? s0 : c0;
    *sn = (q & 2) ? -sv : sv;
    *cs = ((q + 1) & 2) ? -cv : cv;
}
MJX_DEV void axis_angle_quat(double *q, const double *axis, double angle) {
    double s, c;
    sincos_fast(angle * 0.5, &s, &c);
    q[0] = c, q[1] = axis[0] * s, q[2] = axis[1] * s, q[3] = axis[2] * s;
}
// spatial cross products on [rotational; translational] 6-vectors
MJX_DEV void cross_motion(double *r, const double *v, const double *s) {
    double a[3], b[3], c[3];
    cross3(a, v, s), cross3(b, v, s + 3), cross3(c, v + 3, s);
    r[0] = a[0], r[1] = a[1], r[2] = a[2], r[3] = b[0] + c[0], r[4] = b[1] + c[1], r[5] = b[2] + c[2];
}
MJX_DEV void cross_force(double *r, const double *v, const double *f) {
    double a[3], b[3], c[3];
    cross3(a, v, f), cross3(b, v + 3, f + 3), cross3(c, v, f + 3);
    r[0] = a[0] + b[0], r[1] = a[1] + b[1], r[2] = a[2] + b[2], r[3] = c[0], r[4] = c[1], r[5] = c[2];
}
// com-based inertia (Ixx Iyy Izz Ixy Ixz Iyz, m*d, m) times a motion vector
MJX_DEV void inert_mul(double *r, const double *I, const double *v) {
    double c1[3], c2[3];
    cross3(c1, I + 6, v + 3), cross3(c2, I + 6, v);
    r[0] = I[0] * v[0] + I[3] * v[1] + I[4] * v[2] + c1[0];
    r[1] = I[3] * v[0] + I[1] * v[1] + I[5] * v[2] + c1[1];
    r[2] = I[4] * v[0] + I[5] * v[1] + I[2] * v[2] + c1[2];
    r[3] = I[9] * v[3] - c2[0], r[4] = I[9] * v[4] - c2[1], r[5] = I[9] * v[5] - c2[2];
}

constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }  // packed lower triangle, j <= i

template <class M>
struct Limits {
    // worst-case constraint bookkeeping sizes: every limited joint, two contacts per plane-capsule pair, one otherwise
    static constexpr int max_contacts() {
        int n = 0;
        for (int p = 0; p < M::NPAIR; p++)
            n += (M::geom_type[M::pair_geom1[p]] == PLANE && M::geom_type[M::pair_geom2[p]] == CAPSULE) ? 2 : 1;
        return n;
    }
    static constexpr int MAXCON = max_contacts() < 40 ? (max_contacts() > 0 ? max_contacts() : 1) : 40;  // >= 1: C++ has no empty arrays
};

template <class M>
struct Contact {
    double dist, pos[3], frame[9];
    int pair;
};

// per-lane simulation state of one forward evaluation
template <class M>
struct Data {
    static constexpr int NQ = M::NQ, NV = M::NV, NB = M::NBODY, NJ = M::NJNT, NU = M::NU, NTRI = M::NV * (M::NV + 1) / 2;
    static constexpr int MAXCON = Limits<M>::MAXCON;
    double qpos[NQ], qvel[NV], ctrl[NU];
    double xpos[NB][3], xquat[NB][4], xmat[NB][9], xipos[NB][3], xanchor[NJ][3], xaxis[NJ][3];
    double com[3];  // subtree centre of mass of the (single) kinematic tree: the reference point of all spatial quantities
    double cinert[NB][10], cdof[NV][6], cvel[NB][6];
    double qM[NTRI], qL[NTRI];
    double qfrc_smooth[NV], qfrc_actuator[NV], qacc_smooth[NV], qacc[NV], qfrc_constraint[NV];
    double qacc_warm[NV];  // qacc of the previous forward pass (mj qacc_warmstart): where a constrained solve starts
    int ncon, nlimit;
    Contact<M> con[MAXCON];
    // joint-limit rows
    int lim_dof[NJ];
    double lim_sign[NJ], lim_D[NJ], lim_aref[NJ], lim_force[NJ];
    // contact rows: per contact D (shared by its pyramid edges), aref (shared), force per edge
    double con_D[MAXCON], con_aref[MAXCON], con_force[MAXCON][4];
    // fixed tendons (mj_tendon / mj_fwdVelocity for joint wraps): length and velocity at THIS forward pass's qpos / qvel
    double ten_length[M::NTENDON > 0 ? M::NTENDON : 1], ten_velocity[M::NTENDON > 0 ? M::NTENDON : 1];
};

// ten_length[t] = sum_k coef_k qpos[joint_k], ten_velocity[t] = sum_k coef_k qvel[joint_k]  (humanoid.xml:91-100)
template <class M>
MJX_DEV void tendons(const double *qpos, const double *qvel, double *len, double *vel) {
#pragma unroll
    for (int t = 0; t < M::NTENDON; t++) {
        double l = 0, v = 0;
#pragma unroll
        for (int k = 0; k < M::MAXWRAP; k++)
            if (k < M::tendon_num[t]) l += M::tendon_coef[t][k] * qpos[M::tendon_qposadr[t][k]], v += M::tendon_coef[t][k] * qvel[M::tendon_dofadr[t][k]];
        len[t] = l, vel[t] = v;
    }
}

// ---- position stage -----------------------------------------------------------------------------------------------
template <class M>
MJX_DEV void kinematics(Data<M> &d) {
    d.xpos[0][0] = d.xpos[0][1] = d.xpos[0][2] = 0, d.xquat[0][0] = 1, d.xquat[0][1] = d.xquat[0][2] = d.xquat[0][3] = 0;
    quat_to_mat(d.xmat[0], d.xquat[0]);
    d.xipos[0][0] = d.xipos[0][1] = d.xipos[0][2] = 0;
#pragma unroll
    for (int b = 1; b < M::NBODY; b++) {
        const int p = M::body_parentid[b], ja = M::body_jntadr[b], jn = M::body_jntnum[b];
        double pos[3], quat[4];
        if (jn == 1 && M::jnt_type[ja] == FREE) {
            const int qa = M::jnt_qposadr[ja];
            pos[0] = d.qpos[qa], pos[1] = d.qpos[qa + 1], pos[2] = d.qpos[qa + 2];
            quat[0] = d.qpos[qa + 3], quat[1] = d.qpos[qa + 4], quat[2] = d.qpos[qa + 5], quat[3] = d.qpos[qa + 6];
            quat_normalize(quat);
            d.xanchor[ja][0] = pos[0], d.xanchor[ja][1] = pos[1], d.xanchor[ja][2] = pos[2];
            d.xaxis[ja][0] = 0, d.xaxis[ja][1] = 0, d.xaxis[ja][2] = 1;
        } else {
            double t[3];
            rot_vec(t, d.xmat[p], M::body_pos[b]);
            pos[0] = d.xpos[p][0] + t[0], pos[1] = d.xpos[p][1] + t[1], pos[2] = d.xpos[p][2] + t[2];
            quat_mul(quat, d.xquat[p], M::body_quat[b]);
#pragma unroll
            for (int j = ja; j < ja + jn; j++) {
                double R[9], ql[4];
                quat_to_mat(R, quat);
                rot_vec(t, R, M::jnt_pos[j]);
                d.xanchor[j][0] = pos[0] + t[0], d.xanchor[j][1] = pos[1] + t[1], d.xanchor[j][2] = pos[2] + t[2];
                rot_vec(d.xaxis[j], R, M::jnt_axis[j]);
                const double q = d.qpos[M::jnt_qposadr[j]] - M::qpos0[M::jnt_qposadr[j]];
                if (M::jnt_type[j] == HINGE) {
                    axis_angle_quat(ql, M::jnt_axis[j], q);
                    quat_mul(quat, quat, ql);
                    quat_to_mat(R, quat);
                    rot_vec(t, R, M::jnt_pos[j]);
                    pos[0] = d.xanchor[j][0] - t[0], pos[1] = d.xanchor[j][1] - t[1], pos[2] = d.xanchor[j][2] - t[2];
                } else {
                    pos[0] += d.xaxis[j][0] * q, pos[1] += d.xaxis[j][1] * q, pos[2] += d.xaxis[j][2] * q;
                }
            }
        }
        quat_normalize(quat);
#pragma unroll
        for (int k = 0; k < 3; k++) d.xpos[b][k] = pos[k];
#pragma unroll
        for (int k = 0; k < 4; k++) d.xquat[b][k] = quat[k];
        quat_to_mat(d.xmat[b], quat);
        double t[3];
        rot_vec(t, d.xmat[b], M::body_ipos[b]);
        d.xipos[b][0] = pos[0] + t[0], d.xipos[b][1] = pos[1] + t[1], d.xipos[b][2] = pos[2] + t[2];
    }
}

template <class M>
MJX_DEV void com_pos(Data<M> &d) {
    double mass = 0, c[3] = {0, 0, 0};
#pragma unroll
    for (int b = 1; b < M::NBODY; b++) {
        mass += M::body_mass[b];
        c[0] += M::body_mass[b] * d.xipos[b][0], c[1] += M::body_mass[b] * d.xipos[b][1], c[2] += M::body_mass[b] * d.xipos[b][2];
    }
    d.com[0] = c[0] / mass, d.com[1] = c[1] / mass, d.com[2] = c[2] / mass;
#pragma unroll
    for (int k = 0; k < 10; k++) d.cinert[0][k] = 0;
#pragma unroll
    for (int b = 1; b < M::NBODY; b++) {
        const double *R = d.xmat[b], *I = M::body_inertia[b];
        double off[3] = {d.xipos[b][0] - d.com[0], d.xipos[b][1] - d.com[1], d.xipos[b][2] - d.com[2]};
        double T[9], W[9];  // T = R I, W = T R^T
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) T[3 * i + j] = R[3 * i] * I[j] + R[3 * i + 1] * I[3 + j] + R[3 * i + 2] * I[6 + j];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) W[3 * i + j] = T[3 * i] * R[3 * j] + T[3 * i + 1] * R[3 * j + 1] + T[3 * i + 2] * R[3 * j + 2];
        const double mm = M::body_mass[b], dd = dot3(off, off);
        double *ci = d.cinert[b];
        ci[0] = W[0] + mm * (dd - off[0] * off[0]), ci[1] = W[4] + mm * (dd - off[1] * off[1]), ci[2] = W[8] + mm * (dd - off[2] * off[2]);
        ci[3] = W[1] - mm * off[0] * off[1], ci[4] = W[2] - mm * off[0] * off[2], ci[5] = W[5] - mm * off[1] * off[2];
        ci[6] = mm * off[0], ci[7] = mm * off[1], ci[8] = mm * off[2], ci[9] = mm;
    }
#pragma unroll
    for (int j = 0; j < M::NJNT; j++) {
        const int b = M::jnt_bodyid[j], a = M::jnt_dofadr[j];
        double off[3] = {d.com[0] - d.xanchor[j][0], d.com[1] - d.xanchor[j][1], d.com[2] - d.xanchor[j][2]};
        if (M::jnt_type[j] == FREE) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
#pragma unroll
                for (int c6 = 0; c6 < 6; c6++) d.cdof[a + k][c6] = 0;
                d.cdof[a + k][3 + k] = 1.0;
                double ax[3] = {d.xmat[b][k], d.xmat[b][3 + k], d.xmat[b][6 + k]};
                d.cdof[a + 3 + k][0] = ax[0], d.cdof[a + 3 + k][1] = ax[1], d.cdof[a + 3 + k][2] = ax[2];
                cross3(d.cdof[a + 3 + k] + 3, ax, off);
            }
        } else if (M::jnt_type[j] == HINGE) {
            d.cdof[a][0] = d.xaxis[j][0], d.cdof[a][1] = d.xaxis[j][1], d.cdof[a][2] = d.xaxis[j][2];
            cross3(d.cdof[a] + 3, d.xaxis[j], off);
        } else {
            d.cdof[a][0] = d.cdof[a][1] = d.cdof[a][2] = 0;
            d.cdof[a][3] = d.xaxis[j][0], d.cdof[a][4] = d.xaxis[j][1], d.cdof[a][5] = d.xaxis[j][2];
        }
    }
}

// velocity stage + bias forces in one sweep: cvel, and the recursive Newton-Euler pass with qacc = 0.
// cdof_dot is formed on the fly (it is only ever used multiplied by qvel).
template <class M>
MJX_DEV void com_vel_and_bias(Data<M> &d, double *qfrc_bias) {
    double cacc[M::NBODY][6], cfrc[M::NBODY][6];
#pragma unroll
    for (int k = 0; k < 6; k++) d.cvel[0][k] = 0, cacc[0][k] = 0, cfrc[0][k] = 0;
    cacc[0][3] = -M::gravity[0], cacc[0][4] = -M::gravity[1], cacc[0][5] = -M::gravity[2];
#pragma unroll
    for (int b = 1; b < M::NBODY; b++) {
        const int p = M::body_parentid[b];
        double v[6], a[6];
#pragma unroll
        for (int k = 0; k < 6; k++) v[k] = d.cvel[p][k], a[k] = cacc[p][k];
#pragma unroll
        for (int j = M::body_jntadr[b]; j < M::body_jntadr[b] + M::body_jntnum[b]; j++) {
            const int da = M::jnt_dofadr[j];
            if (M::jnt_type[j] == FREE) {
#pragma unroll
                for (int k = 0; k < 3; k++)
#pragma unroll
                    for (int c = 0; c < 6; c++) v[c] += d.cdof[da + k][c] * d.qvel[da + k];
                double dd[3][6];
#pragma unroll
                for (int k = 0; k < 3; k++) cross_motion(dd[k], v, d.cdof[da + 3 + k]);
#pragma unroll
                for (int k = 0; k < 3; k++)
#pragma unroll
                    for (int c = 0; c < 6; c++) v[c] += d.cdof[da + 3 + k][c] * d.qvel[da + 3 + k], a[c] += dd[k][c] * d.qvel[da + 3 + k];
            } else {
                double dd[6];
                cross_motion(dd, v, d.cdof[da]);
#pragma unroll
                for (int c = 0; c < 6; c++) v[c] += d.cdof[da][c] * d.qvel[da], a[c] += dd[c] * d.qvel[da];
            }
        }
        double Ia[6], Iv[6], x[6];
        inert_mul(Ia, d.cinert[b], a), inert_mul(Iv, d.cinert[b], v), cross_force(x, v, Iv);
#pragma unroll
        for (int k = 0; k < 6; k++) d.cvel[b][k] = v[k], cacc[b][k] = a[k], cfrc[b][k] = Ia[k] + x[k];
    }
#pragma unroll
    for (int b = M::NBODY - 1; b > 0; b--) {
        const int p = M::body_parentid[b];
        if (p > 0)
#pragma unroll
            for (int k = 0; k < 6; k++) cfrc[p][k] += cfrc[b][k];
    }
#pragma unroll
    for (int i = 0; i < M::NV; i++) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) s += d.cdof[i][k] * cfrc[M::dof_bodyid[i]][k];
        qfrc_bias[i] = s;
    }
}

// composite rigid body: accumulates the subtree inertias IN PLACE (cinert is not needed per body afterwards) and fills the
// packed mass matrix
template <class M>
MJX_DEV void crb(Data<M> &d) {
#pragma unroll
    for (int b = M::NBODY - 1; b > 0; b--) {
        const int p = M::body_parentid[b];
        if (p > 0)
#pragma unroll
            for (int k = 0; k < 10; k++) d.cinert[p][k] += d.cinert[b][k];
    }
#pragma unroll
    for (int k = 0; k < Data<M>::NTRI; k++) d.qM[k] = 0;
#pragma unroll
    for (int i = 0; i < M::NV; i++) {
        double buf[6];
        inert_mul(buf, d.cinert[M::dof_bodyid[i]], d.cdof[i]);
#pragma unroll
        for (int j = i; j >= 0; j = M::dof_parentid[j]) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 6; k++) s += d.cdof[j][k] * buf[k];
            d.qM[tri(i, j)] = s;
        }
        d.qM[tri(i, i)] += M::dof_armature[i];
    }
}

// packed Cholesky A = L L^T and solve; loops are rolled (NV^3/6 work) to keep code size in check
template <int N>
MJX_DEV void chol_factor(const double *A, double *L) {
    for (int i = 0; i < N; i++)
        for (int j = 0; j <= i; j++) {
            double s = A[tri(i, j)];
            for (int k = 0; k < j; k++) s -= L[tri(i, k)] * L[tri(j, k)];
            if (i == j)
                L[tri(i, i)] = sqrt(s < kMinVal ? kMinVal : s);
            else
                L[tri(i, j)] = s / L[tri(j, j)];
        }
}
template <int N>
MJX_DEV void chol_solve(const double *L, double *x) {
    for (int i = 0; i < N; i++) {
        double s = x[i];
        for (int k = 0; k < i; k++) s -= L[tri(i, k)] * x[k];
        x[i] = s / L[tri(i, i)];
    }
    for (int i = N - 1; i >= 0; i--) {
        double s = x[i];
        for (int k = i + 1; k < N; k++) s -= L[tri(k, i)] * x[k];
        x[i] = s / L[tri(i, i)];
    }
}
template <int N>
MJX_DEV void sym_mul(const double *A, const double *x, double *y) {  // y = A x, A packed symmetric
    for (int i = 0; i < N; i++) {
        double s = 0;
        for (int k = 0; k < N; k++) s += A[k <= i ? tri(i, k) : tri(k, i)] * x[k];
        y[i] = s;
    }
}

// ---- collision ----------------------------------------------------------------------------------------------------
MJX_DEV void make_frame(double *f) {
    normalize3(f);
    if (sqrt(dot3(f + 3, f + 3)) < 0.5) {
        f[3] = f[4] = f[5] = 0;
        if (f[1] < 0.5 && f[1] > -0.5)
            f[4] = 1;
        else
            f[5] = 1;
    }
    const double t = dot3(f, f + 3);
    f[3] -= t * f[0], f[4] -= t * f[1], f[5] -= t * f[2];
    normalize3(f + 3);
    cross3(f + 6, f, f + 3);
}

template <class M>
MJX_DEV void add_contact(Data<M> &d, int pair, double dist, const double *pos, const double *n, const double *tangent, bool flip) {
    if (!(dist < M::pair_margin[pair]) || d.ncon >= Data<M>::MAXCON) return;
    Contact<M> &c = d.con[d.ncon++];
    c.dist = dist, c.pair = pair;
    const double sg = flip ? -1.0 : 1.0;
#pragma unroll
    for (int k = 0; k < 3; k++) c.pos[k] = pos[k], c.frame[k] = sg * n[k], c.frame[3 + k] = (tangent && !flip) ? tangent[k] : 0.0;
    make_frame(c.frame);
}
template <class M>
MJX_DEV void sphere_pair(Data<M> &d, int pair, const double *p1, double r1, const double *p2, double r2, bool flip) {
    double n[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
    const double dist = sqrt(dot3(n, n));
    if (dist < kMinVal)
        n[0] = 1, n[1] = n[2] = 0;
    else
        n[0] /= dist, n[1] /= dist, n[2] /= dist;
    double pos[3];
    const double mid = r1 + 0.5 * (dist - r1 - r2);
#pragma unroll
    for (int k = 0; k < 3; k++) pos[k] = p1[k] + n[k] * mid;
    add_contact<M>(d, pair, dist - r1 - r2, pos, n, nullptr, flip);
}

// Signed distance from p to the solid cylinder (centre c, unit axis u, radius R, half height H) and its gradient: the same
// case analysis as oracle/mujoco_core.c cylinder_sd().
MJX_DEV double cylinder_sd(const double *p, const double *c, const double *u, double R, double H, double *grad) {
    const double dv[3] = {p[0] - c[0], p[1] - c[1], p[2] - c[2]};
    const double a = dot3(dv, u);
    double rv[3] = {dv[0] - a * u[0], dv[1] - a * u[1], dv[2] - a * u[2]};
    const double rho = sqrt(dot3(rv, rv));
    if (rho > kMinVal) {
        rv[0] /= rho, rv[1] /= rho, rv[2] /= rho;
    } else {
        const double e[3] = {fabs(u[0]) < 0.5 ? 1.0 : 0.0, fabs(u[0]) < 0.5 ? 0.0 : 1.0, 0.0};
        const double t = dot3(e, u);
        rv[0] = e[0] - t * u[0], rv[1] = e[1] - t * u[1], rv[2] = e[2] - t * u[2];
        normalize3(rv);
    }
    const double sa = a >= 0 ? 1.0 : -1.0, ea = fabs(a) - H, er = rho - R;
    const bool cap = (ea <= 0 && er <= 0) ? (ea > er) : (er <= 0);   // nearest feature is a cap
    const bool wall = (ea <= 0 && er <= 0) ? !(ea > er) : (ea <= 0);  // ... or the wall; neither: the rim
    if (cap) {
        grad[0] = sa * u[0], grad[1] = sa * u[1], grad[2] = sa * u[2];
        return ea;
    }
    if (wall) {
        grad[0] = rv[0], grad[1] = rv[1], grad[2] = rv[2];
        return er;
    }
    const double sd = sqrt(ea * ea + er * er);
#pragma unroll
    for (int k = 0; k < 3; k++) grad[k] = (ea * sa * u[k] + er * rv[k]) / sd;
    return sd;
}

// Capsule against cylinder: minimise the (convex) signed distance of the segment point P(t) to the cylinder by bisection on the sign
// of grad . axis (oracle/mujoco_core.c capsule_cylinder()).
template <class M>
MJX_DEV void capsule_cylinder(Data<M> &d, int pair, const double *pc, const double *zc, double hc, double rc, const double *py, const double *zy,
                              double R, double H, bool flip) {
    const double cc[3] = {pc[0] - py[0], pc[1] - py[1], pc[2] - py[2]};
    if (sqrt(dot3(cc, cc)) > hc + rc + sqrt(R * R + H * H) + M::pair_margin[pair]) return;
    double lo = -hc, hi = hc, p[3], g[3], glo[3], ghi[3], t;
    p[0] = pc[0] + lo * zc[0], p[1] = pc[1] + lo * zc[1], p[2] = pc[2] + lo * zc[2];
    cylinder_sd(p, py, zy, R, H, glo);
    if (dot3(glo, zc) >= 0) {
        t = lo;
        g[0] = glo[0], g[1] = glo[1], g[2] = glo[2];
    } else {
        p[0] = pc[0] + hi * zc[0], p[1] = pc[1] + hi * zc[1], p[2] = pc[2] + hi * zc[2];
        cylinder_sd(p, py, zy, R, H, ghi);
        if (dot3(ghi, zc) <= 0) {
            t = hi;
            g[0] = ghi[0], g[1] = ghi[1], g[2] = ghi[2];
        } else {
            for (int it = 0; it < 60; it++) {
                const double mid = 0.5 * (lo + hi);
                p[0] = pc[0] + mid * zc[0], p[1] = pc[1] + mid * zc[1], p[2] = pc[2] + mid * zc[2];
                cylinder_sd(p, py, zy, R, H, g);
                const bool left = dot3(g, zc) < 0;
                lo = left ? mid : lo, hi = left ? hi : mid;
#pragma unroll
                for (int k = 0; k < 3; k++) glo[k] = left ? g[k] : glo[k], ghi[k] = left ? ghi[k] : g[k];
            }
            t = 0.5 * (lo + hi);
            // on a kink of the distance: the subgradient element that is stationary along the segment (see the oracle)
            const double a = dot3(glo, zc), b = dot3(ghi, zc), w = b / (b - a);
#pragma unroll
            for (int k = 0; k < 3; k++) g[k] = w * glo[k] + (1.0 - w) * ghi[k];
            normalize3(g);
        }
    }
    p[0] = pc[0] + t * zc[0], p[1] = pc[1] + t * zc[1], p[2] = pc[2] + t * zc[2];
    double gt[3];
    const double sd = cylinder_sd(p, py, zy, R, H, gt), dist = sd - rc;
    double n[3], pos[3];
#pragma unroll
    for (int k = 0; k < 3; k++) n[k] = -g[k], pos[k] = p[k] - g[k] * (rc + 0.5 * dist);
    add_contact<M>(d, pair, dist, pos, n, nullptr, flip);
}

template <class M>
MJX_DEV void geom_pose(const Data<M> &d, int g, double *pos, double *axis_z) {
    const int b = M::geom_bodyid[g];
    double t[3];
    rot_vec(t, d.xmat[b], M::geom_pos[g]);
    pos[0] = d.xpos[b][0] + t[0], pos[1] = d.xpos[b][1] + t[1], pos[2] = d.xpos[b][2] + t[2];
    const double lz[3] = {M::geom_mat[g][2], M::geom_mat[g][5], M::geom_mat[g][8]};  // geom z axis in the body frame
    rot_vec(axis_z, d.xmat[b], lz);
}

template <class M>
MJX_DEV void collision(Data<M> &d) {
    d.ncon = 0;
    for (int p = 0; p < M::NPAIR; p++) {
        int g1 = M::pair_geom1[p], g2 = M::pair_geom2[p];
        bool flip = false;
        if (M::geom_type[g1] > M::geom_type[g2]) {
            const int t = g1;
            g1 = g2, g2 = t, flip = true;
        }
        const int t1 = M::geom_type[g1], t2 = M::geom_type[g2];
        double p1[3], z1[3], p2[3], z2[3];
        geom_pose<M>(d, g1, p1, z1), geom_pose<M>(d, g2, p2, z2);
        const double r1 = M::geom_size[g1][0], r2 = M::geom_size[g2][0], h1 = M::geom_size[g1][1], h2 = M::geom_size[g2][1];
        if (t1 == PLANE) {
            if (t2 == SPHERE) {
                double v[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, pos[3];
                const double dist = dot3(v, z1) - r2;
#pragma unroll
                for (int k = 0; k < 3; k++) pos[k] = p2[k] - z1[k] * (r2 + 0.5 * dist);
                add_contact<M>(d, p, dist, pos, z1, nullptr, false);
            } else {
                for (int s = 1; s >= -1; s -= 2) {
                    double c[3], v[3], pos[3];
#pragma unroll
                    for (int k = 0; k < 3; k++) c[k] = p2[k] + s * h2 * z2[k], v[k] = c[k] - p1[k];
                    const double dist = dot3(v, z1) - r2;
#pragma unroll
                    for (int k = 0; k < 3; k++) pos[k] = c[k] - z1[k] * (r2 + 0.5 * dist);
                    add_contact<M>(d, p, dist, pos, z1, z2, false);
                }
            }
        } else if (t1 == SPHERE && t2 == SPHERE) {
            sphere_pair<M>(d, p, p1, r1, p2, r2, flip);
        } else if (t1 == SPHERE && t2 == CAPSULE) {
            double v[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
            double x = dot3(v, z2);
            x = x > h2 ? h2 : (x < -h2 ? -h2 : x);
            double c[3] = {p2[0] + x * z2[0], p2[1] + x * z2[1], p2[2] + x * z2[2]};
            sphere_pair<M>(d, p, p1, r1, c, r2, flip);
        } else if (t1 == CAPSULE && t2 == CAPSULE) {
            double dif[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
            const double mb = -dot3(z1, z2), u = -dot3(z1, dif), v = dot3(z2, dif), det = 1.0 - mb * mb;
            double x1, x2;
            if (fabs(det) >= 1e-12) {
                x1 = (u - mb * v) / det, x2 = (v - mb * u) / det;
                if (x1 > h1)
                    x1 = h1, x2 = v - mb * h1;
                else if (x1 < -h1)
                    x1 = -h1, x2 = v + mb * h1;
                if (x2 > h2) {
                    x2 = h2, x1 = u - mb * h2;
                    x1 = x1 > h1 ? h1 : (x1 < -h1 ? -h1 : x1);
                } else if (x2 < -h2) {
                    x2 = -h2, x1 = u + mb * h2;
                    x1 = x1 > h1 ? h1 : (x1 < -h1 ? -h1 : x1);
                }
            } else {
                x2 = v;
                x2 = x2 > h2 ? h2 : (x2 < -h2 ? -h2 : x2);
                x1 = u - mb * x2;
                x1 = x1 > h1 ? h1 : (x1 < -h1 ? -h1 : x1);
            }
            double c1[3], c2[3];
#pragma unroll
            for (int k = 0; k < 3; k++) c1[k] = p1[k] + x1 * z1[k], c2[k] = p2[k] + x2 * z2[k];
            sphere_pair<M>(d, p, c1, r1, c2, r2, flip);
        } else if (t1 == CAPSULE && t2 == CYLINDER) {
            capsule_cylinder<M>(d, p, p1, z1, h1, r1, p2, z2, r2, h2, flip);
        }
    }
}

// ---- constraints --------------------------------------------------------------------------------------------------
// POW2: every solimp of the model has power == 2 (the MuJoCo default), so pow() -- ~1000 instructions per inlined call -- is
// not even compiled in
template <bool POW2 = false>
MJX_DEV double impedance(const double *solimp, double pos, double margin) {
    double dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
    dmin = dmin < kMinImp ? kMinImp : (dmin > kMaxImp ? kMaxImp : dmin);
    dmax = dmax < kMinImp ? kMinImp : (dmax > kMaxImp ? kMaxImp : dmax);
    width = width < kMinVal ? kMinVal : width;
    mid = mid < kMinImp ? kMinImp : (mid > kMaxImp ? kMaxImp : mid);
    power = power < 1 ? 1 : power;
    const double x = fabs(pos - margin) / width;
    double y;
    if (x >= 1)
        y = 1;
    else if (x <= 0)
        y = 0;
    else if (power == 1)
        y = x;
    else if (POW2 || power == 2)  // x^2 / mid and 1 - (1 - x)^2 / (1 - mid) without pow()
        y = x <= mid ? x * x / mid : 1 - (1 - x) * (1 - x) / (1 - mid);
    else if (x <= mid)
        y = pow(x, power) / pow(mid, power - 1);
    else
        y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
    return dmin + y * (dmax - dmin);
}
template <class M>
constexpr bool all_power_two() {
    for (int j = 0; j < M::NJNT; j++)
        if (M::jnt_solimp[j][4] != 2.0) return false;
    for (int p = 0; p < M::NPAIR; p++)
        if (M::pair_solimp[p][4] != 2.0) return false;
    return true;
}
// stiffness k, damping b, impedance and regulariser R of one row
template <class M>
MJX_DEV void row_params(const double *solref, const double *solimp, double pos, double margin, double diag_approx, double &k,
                        double &b, double &imp, double &R) {
    double timeconst = solref[0];
    const double dampratio = solref[1];
    const double dmax = solimp[1] < kMinImp ? kMinImp : (solimp[1] > kMaxImp ? kMaxImp : solimp[1]);
    if (timeconst < 2 * M::TIMESTEP) timeconst = 2 * M::TIMESTEP;
    k = 1.0 / (dmax * dmax * timeconst * timeconst * dampratio * dampratio), b = 2.0 / (dmax * timeconst);
    imp = impedance<all_power_two<M>()>(solimp, pos, margin);
    R = (1 - imp) * diag_approx / imp;
    if (R < kMinVal) R = kMinVal;
}

// Jacobian of the contact point: rows = contact-frame axes, columns = dofs; velocity of (body2 - body1) at the point
template <class M>
MJX_DEV void contact_jacobian(const Data<M> &d, const Contact<M> &c, double Jc[3][M::NV]) {
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int i = 0; i < M::NV; i++) Jc[a][i] = 0;
    const double r[3] = {c.pos[0] - d.com[0], c.pos[1] - d.com[1], c.pos[2] - d.com[2]};
    const int bodies[2] = {M::geom_bodyid[M::pair_geom1[c.pair]], M::geom_bodyid[M::pair_geom2[c.pair]]};
    for (int s = 0; s < 2; s++) {
        int b = bodies[s];
        while (b > 0 && M::body_dofnum[b] == 0) b = M::body_parentid[b];
        if (b <= 0) continue;
        const double sg = s == 0 ? -1.0 : 1.0;
        for (int i = M::body_dofadr[b] + M::body_dofnum[b] - 1; i >= 0; i = M::dof_parentid[i]) {
            double t[3];
            cross3(t, d.cdof[i], r);
            const double v[3] = {d.cdof[i][3] + t[0], d.cdof[i][4] + t[1], d.cdof[i][5] + t[2]};
#pragma unroll
            for (int a = 0; a < 3; a++) Jc[a][i] += sg * dot3(c.frame + 3 * a, v);
        }
    }
}

// limits + per-contact parameters (D, aref); needs qvel for the reference acceleration
template <class M>
MJX_DEV void make_constraint(Data<M> &d) {
    d.nlimit = 0;
#pragma unroll
    for (int j = 0; j < M::NJNT; j++) {
        if (!M::jnt_limited[j] || (M::jnt_type[j] != HINGE && M::jnt_type[j] != SLIDE)) continue;
        const double value = d.qpos[M::jnt_qposadr[j]];
        for (int side = -1; side <= 1; side += 2) {
            const double dist = side * (M::jnt_range[j][side < 0 ? 0 : 1] - value);
            if (dist < M::jnt_margin[j] && d.nlimit < M::NJNT) {
                const int n = d.nlimit++, dof = M::jnt_dofadr[j];
                double k, b, imp, R;
                row_params<M>(M::jnt_solref[j], M::jnt_solimp[j], dist, M::jnt_margin[j], M::dof_invweight0[dof], k, b, imp, R);
                d.lim_dof[n] = dof, d.lim_sign[n] = -side, d.lim_D[n] = 1.0 / R;
                d.lim_aref[n] = -b * (-side * d.qvel[dof]) - k * imp * (dist - M::jnt_margin[j]);
            }
        }
    }
    for (int c = 0; c < d.ncon; c++) {
        const Contact<M> &con = d.con[c];
        const int p = con.pair;
        const int b1 = M::geom_bodyid[M::pair_geom1[p]], b2 = M::geom_bodyid[M::pair_geom2[p]];
        const double tran = M::body_invweight0[b1][0] + M::body_invweight0[b2][0], mu = M::pair_friction[p];
        double k, b, imp, R;
        const bool pyramid = M::pair_condim[p] > 1;
        row_params<M>(M::pair_solref[p], M::pair_solimp[p], con.dist, M::pair_margin[p], pyramid ? tran + mu * mu * tran : tran, k, b, imp, R);
        if (pyramid) {
            R = 2 * mu * mu * R;
            if (R < kMinVal) R = kMinVal;
        }
        d.con_D[c] = 1.0 / R;
        // aref = -b * (J qvel) - k * imp * (pos - margin); the normal-row velocity for edge e is added per edge in the solver
        d.con_aref[c] = -k * imp * (con.dist - M::pair_margin[p]);
        d.con_force[c][0] = b;  // stash the damping coefficient until the solver forms the per-edge reference
    }
}

// ---- primal Newton solver ---------------------------------------------------------------------------------------------
// cost(x) = 1/2 (x - x_s)' M (x - x_s) + sum_rows 1/2 D min(0, J x - aref)^2
template <class M>
struct RowView {  // all rows, regenerated on the fly
    template <class F>
    static MJX_DEV void for_each(const Data<M> &d, const double *edge_aref, F &&f) {
        // f(row_index, D, aref, Jrow[NV] dense)
        double J[M::NV];
        for (int n = 0; n < d.nlimit; n++) {
#pragma unroll
            for (int i = 0; i < M::NV; i++) J[i] = 0;
            J[d.lim_dof[n]] = d.lim_sign[n];
            f(n, d.lim_D[n], d.lim_aref[n], J);
        }
        int row = d.nlimit;
        for (int c = 0; c < d.ncon; c++) {
            double Jc[3][M::NV];
            contact_jacobian<M>(d, d.con[c], Jc);
            const int p = d.con[c].pair;
            if (M::pair_condim[p] == 1) {
                f(row, d.con_D[c], edge_aref[row], Jc[0]);
                row++;
            } else {
                const double mu = M::pair_friction[p];
                for (int e = 0; e < 4; e++) {
                    const double sg = (e & 1) ? -mu : mu;
                    const int t = 1 + e / 2;
#pragma unroll
                    for (int i = 0; i < M::NV; i++) J[i] = Jc[0][i] + sg * Jc[t][i];
                    f(row, d.con_D[c], edge_aref[row], J);
                    row++;
                }
            }
        }
    }
};

template <class M>
MJX_DEV int solve_newton(Data<M> &d) {
    constexpr int NV = M::NV, MAXROW = M::NJNT + 4 * Data<M>::MAXCON;
    double aref[MAXROW], jar[MAXROW], jd[MAXROW], Dv[MAXROW];
    // per-row reference accelerations (needs J qvel per edge)
    {
        RowView<M>::for_each(d, aref, [&](int r, double D, double, const double *J) {
            Dv[r] = D;
            if (r < d.nlimit) {
                aref[r] = d.lim_aref[r];
            } else {
                double v = 0;
#pragma unroll
                for (int i = 0; i < NV; i++) v += J[i] * d.qvel[i];
                jar[r] = v;  // stash
            }
        });
        int row = d.nlimit;
        for (int c = 0; c < d.ncon; c++) {
            const int rows = M::pair_condim[d.con[c].pair] == 1 ? 1 : 4;
            for (int e = 0; e < rows; e++, row++) aref[row] = -d.con_force[c][0] * jar[row] + d.con_aref[c];
        }
    }
    const int nrow = d.nlimit + [&] {
        int n = 0;
        for (int c = 0; c < d.ncon; c++) n += M::pair_condim[d.con[c].pair] == 1 ? 1 : 4;
        return n;
    }();
    double x[NV], grad[NV], dir[NV], H[Data<M>::NTRI], HL[Data<M>::NTRI], dx[NV], Mdx[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) x[i] = d.qacc_warm[i];  // warm start; M (x - x_smooth) = M x - qfrc_smooth needs no x_smooth
    const double scale = 1.0 / (M::MEANINERTIA * (NV > 1 ? NV : 1));
    int it = 0;
    for (; it < 50; it++) {
        sym_mul<NV>(d.qM, x, dx);
#pragma unroll
        for (int i = 0; i < NV; i++) Mdx[i] = dx[i] - d.qfrc_smooth[i];
#pragma unroll
        for (int i = 0; i < NV; i++) grad[i] = Mdx[i];
        for (int k = 0; k < Data<M>::NTRI; k++) H[k] = d.qM[k];
        RowView<M>::for_each(d, aref, [&](int r, double D, double ar, const double *J) {
            double v = -ar;
#pragma unroll
            for (int i = 0; i < NV; i++) v += J[i] * x[i];
            jar[r] = v;
            if (v < 0) {
                const double Dv_ = D * v;
                for (int i = 0; i < NV; i++) {
                    if (J[i] == 0.0) continue;
                    grad[i] += J[i] * Dv_;
                    const double DJ = D * J[i];
                    for (int k = 0; k <= i; k++) H[tri(i, k)] += DJ * J[k];
                }
            }
        });
        double gn = 0;
#pragma unroll
        for (int i = 0; i < NV; i++) gn += grad[i] * grad[i];
        if (sqrt(gn) * scale < 1e-10) break;
        chol_factor<NV>(H, HL);
#pragma unroll
        for (int i = 0; i < NV; i++) dir[i] = -grad[i];
        chol_solve<NV>(HL, dir);
        // line search: phi'(alpha) = g0 + alpha h0 + sum_active D (jar + alpha jd) jd
        double Md[NV], g0 = 0, h0 = 0;
        sym_mul<NV>(d.qM, dir, Md);
#pragma unroll
        for (int i = 0; i < NV; i++) h0 += dir[i] * Md[i], g0 += dir[i] * Mdx[i];
        RowView<M>::for_each(d, aref, [&](int r, double, double, const double *J) {
            double v = 0;
#pragma unroll
            for (int i = 0; i < NV; i++) v += J[i] * dir[i];
            jd[r] = v;
        });
        double alpha = 0, lo = 0, hi = INFINITY;
        for (int ls = 0; ls < 40; ls++) {
            double g = g0 + alpha * h0, h = h0;
            for (int r = 0; r < nrow; r++) {
                const double v = jar[r] + alpha * jd[r];
                if (v < 0) g += Dv[r] * v * jd[r], h += Dv[r] * jd[r] * jd[r];
            }
            if (fabs(g) <= 1e-14 * (fabs(g0) + 1e-300)) break;
            if (g < 0)
                lo = alpha;
            else
                hi = alpha;
            double next = alpha - g / h;
            if (!(next > lo && next < hi)) next = hi < INFINITY ? 0.5 * (lo + hi) : 2 * alpha + 1.0;  // safeguard
            if (next == alpha) break;
            alpha = next;
        }
        if (!(alpha > 0)) break;
        double move = 0;
#pragma unroll
        for (int i = 0; i < NV; i++) x[i] += alpha * dir[i], move += fabs(alpha * dir[i]);
        if (move * scale < 1e-16) break;
    }
    // forces at the solution
#pragma unroll
    for (int i = 0; i < NV; i++) d.qacc[i] = x[i], d.qfrc_constraint[i] = 0;
    RowView<M>::for_each(d, aref, [&](int r, double D, double ar, const double *J) {
        double v = -ar;
#pragma unroll
        for (int i = 0; i < NV; i++) v += J[i] * x[i];
        const double f = v < 0 ? -D * v : 0.0;
        jar[r] = f;
        if (f != 0.0)
#pragma unroll
            for (int i = 0; i < NV; i++) d.qfrc_constraint[i] += J[i] * f;
    });
    for (int n = 0; n < d.nlimit; n++) d.lim_force[n] = jar[n];
    int row = d.nlimit;
    for (int c = 0; c < d.ncon; c++) {
        const int rows = M::pair_condim[d.con[c].pair] == 1 ? 1 : 4;
        for (int e = 0; e < 4; e++) d.con_force[c][e] = e < rows ? jar[row + e] : 0.0;
        row += rows;
    }
    return it;
}

// ---- dual projected Gauss-Seidel (humanoid.xml:8 `solver="PGS" iterations="50"`) ----------------------------------------------
// The same rows as solve_newton, solved the way mj_solPGS does: forces f >= 0 of the unilateral rows (limits, frictionless contacts,
// pyramid edges) are relaxed one at a time in row order, f_r <- max(0, f_r - res_r / AR_rr) with res = (J M^-1 J^T + R) f + b, for at most
// M::ITERATIONS sweeps or until the scaled cost improvement of a sweep drops below the tolerance (1e-8).  Matrix-free in acceleration
// space: a = qacc_smooth + M^-1 J^T f is carried along, so res_r = J_r a - aref_r + R_r f_r and a row update adds (M^-1 J_r^T) delta.
// Warm start (mj's dual warmstart): forces implied by qacc_warmstart, dropped for zero if their dual cost is positive.
// d.qL must hold the Cholesky factor of M and d.qacc_smooth the unconstrained acceleration.  One-lane cross-check of mjx_coop.h pgs().
template <class M>
MJX_DEV int solve_pgs(Data<M> &d) {
    constexpr int NV = M::NV, MAXROW = M::NJNT + 4 * Data<M>::MAXCON;
    double aref[MAXROW], jar[MAXROW], Rr[MAXROW], f[MAXROW], b[MAXROW], ARd[MAXROW];
    double MiJT[MAXROW][NV];
    RowView<M>::for_each(d, aref, [&](int r, double D, double, const double *J) {
        Rr[r] = 1.0 / D;
        if (r < d.nlimit) {
            aref[r] = d.lim_aref[r];
        } else {
            double v = 0;
#pragma unroll
            for (int i = 0; i < NV; i++) v += J[i] * d.qvel[i];
            jar[r] = v;  // stash J qvel
        }
    });
    int nrow = d.nlimit;
    for (int c = 0; c < d.ncon; c++) {
        const int rows = M::pair_condim[d.con[c].pair] == 1 ? 1 : 4;
        for (int e = 0; e < rows; e++, nrow++) aref[nrow] = -d.con_force[c][0] * jar[nrow] + d.con_aref[c];
    }
    double a[NV], qfrc[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) a[i] = 0, qfrc[i] = 0;
    double cost = 0;
    RowView<M>::for_each(d, aref, [&](int r, double D, double ar, const double *J) {
        double js = -ar, jw = -ar, diag = 0;
#pragma unroll
        for (int i = 0; i < NV; i++) MiJT[r][i] = J[i];
        chol_solve<NV>(d.qL, MiJT[r]);
#pragma unroll
        for (int i = 0; i < NV; i++) js += J[i] * d.qacc_smooth[i], jw += J[i] * d.qacc_warm[i], diag += J[i] * MiJT[r][i];
        b[r] = js, ARd[r] = diag + Rr[r];
        f[r] = jw < 0 ? -D * jw : 0.0;  // mj_constraintUpdate at qacc_warmstart
        cost += f[r] * (0.5 * Rr[r] * f[r] + b[r]);
#pragma unroll
        for (int i = 0; i < NV; i++) a[i] += MiJT[r][i] * f[r], qfrc[i] += J[i] * f[r];
    });
#pragma unroll
    for (int i = 0; i < NV; i++) cost += 0.5 * qfrc[i] * a[i];  // f' J M^-1 J' f
    if (cost > 0) {
        for (int r = 0; r < nrow; r++) f[r] = 0;
#pragma unroll
        for (int i = 0; i < NV; i++) a[i] = 0;
    }
#pragma unroll
    for (int i = 0; i < NV; i++) a[i] += d.qacc_smooth[i];
    const double scale = 1.0 / (M::MEANINERTIA * (NV > 1 ? NV : 1));
    int it = 0;
    for (; it < M::ITERATIONS;) {
        double improvement = 0;
        RowView<M>::for_each(d, aref, [&](int r, double, double ar, const double *J) {
            double res = Rr[r] * f[r] - ar;
#pragma unroll
            for (int i = 0; i < NV; i++) res += J[i] * a[i];
            double nw = f[r] - res / ARd[r];
            nw = nw < 0 ? 0.0 : nw;
            const double delta = nw - f[r];
            f[r] = nw;
            improvement -= delta * (0.5 * delta * ARd[r] + res);
#pragma unroll
            for (int i = 0; i < NV; i++) a[i] += MiJT[r][i] * delta;
        });
        it++;
        if (improvement * scale < 1e-8) break;
    }
#pragma unroll
    for (int i = 0; i < NV; i++) d.qacc[i] = a[i], d.qfrc_constraint[i] = 0;
    RowView<M>::for_each(d, aref, [&](int r, double, double, const double *J) {
        if (f[r] != 0.0)
#pragma unroll
            for (int i = 0; i < NV; i++) d.qfrc_constraint[i] += J[i] * f[r];
    });
    for (int n = 0; n < d.nlimit; n++) d.lim_force[n] = f[n];
    int row = d.nlimit;
    for (int c = 0; c < d.ncon; c++) {
        const int rows = M::pair_condim[d.con[c].pair] == 1 ? 1 : 4;
        for (int e = 0; e < 4; e++) d.con_force[c][e] = e < rows ? f[row + e] : 0.0;
        row += rows;
    }
    return it;
}

// ---- fluid forces of the medium (option density / viscosity), MuJoCo's inertia-box model ---------------------------------
// Every body is replaced by the box with its mass and principal moments (M::body_fluidbox, axes M::body_imat); in that frame,
// at the body's centre of mass, the velocity (w, v) gives  viscous: t -= pi d^3 mu w, f -= 3 pi d mu v  (d = mean edge) and
// drag: t_k -= rho b_k (b_i^4 + b_j^4) |w_k| w_k / 64, f_k -= rho b_i b_j |v_k| v_k / 2; the wrench is applied at the centre of
// mass.  Same arithmetic as oracle/mujoco_core.c fluid().  Adds into qfrc[NV].
template <class M>
MJX_DEV void fluid(const Data<M> &d, double *qfrc) {
    constexpr double PI = 3.14159265358979323846;
#pragma unroll
    for (int b = 1; b < M::NBODY; b++) {
        if (M::body_mass[b] < 1e-15) continue;
        double R[9];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++)
                R[3 * i + j] = d.xmat[b][3 * i] * M::body_imat[b][j] + d.xmat[b][3 * i + 1] * M::body_imat[b][3 + j] + d.xmat[b][3 * i + 2] * M::body_imat[b][6 + j];
        const double off[3] = {d.xipos[b][0] - d.com[0], d.xipos[b][1] - d.com[1], d.xipos[b][2] - d.com[2]};
        const double *w = d.cvel[b], *vl = d.cvel[b] + 3;
        const double vc[3] = {vl[0] + w[1] * off[2] - w[2] * off[1], vl[1] + w[2] * off[0] - w[0] * off[2], vl[2] + w[0] * off[1] - w[1] * off[0]};
        double lw[3], lv[3], lt[3], lf[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            lw[k] = R[k] * w[0] + R[3 + k] * w[1] + R[6 + k] * w[2];
            lv[k] = R[k] * vc[0] + R[3 + k] * vc[1] + R[6 + k] * vc[2];
            lt[k] = 0, lf[k] = 0;
        }
        const double bx0 = M::body_fluidbox[b][0], bx1 = M::body_fluidbox[b][1], bx2 = M::body_fluidbox[b][2];
        if (M::VISCOSITY > 0) {
            const double diam = (bx0 + bx1 + bx2) / 3.0;
#pragma unroll
            for (int k = 0; k < 3; k++) lt[k] += -PI * diam * diam * diam * M::VISCOSITY * lw[k], lf[k] += -3.0 * PI * diam * M::VISCOSITY * lv[k];
        }
        if (M::DENSITY > 0) {
            lf[0] -= 0.5 * M::DENSITY * bx1 * bx2 * fabs(lv[0]) * lv[0];
            lf[1] -= 0.5 * M::DENSITY * bx0 * bx2 * fabs(lv[1]) * lv[1];
            lf[2] -= 0.5 * M::DENSITY * bx0 * bx1 * fabs(lv[2]) * lv[2];
            lt[0] -= M::DENSITY * bx0 * (bx1 * bx1 * bx1 * bx1 + bx2 * bx2 * bx2 * bx2) * fabs(lw[0]) * lw[0] / 64.0;
            lt[1] -= M::DENSITY * bx1 * (bx0 * bx0 * bx0 * bx0 + bx2 * bx2 * bx2 * bx2) * fabs(lw[1]) * lw[1] / 64.0;
            lt[2] -= M::DENSITY * bx2 * (bx0 * bx0 * bx0 * bx0 + bx1 * bx1 * bx1 * bx1) * fabs(lw[2]) * lw[2] / 64.0;
        }
        double gt[3], gf[3];
#pragma unroll
        for (int k = 0; k < 3; k++)
            gt[k] = R[3 * k] * lt[0] + R[3 * k + 1] * lt[1] + R[3 * k + 2] * lt[2], gf[k] = R[3 * k] * lf[0] + R[3 * k + 1] * lf[1] + R[3 * k + 2] * lf[2];
        const double tq[3] = {gt[0] + off[1] * gf[2] - off[2] * gf[1], gt[1] + off[2] * gf[0] - off[0] * gf[2], gt[2] + off[0] * gf[1] - off[1] * gf[0]};
        for (int i = M::body_dofadr[b] + M::body_dofnum[b] - 1; i >= 0; i = M::dof_parentid[i])
            qfrc[i] += d.cdof[i][0] * tq[0] + d.cdof[i][1] * tq[1] + d.cdof[i][2] * tq[2] + d.cdof[i][3] * gf[0] + d.cdof[i][4] * gf[1] + d.cdof[i][5] * gf[2];
    }
}

// ---- forward dynamics -------------------------------------------------------------------------------------------------
// PGS: the reference's solver choice for this model (M::SOLVER, humanoid.xml:8); false = the primal Newton method for every model.
// With PGS the warm start follows MuJoCo's rule -- qacc_warmstart is what the previous mj_step left (saved after the integrator), the
// same for every RK4 stage -- so forward() does not touch d.qacc_warm; step() does.
template <class M, bool PGS = (M::SOLVER == 1)>
MJX_DEVN void forward(Data<M> &d) {
    constexpr int NV = M::NV;
    kinematics<M>(d);
    if constexpr (M::NTENDON > 0) tendons<M>(d.qpos, d.qvel, d.ten_length, d.ten_velocity);
    com_pos<M>(d);
    double bias[NV];
    com_vel_and_bias<M>(d, bias);  // uses the per-body cinert; crb() then turns cinert into composite inertias
    crb<M>(d);
    collision<M>(d);
    make_constraint<M>(d);
#pragma unroll
    for (int i = 0; i < NV; i++) d.qfrc_actuator[i] = 0;
#pragma unroll
    for (int u = 0; u < M::NU; u++) {
        double c = d.ctrl[u];
        c = c < M::actuator_ctrlrange[u][0] ? M::actuator_ctrlrange[u][0] : (c > M::actuator_ctrlrange[u][1] ? M::actuator_ctrlrange[u][1] : c);
        d.qfrc_actuator[M::actuator_dofadr[u]] += M::actuator_gear[u] * c;
    }
    double fl[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) fl[i] = 0;
    if constexpr (M::DENSITY > 0 || M::VISCOSITY > 0) fluid<M>(d, fl);
#pragma unroll
    for (int i = 0; i < NV; i++) {
        double passive = fl[i] - M::dof_damping[i] * d.qvel[i];
        const int j = M::dof_jntid[i];
        if (M::jnt_type[j] == HINGE || M::jnt_type[j] == SLIDE)
            passive -= M::jnt_stiffness[j] * (d.qpos[M::jnt_qposadr[j]] - M::qpos0[M::jnt_qposadr[j]]);
        d.qfrc_smooth[i] = passive - bias[i] + d.qfrc_actuator[i];
        d.qacc_smooth[i] = d.qfrc_smooth[i];
    }
    if (d.nlimit + d.ncon == 0) {  // unconstrained: one factorisation of M
        chol_factor<NV>(d.qM, d.qL);
        chol_solve<NV>(d.qL, d.qacc_smooth);
#pragma unroll
        for (int i = 0; i < NV; i++) d.qacc[i] = d.qacc_smooth[i], d.qfrc_constraint[i] = 0;
    } else if constexpr (PGS) {
        chol_factor<NV>(d.qM, d.qL);
        chol_solve<NV>(d.qL, d.qacc_smooth);
        solve_pgs<M>(d);
    } else {  // constrained: Newton from the warm start, M is never factorised by itself (see mjx_coop.h forward())
        solve_newton<M>(d);
    }
    if constexpr (!PGS) {
#pragma unroll
        for (int i = 0; i < NV; i++) d.qacc_warm[i] = d.qacc[i];
    }
}

template <class M>
MJX_DEV void integrate_pos(double *qpos, const double *qvel, double h) {
#pragma unroll
    for (int j = 0; j < M::NJNT; j++) {
        const int qa = M::jnt_qposadr[j], va = M::jnt_dofadr[j];
        if (M::jnt_type[j] == FREE) {
            qpos[qa] += h * qvel[va], qpos[qa + 1] += h * qvel[va + 1], qpos[qa + 2] += h * qvel[va + 2];
            double w[3] = {qvel[va + 3], qvel[va + 4], qvel[va + 5]}, qr[4];
            const double ang = h * normalize3(w);
            axis_angle_quat(qr, w, ang);
            quat_normalize(qpos + qa + 3);
            quat_mul(qpos + qa + 3, qpos + qa + 3, qr);
        } else {
            qpos[qa] += h * qvel[va];
        }
    }
}

// one mj_step: forward + integrator (semi-implicit Euler with implicit joint damping, or RK4)
template <class M, bool PGS = (M::SOLVER == 1)>
MJX_DEV void step(Data<M> &d) {
    constexpr int NV = M::NV, NQ = M::NQ;
    constexpr double h = M::TIMESTEP;
    forward<M, PGS>(d);
    if (M::INTEGRATOR == 0) {
        double qacc[NV];
        bool damped = false;
#pragma unroll
        for (int i = 0; i < NV; i++) damped |= M::dof_damping[i] > 0;
        if (damped) {
            double A[Data<M>::NTRI], L[Data<M>::NTRI];
            for (int k = 0; k < Data<M>::NTRI; k++) A[k] = d.qM[k];
#pragma unroll
            for (int i = 0; i < NV; i++) A[tri(i, i)] += h * M::dof_damping[i], qacc[i] = d.qfrc_smooth[i] + d.qfrc_constraint[i];
            chol_factor<NV>(A, L);
            chol_solve<NV>(L, qacc);
        } else {
#pragma unroll
            for (int i = 0; i < NV; i++) qacc[i] = d.qacc[i];
        }
#pragma unroll
        for (int i = 0; i < NV; i++) d.qvel[i] += h * qacc[i];
        integrate_pos<M>(d.qpos, d.qvel, h);
    } else {
        const double A[3][3] = {{0.5, 0, 0}, {0, 0.5, 0}, {0, 0, 1.0}}, B[4] = {1.0 / 6, 1.0 / 3, 1.0 / 3, 1.0 / 6};
        double q0[NQ], v0[NV], Fv[4][NV], Fa[4][NV], dv[NV], da[NV];
#pragma unroll
        for (int k = 0; k < NQ; k++) q0[k] = d.qpos[k];
#pragma unroll
        for (int k = 0; k < NV; k++) v0[k] = d.qvel[k], Fv[0][k] = d.qvel[k], Fa[0][k] = d.qacc[k];
        for (int i = 1; i < 4; i++) {
            for (int k = 0; k < NV; k++) {
                dv[k] = da[k] = 0;
                for (int j = 0; j < i; j++) dv[k] += A[i - 1][j] * Fv[j][k], da[k] += A[i - 1][j] * Fa[j][k];
            }
#pragma unroll
            for (int k = 0; k < NQ; k++) d.qpos[k] = q0[k];
            integrate_pos<M>(d.qpos, dv, h);
#pragma unroll
            for (int k = 0; k < NV; k++) d.qvel[k] = v0[k] + h * da[k];
            forward<M, PGS>(d);
#pragma unroll
            for (int k = 0; k < NV; k++) Fv[i][k] = d.qvel[k], Fa[i][k] = d.qacc[k];
        }
        for (int k = 0; k < NV; k++) {
            dv[k] = da[k] = 0;
            for (int j = 0; j < 4; j++) dv[k] += B[j] * Fv[j][k], da[k] += B[j] * Fa[j][k];
        }
#pragma unroll
        for (int k = 0; k < NQ; k++) d.qpos[k] = q0[k];
#pragma unroll
        for (int k = 0; k < NV; k++) d.qvel[k] = v0[k] + h * da[k];
        integrate_pos<M>(d.qpos, dv, h);
    }
    if constexpr (PGS) {  // mj_advance: "save qacc for next step warmstart" -- the last forward pass's qacc, once per step
#pragma unroll
        for (int i = 0; i < NV; i++) d.qacc_warm[i] = d.qacc[i];
    }
}

// mj_rnePostConstraint, the part the envs read: external (contact) forces per body, [torque; force] about the tree com
template <class M>
MJX_DEV void contact_forces(const Data<M> &d, double cfrc_ext[M::NBODY][6]) {
#pragma unroll
    for (int b = 0; b < M::NBODY; b++)
#pragma unroll
        for (int k = 0; k < 6; k++) cfrc_ext[b][k] = 0;
    for (int c = 0; c < d.ncon; c++) {
        const Contact<M> &con = d.con[c];
        const int p = con.pair;
        const double *f = d.con_force[c];
        double lf[3] = {f[0], 0, 0};
        if (M::pair_condim[p] > 1) {
            const double mu = M::pair_friction[p];
            lf[0] = f[0] + f[1] + f[2] + f[3], lf[1] = (f[0] - f[1]) * mu, lf[2] = (f[2] - f[3]) * mu;
        }
        double F[3], r[3] = {con.pos[0] - d.com[0], con.pos[1] - d.com[1], con.pos[2] - d.com[2]}, tq[3];
#pragma unroll
        for (int k = 0; k < 3; k++) F[k] = con.frame[k] * lf[0] + con.frame[3 + k] * lf[1] + con.frame[6 + k] * lf[2];
        cross3(tq, r, F);
        const int bodies[2] = {M::geom_bodyid[M::pair_geom1[p]], M::geom_bodyid[M::pair_geom2[p]]};
        for (int s = 0; s < 2; s++) {
            const int b = bodies[s];
            if (b <= 0) continue;
            const double sg = s == 0 ? -1.0 : 1.0;
            for (int k = 0; k < 3; k++) cfrc_ext[b][k] += sg * tq[k], cfrc_ext[b][3 + k] += sg * F[k];
        }
    }
}

}  // namespace mjx

#if !defined(MJX_HOST_EMU)
#pragma clang fp contract(off)
#endif
