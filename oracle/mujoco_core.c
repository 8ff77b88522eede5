/*
 * mujoco_core.c -- TEST INFRASTRUCTURE (CPU oracle), PARITY UNPINNED: see mujoco_core.h.
 *
 * Restatement of the MuJoCo forward-dynamics pipeline for hinge / slide / free joints, plane / sphere / capsule geoms,
 * joint limits, frictional (pyramidal) and frictionless contacts, motors, Euler and RK4 integration.  The names of the
 * stages follow MuJoCo's documented pipeline (mj_step = mj_forward + integrator):
 *
 *   position:     kinematics -> com_pos -> crb -> factor -> collision -> make_constraint
 *   velocity:     com_vel -> passive -> rne (bias) -> efc_vel
 *   actuation:    motors: force = gear * clamp(ctrl)
 *   acceleration: qacc_smooth = M^-1 (passive - bias + actuator)
 *   constraint:   min_qacc 1/2 |qacc - qacc_smooth|^2_M + sum_i s_i(J qacc - aref)   (Newton)  /  dual PGS
 *
 * Everything is dense and loop-based on purpose (clarity over speed): this is the checker, the product is the HIP engine.
 * Reference call sites that define what must be computed: gymnasium/envs/mujoco/mujoco_env.py:132-155,172-187.
 */
#include "mujoco_core.h"

#include <math.h>
#include <string.h>

#define MINVAL 1e-15
#define MINIMP 0.0001
#define MAXIMP 0.9999
/* PGS early termination: MuJoCo's default option tolerance (1e-8 on the scaled cost improvement of a sweep).  TEST KNOB: convergence studies lift it
 * together with the sweep cap (tests/test_mujoco_closed_forms.py: PGS must meet the Newton solution of the same convex problem). */
static double g_pgs_tolerance = 1e-8;
__attribute__((visibility("default"))) void orc_mj_set_pgs_tolerance(double tol) { g_pgs_tolerance = tol; }

/* ---- small vector helpers ----------------------------------------------------------------------------------- */
static double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void cross3(double *r, const double *a, const double *b) {
    double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    r[0] = x, r[1] = y, r[2] = z;
}
static double norm3(const double *a) { return sqrt(dot3(a, a)); }
static double normalize3(double *a) {
    double n = norm3(a);
    if (n < MINVAL) {
        a[0] = 1, a[1] = 0, a[2] = 0;
        return n;
    }
    a[0] /= n, a[1] /= n, a[2] /= n;
    return n;
}
static void mat_vec3(double *r, const double *m, const double *v) { /* r = M v, M row-major 3x3 */
    double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2],
           z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
    r[0] = x, r[1] = y, r[2] = z;
}
static void mat_mul3(double *r, const double *a, const double *b) {
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
    memcpy(r, t, sizeof t);
}
static void quat_mul(double *r, const double *a, const double *b) {
    double t[4] = {a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                   a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]};
    memcpy(r, t, sizeof t);
}
static void quat_normalize(double *q) {
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n < MINVAL) {
        q[0] = 1, q[1] = q[2] = q[3] = 0;
        return;
    }
    for (int k = 0; k < 4; k++) q[k] /= n;
}
static void quat_to_mat(double *m, const double *q) {
    double w = q[0], x = q[1], y = q[2], z = q[3];
    m[0] = w * w + x * x - y * y - z * z, m[1] = 2 * (x * y - w * z), m[2] = 2 * (x * z + w * y);
    m[3] = 2 * (x * y + w * z), m[4] = w * w - x * x + y * y - z * z, m[5] = 2 * (y * z - w * x);
    m[6] = 2 * (x * z - w * y), m[7] = 2 * (y * z + w * x), m[8] = w * w - x * x - y * y + z * z;
}
static void axis_angle_to_quat(double *q, const double *axis, double angle) {
    double s = sin(angle * 0.5);
    q[0] = cos(angle * 0.5), q[1] = axis[0] * s, q[2] = axis[1] * s, q[3] = axis[2] * s;
}

/* spatial algebra on 6-vectors [rotational(3); translational(3)] referred to the subtree com of the kinematic tree */
static void cross_motion(double *r, const double *v, const double *s) {
    double a[3], b[3], c[3];
    cross3(a, v, s);          /* w x sw */
    cross3(b, v, s + 3);      /* w x sv */
    cross3(c, v + 3, s);      /* v x sw */
    r[0] = a[0], r[1] = a[1], r[2] = a[2], r[3] = b[0] + c[0], r[4] = b[1] + c[1], r[5] = b[2] + c[2];
}
static void cross_force(double *r, const double *v, const double *f) {
    double a[3], b[3], c[3];
    cross3(a, v, f);          /* w x f_ang */
    cross3(b, v + 3, f + 3);  /* v x f_lin */
    cross3(c, v, f + 3);      /* w x f_lin */
    r[0] = a[0] + b[0], r[1] = a[1] + b[1], r[2] = a[2] + b[2], r[3] = c[0], r[4] = c[1], r[5] = c[2];
}
/* 10-number com-based inertia [Ixx Iyy Izz Ixy Ixz Iyz, m*dx m*dy m*dz, m] times a motion vector -> force vector */
static void inert_mul(double *r, const double *I, const double *v) {
    const double *w = v, *l = v + 3, *md = I + 6;
    double c1[3], c2[3];
    cross3(c1, md, l); /* md x v */
    cross3(c2, md, w); /* md x w */
    r[0] = I[0] * w[0] + I[3] * w[1] + I[4] * w[2] + c1[0];
    r[1] = I[3] * w[0] + I[1] * w[1] + I[5] * w[2] + c1[1];
    r[2] = I[4] * w[0] + I[5] * w[1] + I[2] * w[2] + c1[2];
    r[3] = I[9] * l[0] - c2[0], r[4] = I[9] * l[1] - c2[1], r[5] = I[9] * l[2] - c2[2];
}

/* ---- model blob ---------------------------------------------------------------------------------------------- */
#define TAKE_I(dst, cnt) do { for (int k_ = 0; k_ < (cnt); k_++) (dst)[k_] = (int)*p++; } while (0)
#define TAKE_D(dst, cnt) do { for (int k_ = 0; k_ < (cnt); k_++) ((double *)(dst))[k_] = *p++; } while (0)
int mjo_model_from_blob(mjo_model *m, const double *blob, int n) {
    const double *p = blob;
    memset(m, 0, sizeof *m);
    m->nq = (int)*p++, m->nv = (int)*p++, m->nu = (int)*p++, m->nbody = (int)*p++, m->njnt = (int)*p++, m->ngeom = (int)*p++;
    m->npair = (int)*p++, m->integrator = (int)*p++, m->solver = (int)*p++, m->iterations = (int)*p++;
    if (m->nq > MJO_MAXQ || m->nv > MJO_MAXV || m->nu > MJO_MAXU || m->nbody > MJO_MAXB || m->njnt > MJO_MAXJ ||
        m->ngeom > MJO_MAXG || m->npair > MJO_MAXPAIR)
        return -1;
    m->timestep = *p++;
    TAKE_D(m->gravity, 3);
    m->meaninertia = *p++, m->density = *p++, m->viscosity = *p++;
    int nb = m->nbody, nj = m->njnt, nv = m->nv, ng = m->ngeom, np = m->npair, nu = m->nu;
    TAKE_I(m->body_parentid, nb); TAKE_I(m->body_rootid, nb); TAKE_I(m->body_jntadr, nb); TAKE_I(m->body_jntnum, nb);
    TAKE_I(m->body_dofadr, nb); TAKE_I(m->body_dofnum, nb);
    TAKE_D(m->body_pos, 3 * nb); TAKE_D(m->body_quat, 4 * nb); TAKE_D(m->body_mass, nb); TAKE_D(m->body_ipos, 3 * nb);
    TAKE_D(m->body_inertia, 9 * nb); TAKE_D(m->body_invweight0, 2 * nb); TAKE_D(m->body_fluidbox, 3 * nb); TAKE_D(m->body_imat, 9 * nb);
    TAKE_I(m->jnt_type, nj); TAKE_I(m->jnt_qposadr, nj); TAKE_I(m->jnt_dofadr, nj); TAKE_I(m->jnt_bodyid, nj); TAKE_I(m->jnt_limited, nj);
    TAKE_D(m->jnt_pos, 3 * nj); TAKE_D(m->jnt_axis, 3 * nj); TAKE_D(m->jnt_range, 2 * nj); TAKE_D(m->jnt_stiffness, nj);
    TAKE_D(m->jnt_margin, nj); TAKE_D(m->jnt_solref, 2 * nj); TAKE_D(m->jnt_solimp, 5 * nj);
    TAKE_I(m->dof_bodyid, nv); TAKE_I(m->dof_jntid, nv); TAKE_I(m->dof_parentid, nv);
    TAKE_D(m->dof_armature, nv); TAKE_D(m->dof_damping, nv); TAKE_D(m->dof_invweight0, nv);
    TAKE_D(m->qpos0, m->nq); TAKE_D(m->qpos_spring, m->nq);
    TAKE_I(m->geom_type, ng); TAKE_I(m->geom_bodyid, ng);
    TAKE_D(m->geom_size, 3 * ng); TAKE_D(m->geom_pos, 3 * ng); TAKE_D(m->geom_mat, 9 * ng);
    TAKE_I(m->pair_geom1, np); TAKE_I(m->pair_geom2, np); TAKE_I(m->pair_condim, np);
    TAKE_D(m->pair_friction, 3 * np); TAKE_D(m->pair_margin, np); TAKE_D(m->pair_solref, 2 * np); TAKE_D(m->pair_solimp, 5 * np);
    TAKE_I(m->actuator_dofadr, nu);
    TAKE_D(m->actuator_gear, nu); TAKE_D(m->actuator_ctrlrange, 2 * nu);
    m->ntendon = (int)*p++;
    if (m->ntendon > MJO_MAXT) return -1;
    for (int t = 0; t < m->ntendon; t++) {
        m->tendon_num[t] = (int)*p++;
        if (m->tendon_num[t] > MJO_MAXWRAP) return -1;
        for (int k = 0; k < m->tendon_num[t]; k++) m->wrap_qposadr[t][k] = (int)*p++, m->wrap_dofadr[t][k] = (int)*p++, m->wrap_coef[t][k] = *p++;
    }
    return (int)(p - blob) == n ? 0 : -2;
}

void mjo_reset_data(const mjo_model *m, mjo_data *d) {
    memset(d, 0, sizeof *d);
    memcpy(d->qpos, m->qpos0, sizeof(double) * m->nq);
}

/* ---- position stage ------------------------------------------------------------------------------------------ */
static void kinematics(const mjo_model *m, mjo_data *d) {
    memset(d->xpos[0], 0, 24), memset(d->xquat[0], 0, 32);
    d->xquat[0][0] = 1;
    quat_to_mat(d->xmat[0], d->xquat[0]);
    memset(d->xipos[0], 0, 24);
    for (int b = 1; b < m->nbody; b++) {
        int p = m->body_parentid[b], ja = m->body_jntadr[b], jn = m->body_jntnum[b];
        double pos[3], quat[4];
        if (jn == 1 && m->jnt_type[ja] == MJO_FREE) {
            const double *q = d->qpos + m->jnt_qposadr[ja];
            memcpy(pos, q, 24), memcpy(quat, q + 3, 32);
            quat_normalize(quat);
            memcpy(d->xanchor[ja], pos, 24);
            d->xaxis[ja][0] = 0, d->xaxis[ja][1] = 0, d->xaxis[ja][2] = 1;
        } else {
            double t[3];
            mat_vec3(t, d->xmat[p], m->body_pos[b]);
            for (int k = 0; k < 3; k++) pos[k] = d->xpos[p][k] + t[k];
            quat_mul(quat, d->xquat[p], m->body_quat[b]);
            for (int j = ja; j < ja + jn; j++) {
                double R[9], qloc[4];
                quat_to_mat(R, quat);
                mat_vec3(t, R, m->jnt_pos[j]);
                for (int k = 0; k < 3; k++) d->xanchor[j][k] = pos[k] + t[k];
                mat_vec3(d->xaxis[j], R, m->jnt_axis[j]);
                double q = d->qpos[m->jnt_qposadr[j]] - m->qpos0[m->jnt_qposadr[j]];
                if (m->jnt_type[j] == MJO_HINGE) {
                    axis_angle_to_quat(qloc, m->jnt_axis[j], q);
                    quat_mul(quat, quat, qloc);
                    quat_to_mat(R, quat);
                    mat_vec3(t, R, m->jnt_pos[j]); /* off-centre rotation: keep the anchor fixed */
                    for (int k = 0; k < 3; k++) pos[k] = d->xanchor[j][k] - t[k];
                } else { /* slide */
                    for (int k = 0; k < 3; k++) pos[k] += d->xaxis[j][k] * q;
                }
            }
        }
        quat_normalize(quat);
        memcpy(d->xpos[b], pos, 24), memcpy(d->xquat[b], quat, 32);
        quat_to_mat(d->xmat[b], quat);
        double t[3];
        mat_vec3(t, d->xmat[b], m->body_ipos[b]);
        for (int k = 0; k < 3; k++) d->xipos[b][k] = pos[k] + t[k];
    }
    for (int g = 0; g < m->ngeom; g++) {
        int b = m->geom_bodyid[g];
        double t[3];
        mat_vec3(t, d->xmat[b], m->geom_pos[g]);
        for (int k = 0; k < 3; k++) d->geom_xpos[g][k] = d->xpos[b][k] + t[k];
        mat_mul3(d->geom_xmat[g], d->xmat[b], m->geom_mat[g]);
    }
}

static void com_pos(const mjo_model *m, mjo_data *d) {
    double mass[MJO_MAXB];
    for (int b = 0; b < m->nbody; b++) {
        mass[b] = m->body_mass[b];
        for (int k = 0; k < 3; k++) d->subtree_com[b][k] = m->body_mass[b] * d->xipos[b][k];
    }
    for (int b = m->nbody - 1; b > 0; b--) {
        int p = m->body_parentid[b];
        mass[p] += mass[b];
        for (int k = 0; k < 3; k++) d->subtree_com[p][k] += d->subtree_com[b][k];
    }
    for (int b = 0; b < m->nbody; b++)
        for (int k = 0; k < 3; k++) d->subtree_com[b][k] = mass[b] > MINVAL ? d->subtree_com[b][k] / mass[b] : d->xipos[b][k];
    memset(d->cinert[0], 0, sizeof d->cinert[0]);
    for (int b = 1; b < m->nbody; b++) {
        const double *O = d->subtree_com[m->body_rootid[b]];
        double off[3], Iw[9], t[9], Rt[9];
        for (int k = 0; k < 3; k++) off[k] = d->xipos[b][k] - O[k];
        /* world-orientation inertia: R I R^T */
        mat_mul3(t, d->xmat[b], m->body_inertia[b]);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Rt[3 * i + j] = d->xmat[b][3 * j + i];
        mat_mul3(Iw, t, Rt);
        double mm = m->body_mass[b], dd = dot3(off, off);
        double *c = d->cinert[b];
        c[0] = Iw[0] + mm * (dd - off[0] * off[0]), c[1] = Iw[4] + mm * (dd - off[1] * off[1]), c[2] = Iw[8] + mm * (dd - off[2] * off[2]);
        c[3] = Iw[1] - mm * off[0] * off[1], c[4] = Iw[2] - mm * off[0] * off[2], c[5] = Iw[5] - mm * off[1] * off[2];
        c[6] = mm * off[0], c[7] = mm * off[1], c[8] = mm * off[2], c[9] = mm;
    }
    for (int j = 0; j < m->njnt; j++) {
        int b = m->jnt_bodyid[j], a = m->jnt_dofadr[j];
        const double *O = d->subtree_com[m->body_rootid[b]];
        double off[3];
        for (int k = 0; k < 3; k++) off[k] = O[k] - d->xanchor[j][k];
        if (m->jnt_type[j] == MJO_FREE) {
            for (int k = 0; k < 3; k++) {
                memset(d->cdof[a + k], 0, 48);
                d->cdof[a + k][3 + k] = 1.0;
                double ax[3] = {d->xmat[b][k], d->xmat[b][3 + k], d->xmat[b][6 + k]}; /* body axis k in world coordinates */
                memcpy(d->cdof[a + 3 + k], ax, 24);
                cross3(d->cdof[a + 3 + k] + 3, ax, off);
            }
        } else if (m->jnt_type[j] == MJO_HINGE) {
            memcpy(d->cdof[a], d->xaxis[j], 24);
            cross3(d->cdof[a] + 3, d->xaxis[j], off);
        } else {
            memset(d->cdof[a], 0, 24);
            memcpy(d->cdof[a] + 3, d->xaxis[j], 24);
        }
    }
}

static void crb(const mjo_model *m, mjo_data *d) {
    double c[MJO_MAXB][10];
    memcpy(c, d->cinert, sizeof c);
    for (int b = m->nbody - 1; b > 0; b--) {
        int p = m->body_parentid[b];
        if (p > 0)
            for (int k = 0; k < 10; k++) c[p][k] += c[b][k];
    }
    memset(d->qM, 0, sizeof d->qM);
    for (int i = 0; i < m->nv; i++) {
        double buf[6];
        inert_mul(buf, c[m->dof_bodyid[i]], d->cdof[i]);
        for (int j = i; j >= 0; j = m->dof_parentid[j]) {
            double v = 0;
            for (int k = 0; k < 6; k++) v += d->cdof[j][k] * buf[k];
            d->qM[i][j] = d->qM[j][i] = v;
        }
        d->qM[i][i] += m->dof_armature[i];
    }
}

/* dense Cholesky A = L L^T (lower), n x n inside MJO_MAXV-strided storage; returns 0 on success */
static int chol_factor(int n, double A[][MJO_MAXV], double L[][MJO_MAXV]) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++) {
            double s = A[i][j];
            for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
            if (i == j) {
                if (s < MINVAL) s = MINVAL;
                L[i][i] = sqrt(s);
            } else {
                L[i][j] = s / L[j][j];
            }
        }
    return 0;
}
static void chol_solve(int n, double L[][MJO_MAXV], double *x) { /* in place */
    for (int i = 0; i < n; i++) {
        double s = x[i];
        for (int k = 0; k < i; k++) s -= L[i][k] * x[k];
        x[i] = s / L[i][i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = x[i];
        for (int k = i + 1; k < n; k++) s -= L[k][i] * x[k];
        x[i] = s / L[i][i];
    }
}

/* ---- collision ------------------------------------------------------------------------------------------------ */
static void make_frame(double *f) {
    normalize3(f);
    if (norm3(f + 3) < 0.5) {
        f[3] = f[4] = f[5] = 0;
        if (f[1] < 0.5 && f[1] > -0.5)
            f[4] = 1;
        else
            f[5] = 1;
    }
    double t = dot3(f, f + 3);
    for (int k = 0; k < 3; k++) f[3 + k] -= t * f[k];
    normalize3(f + 3);
    cross3(f + 6, f, f + 3);
}

static int add_contact(const mjo_model *m, mjo_data *d, int pair, double dist, const double *pos, const double *normal,
                       const double *tangent) {
    if (!(dist < m->pair_margin[pair]) || d->ncon >= MJO_MAXCON) return 0;
    mjo_contact *c = &d->contact[d->ncon++];
    memset(c, 0, sizeof *c);
    c->dist = dist;
    memcpy(c->pos, pos, 24), memcpy(c->frame, normal, 24);
    if (tangent) memcpy(c->frame + 3, tangent, 24);
    make_frame(c->frame);
    c->geom1 = m->pair_geom1[pair], c->geom2 = m->pair_geom2[pair], c->dim = m->pair_condim[pair];
    c->friction = m->pair_friction[pair][0], c->margin = m->pair_margin[pair];
    memcpy(c->solref, m->pair_solref[pair], 16), memcpy(c->solimp, m->pair_solimp[pair], 40);
    c->efc_address = -1;
    return 1;
}

static void sphere_sphere_raw(const mjo_model *m, mjo_data *d, int pair, const double *p1, double r1, const double *p2, double r2,
                              const double *tangent) {
    double n[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
    double dist = norm3(n);
    if (dist < MINVAL) {
        n[0] = 1, n[1] = n[2] = 0;
    } else {
        for (int k = 0; k < 3; k++) n[k] /= dist;
    }
    double pos[3];
    for (int k = 0; k < 3; k++) pos[k] = p1[k] + n[k] * (r1 + 0.5 * (dist - r1 - r2));
    add_contact(m, d, pair, dist - r1 - r2, pos, n, tangent);
}

/* Signed distance from point p to the solid cylinder (centre c, unit axis u, radius R, half height H); grad = its gradient (the
 * outward direction at the nearest surface point, which is p - sd * grad).  A convex function of p. */
static double cylinder_sd(const double *p, const double *c, const double *u, double R, double H, double *grad) {
    const double dv[3] = {p[0] - c[0], p[1] - c[1], p[2] - c[2]};
    const double a = dot3(dv, u);
    double rv[3] = {dv[0] - a * u[0], dv[1] - a * u[1], dv[2] - a * u[2]};
    const double rho = norm3(rv);
    if (rho > MINVAL) {
        for (int k = 0; k < 3; k++) rv[k] /= rho;
    } else { /* on the axis: any direction orthogonal to it */
        const double e[3] = {fabs(u[0]) < 0.5 ? 1.0 : 0.0, fabs(u[0]) < 0.5 ? 0.0 : 1.0, 0.0};
        const double t = dot3(e, u);
        for (int k = 0; k < 3; k++) rv[k] = e[k] - t * u[k];
        normalize3(rv);
    }
    const double sa = a >= 0 ? 1.0 : -1.0, ea = fabs(a) - H, er = rho - R;
    if (ea <= 0 && er <= 0) { /* inside: the nearer of the cap and the wall */
        if (ea > er) {
            for (int k = 0; k < 3; k++) grad[k] = sa * u[k];
            return ea;
        }
        for (int k = 0; k < 3; k++) grad[k] = rv[k];
        return er;
    }
    if (er <= 0) {
        for (int k = 0; k < 3; k++) grad[k] = sa * u[k];
        return ea;
    }
    if (ea <= 0) {
        for (int k = 0; k < 3; k++) grad[k] = rv[k];
        return er;
    }
    const double sd = sqrt(ea * ea + er * er); /* nearest point on the rim */
    for (int k = 0; k < 3; k++) grad[k] = (ea * sa * u[k] + er * rv[k]) / sd;
    return sd;
}

/* Capsule (segment centre pc, unit axis zc, half length hc, radius rc) against a cylinder.  MuJoCo has no analytic function for this
 * pair (it runs its general convex collider); this is the exact geometry instead: the signed distance of the segment point P(t) to the
 * cylinder is convex in t, so its minimiser is where the directional derivative grad . axis changes sign -- found by bisection. */
static void capsule_cylinder(const mjo_model *m, mjo_data *d, int pair, const double *pc, const double *zc, double hc, double rc,
                             const double *py, const double *zy, double R, double H) {
    const double cc[3] = {pc[0] - py[0], pc[1] - py[1], pc[2] - py[2]};
    if (norm3(cc) > hc + rc + sqrt(R * R + H * H) + m->pair_margin[pair]) return; /* bounding spheres apart */
    double lo = -hc, hi = hc, p[3], g[3], glo[3], ghi[3], t;
    for (int k = 0; k < 3; k++) p[k] = pc[k] + lo * zc[k];
    cylinder_sd(p, py, zy, R, H, glo);
    if (dot3(glo, zc) >= 0) {
        t = lo;
        memcpy(g, glo, sizeof g);
    } else {
        for (int k = 0; k < 3; k++) p[k] = pc[k] + hi * zc[k];
        cylinder_sd(p, py, zy, R, H, ghi);
        if (dot3(ghi, zc) <= 0) {
            t = hi;
            memcpy(g, ghi, sizeof g);
        } else {
            for (int it = 0; it < 60; it++) {
                const double mid = 0.5 * (lo + hi);
                for (int k = 0; k < 3; k++) p[k] = pc[k] + mid * zc[k];
                cylinder_sd(p, py, zy, R, H, g);
                if (dot3(g, zc) < 0)
                    lo = mid, memcpy(glo, g, sizeof g);
                else
                    hi = mid, memcpy(ghi, g, sizeof g);
            }
            t = 0.5 * (lo + hi);
            /* the minimiser may sit on a kink of the distance (cap and wall, or a face and the rim, equally near): take the element of
             * the subgradient that is stationary along the segment -- the blend of the two one-sided gradients with zero slope */
            const double a = dot3(glo, zc), b = dot3(ghi, zc), w = b / (b - a);
            for (int k = 0; k < 3; k++) g[k] = w * glo[k] + (1.0 - w) * ghi[k];
            normalize3(g);
        }
    }
    for (int k = 0; k < 3; k++) p[k] = pc[k] + t * zc[k];
    double gt[3];
    const double sd = cylinder_sd(p, py, zy, R, H, gt), dist = sd - rc;
    double n[3], pos[3];
    for (int k = 0; k < 3; k++) n[k] = -g[k], pos[k] = p[k] - g[k] * (rc + 0.5 * dist); /* normal: capsule -> cylinder; midpoint of the gap */
    add_contact(m, d, pair, dist, pos, n, 0);
}

static void collide_pair(const mjo_model *m, mjo_data *d, int pair) {
    int g1 = m->pair_geom1[pair], g2 = m->pair_geom2[pair];
    int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
    if (t1 > t2) { /* the narrow-phase functions are written for type1 <= type2 */
        int t = g1;
        g1 = g2, g2 = t, t = t1, t1 = t2, t2 = t;
    }
    const double *p1 = d->geom_xpos[g1], *p2 = d->geom_xpos[g2], *R1 = d->geom_xmat[g1], *R2 = d->geom_xmat[g2];
    int before = d->ncon;
    if (t1 == MJO_PLANE) {
        double n[3] = {R1[2], R1[5], R1[8]};
        double r = m->geom_size[g2][0];
        if (t2 == MJO_SPHERE) {
            double v[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
            double dist = dot3(v, n) - r, pos[3];
            for (int k = 0; k < 3; k++) pos[k] = p2[k] - n[k] * (r + 0.5 * dist);
            add_contact(m, d, pair, dist, pos, n, 0);
        } else { /* capsule: a sphere test at each end of the segment, frames aligned with the capsule axis */
            double axis[3] = {R2[2], R2[5], R2[8]}, half = m->geom_size[g2][1];
            for (int s = 1; s >= -1; s -= 2) {
                double c[3], v[3], pos[3];
                for (int k = 0; k < 3; k++) c[k] = p2[k] + s * half * axis[k], v[k] = c[k] - p1[k];
                double dist = dot3(v, n) - r;
                for (int k = 0; k < 3; k++) pos[k] = c[k] - n[k] * (r + 0.5 * dist);
                add_contact(m, d, pair, dist, pos, n, axis);
            }
        }
    } else if (t1 == MJO_SPHERE && t2 == MJO_SPHERE) {
        sphere_sphere_raw(m, d, pair, p1, m->geom_size[g1][0], p2, m->geom_size[g2][0], 0);
    } else if (t1 == MJO_SPHERE && t2 == MJO_CAPSULE) {
        double axis[3] = {R2[2], R2[5], R2[8]}, half = m->geom_size[g2][1];
        double v[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
        double x = dot3(v, axis);
        x = x > half ? half : (x < -half ? -half : x);
        double c[3] = {p2[0] + x * axis[0], p2[1] + x * axis[1], p2[2] + x * axis[2]};
        sphere_sphere_raw(m, d, pair, p1, m->geom_size[g1][0], c, m->geom_size[g2][0], 0);
    } else if (t1 == MJO_CAPSULE && t2 == MJO_CAPSULE) {
        /* nearest points of the two segments */
        double a1[3] = {R1[2], R1[5], R1[8]}, a2[3] = {R2[2], R2[5], R2[8]};
        double h1 = m->geom_size[g1][1], h2 = m->geom_size[g2][1];
        double dif[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
        double ma = 1.0, mb = -dot3(a1, a2), mc = 1.0, u = -dot3(a1, dif), v = dot3(a2, dif);
        double det = ma * mc - mb * mb, x1, x2;
        if (fabs(det) >= 1e-12) {
            x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
            if (x1 > h1) {
                x1 = h1, x2 = (v - mb * h1) / mc;
            } else if (x1 < -h1) {
                x1 = -h1, x2 = (v + mb * h1) / mc;
            }
            if (x2 > h2) {
                x2 = h2, x1 = (u - mb * h2) / ma;
                x1 = x1 > h1 ? h1 : (x1 < -h1 ? -h1 : x1);
            } else if (x2 < -h2) {
                x2 = -h2, x1 = (u + mb * h2) / ma;
                x1 = x1 > h1 ? h1 : (x1 < -h1 ? -h1 : x1);
            }
        } else { /* parallel: midpoint of the overlap */
            x1 = 0, x2 = v / mc;
            x2 = x2 > h2 ? h2 : (x2 < -h2 ? -h2 : x2);
            x1 = (u - mb * x2) / ma;
            x1 = x1 > h1 ? h1 : (x1 < -h1 ? -h1 : x1);
        }
        double c1[3], c2[3];
        for (int k = 0; k < 3; k++) c1[k] = p1[k] + x1 * a1[k], c2[k] = p2[k] + x2 * a2[k];
        sphere_sphere_raw(m, d, pair, c1, m->geom_size[g1][0], c2, m->geom_size[g2][0], 0);
    } else if (t1 == MJO_CAPSULE && t2 == MJO_CYLINDER) {
        const double a1[3] = {R1[2], R1[5], R1[8]}, a2[3] = {R2[2], R2[5], R2[8]};
        capsule_cylinder(m, d, pair, p1, a1, m->geom_size[g1][1], m->geom_size[g1][0], p2, a2, m->geom_size[g2][0], m->geom_size[g2][1]);
    }
    if (g1 != m->pair_geom1[pair]) /* restore the model's geom order: normal points from pair_geom1 to pair_geom2 */
        for (int c = before; c < d->ncon; c++) {
            for (int k = 0; k < 3; k++) d->contact[c].frame[k] = -d->contact[c].frame[k];
            d->contact[c].frame[3] = d->contact[c].frame[4] = d->contact[c].frame[5] = 0;
            make_frame(d->contact[c].frame);
        }
}

static void collision(const mjo_model *m, mjo_data *d) {
    d->ncon = 0;
    for (int p = 0; p < m->npair; p++) collide_pair(m, d, p);
}

/* ---- constraints ---------------------------------------------------------------------------------------------- */
/* velocity of `point` moving with `body` per unit qvel[i]: translational (jp) and rotational (jr) parts */
static void jac_point(const mjo_model *m, const mjo_data *d, int body, const double *point, double jp[][3], double jr[][3]) {
    for (int i = 0; i < m->nv; i++) memset(jp[i], 0, 24), memset(jr[i], 0, 24);
    if (body <= 0) return;
    const double *O = d->subtree_com[m->body_rootid[body]];
    double off[3] = {point[0] - O[0], point[1] - O[1], point[2] - O[2]};
    int b = body;
    while (b > 0 && m->body_dofnum[b] == 0) b = m->body_parentid[b];
    if (b <= 0) return;
    for (int i = m->body_dofadr[b] + m->body_dofnum[b] - 1; i >= 0; i = m->dof_parentid[i]) {
        double t[3];
        cross3(t, d->cdof[i], off);
        for (int k = 0; k < 3; k++) jr[i][k] = d->cdof[i][k], jp[i][k] = d->cdof[i][3 + k] + t[k];
    }
}

static double impedance(const double *solimp, double pos, double margin) {
    double dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
    dmin = dmin < MINIMP ? MINIMP : (dmin > MAXIMP ? MAXIMP : dmin);
    dmax = dmax < MINIMP ? MINIMP : (dmax > MAXIMP ? MAXIMP : dmax);
    width = width < MINVAL ? MINVAL : width;
    mid = mid < MINIMP ? MINIMP : (mid > MAXIMP ? MAXIMP : mid);
    power = power < 1 ? 1 : power;
    double x = fabs(pos - margin) / width, y;
    if (x >= 1)
        y = 1;
    else if (x <= 0)
        y = 0;
    else if (power == 1)
        y = x;
    else if (x <= mid)
        y = pow(x, power) / pow(mid, power - 1);
    else
        y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
    return dmin + y * (dmax - dmin);
}

static void set_row_params(const mjo_model *m, mjo_data *d, int row, const double *solref, const double *solimp, double pos,
                           double margin, double diag_approx) {
    double timeconst = solref[0], dampratio = solref[1];
    double dmax = solimp[1] < MINIMP ? MINIMP : (solimp[1] > MAXIMP ? MAXIMP : solimp[1]);
    if (timeconst < 2 * m->timestep) timeconst = 2 * m->timestep; /* refsafe */
    double k = 1.0 / (dmax * dmax * timeconst * timeconst * dampratio * dampratio), b = 2.0 / (dmax * timeconst);
    double imp = impedance(solimp, pos, margin);
    double R = (1 - imp) * diag_approx / imp;
    if (R < MINVAL) R = MINVAL;
    d->efc_pos[row] = pos, d->efc_margin[row] = margin, d->efc_R[row] = R, d->efc_D[row] = 1.0 / R;
    d->efc_KBIP[row][0] = k, d->efc_KBIP[row][1] = b, d->efc_KBIP[row][2] = imp, d->efc_KBIP[row][3] = 0;
}

static void make_constraint(const mjo_model *m, mjo_data *d) {
    int n = 0, nv = m->nv;
    /* joint limits */
    for (int j = 0; j < m->njnt; j++) {
        if (!m->jnt_limited[j] || (m->jnt_type[j] != MJO_HINGE && m->jnt_type[j] != MJO_SLIDE)) continue;
        double value = d->qpos[m->jnt_qposadr[j]];
        for (int side = -1; side <= 1; side += 2) {
            double dist = side * (m->jnt_range[j][side < 0 ? 0 : 1] - value);
            if (dist < m->jnt_margin[j] && n < MJO_MAXEFC) {
                memset(d->efc_J[n], 0, sizeof(double) * nv);
                d->efc_J[n][m->jnt_dofadr[j]] = -side;
                set_row_params(m, d, n, m->jnt_solref[j], m->jnt_solimp[j], dist, m->jnt_margin[j], m->dof_invweight0[m->jnt_dofadr[j]]);
                n++;
            }
        }
    }
    /* contacts */
    static __thread double jp1[MJO_MAXV][3], jr1[MJO_MAXV][3], jp2[MJO_MAXV][3], jr2[MJO_MAXV][3];
    for (int c = 0; c < d->ncon; c++) {
        mjo_contact *con = &d->contact[c];
        int b1 = m->geom_bodyid[con->geom1], b2 = m->geom_bodyid[con->geom2];
        int rows = con->dim == 1 ? 1 : 2 * (con->dim - 1);
        if (n + rows > MJO_MAXEFC) break;
        jac_point(m, d, b1, con->pos, jp1, jr1);
        jac_point(m, d, b2, con->pos, jp2, jr2);
        double Jc[3][MJO_MAXV]; /* relative velocity (body2 - body1) along the contact frame axes */
        for (int a = 0; a < 3; a++)
            for (int i = 0; i < nv; i++) {
                double rel[3] = {jp2[i][0] - jp1[i][0], jp2[i][1] - jp1[i][1], jp2[i][2] - jp1[i][2]};
                Jc[a][i] = dot3(con->frame + 3 * a, rel);
            }
        double tran = m->body_invweight0[b1][0] + m->body_invweight0[b2][0];
        con->efc_address = n;
        if (con->dim == 1) {
            memcpy(d->efc_J[n], Jc[0], sizeof(double) * nv);
            set_row_params(m, d, n, con->solref, con->solimp, con->dist, con->margin, tran);
            n++;
        } else { /* pyramidal cone, condim 3: edges normal +- mu * tangent_k */
            double mu = con->friction;
            for (int e = 0; e < rows; e++) {
                int tdir = 1 + e / 2;
                double sgn = (e & 1) ? -1.0 : 1.0;
                for (int i = 0; i < nv; i++) d->efc_J[n + e][i] = Jc[0][i] + sgn * mu * Jc[tdir][i];
                set_row_params(m, d, n + e, con->solref, con->solimp, con->dist, con->margin, tran + mu * mu * tran);
            }
            double Rpy = 2 * mu * mu * d->efc_R[n];
            if (Rpy < MINVAL) Rpy = MINVAL;
            for (int e = 0; e < rows; e++) d->efc_R[n + e] = Rpy, d->efc_D[n + e] = 1.0 / Rpy;
            n += rows;
        }
    }
    d->nefc = n;
}

/* ---- velocity stage ------------------------------------------------------------------------------------------- */
static void com_vel(const mjo_model *m, mjo_data *d) {
    memset(d->cvel[0], 0, 48);
    for (int b = 1; b < m->nbody; b++) {
        double v[6];
        memcpy(v, d->cvel[m->body_parentid[b]], 48);
        for (int j = m->body_jntadr[b]; j < m->body_jntadr[b] + m->body_jntnum[b]; j++) {
            int a = m->jnt_dofadr[j];
            if (m->jnt_type[j] == MJO_FREE) {
                for (int k = 0; k < 3; k++) {
                    memset(d->cdof_dot[a + k], 0, 48);
                    for (int c = 0; c < 6; c++) v[c] += d->cdof[a + k][c] * d->qvel[a + k];
                }
                for (int k = 3; k < 6; k++) cross_motion(d->cdof_dot[a + k], v, d->cdof[a + k]);
                for (int k = 3; k < 6; k++)
                    for (int c = 0; c < 6; c++) v[c] += d->cdof[a + k][c] * d->qvel[a + k];
            } else {
                cross_motion(d->cdof_dot[a], v, d->cdof[a]);
                for (int c = 0; c < 6; c++) v[c] += d->cdof[a][c] * d->qvel[a];
            }
        }
        memcpy(d->cvel[b], v, 48);
    }
}

/* Fluid forces of the medium (option density / viscosity): MuJoCo's inertia-box model (mj_passive, engine_passive.c
 * mj_inertiaBoxFluidModel; documentation "Computation" > passive forces): each body is replaced by the box with its mass and
 * principal moments; in that box's frame (at the body's centre of mass) the velocity (w, v) produces
 *   viscous:  torque -= pi d^3 mu w,              force -= 3 pi d mu v,                     d = mean box edge
 *   drag:     torque_k -= rho b_k (b_i^4 + b_j^4) |w_k| w_k / 64,   force_k -= rho b_i b_j |v_k| v_k / 2
 * and the wrench is applied at the centre of mass (mj_applyFT).  Needs cvel (com_vel) and the position stage. */
static void fluid(const mjo_model *m, mjo_data *d) {
    if (m->density <= 0 && m->viscosity <= 0) return;
    const double PI = 3.14159265358979323846;
    for (int b = 1; b < m->nbody; b++) {
        if (m->body_mass[b] < 1e-15) continue;
        double R[9]; /* ximat = xmat * imat */
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                R[3 * i + j] = d->xmat[b][3 * i] * m->body_imat[b][j] + d->xmat[b][3 * i + 1] * m->body_imat[b][3 + j] + d->xmat[b][3 * i + 2] * m->body_imat[b][6 + j];
        /* velocity of the body's centre of mass: cvel is [angular; linear at the tree com] */
        const double off[3] = {d->xipos[b][0] - d->subtree_com[m->body_rootid[b]][0], d->xipos[b][1] - d->subtree_com[m->body_rootid[b]][1],
                               d->xipos[b][2] - d->subtree_com[m->body_rootid[b]][2]};
        const double *w = d->cvel[b], *vl = d->cvel[b] + 3;
        const double vc[3] = {vl[0] + w[1] * off[2] - w[2] * off[1], vl[1] + w[2] * off[0] - w[0] * off[2], vl[2] + w[0] * off[1] - w[1] * off[0]};
        double lw[3], lv[3], lt[3], lf[3];
        for (int k = 0; k < 3; k++) {
            lw[k] = R[k] * w[0] + R[3 + k] * w[1] + R[6 + k] * w[2];
            lv[k] = R[k] * vc[0] + R[3 + k] * vc[1] + R[6 + k] * vc[2];
            lt[k] = 0, lf[k] = 0;
        }
        const double *bx = m->body_fluidbox[b];
        if (m->viscosity > 0) {
            const double diam = (bx[0] + bx[1] + bx[2]) / 3.0;
            for (int k = 0; k < 3; k++) lt[k] += -PI * diam * diam * diam * m->viscosity * lw[k], lf[k] += -3.0 * PI * diam * m->viscosity * lv[k];
        }
        if (m->density > 0) {
            lf[0] -= 0.5 * m->density * bx[1] * bx[2] * fabs(lv[0]) * lv[0];
            lf[1] -= 0.5 * m->density * bx[0] * bx[2] * fabs(lv[1]) * lv[1];
            lf[2] -= 0.5 * m->density * bx[0] * bx[1] * fabs(lv[2]) * lv[2];
            lt[0] -= m->density * bx[0] * (bx[1] * bx[1] * bx[1] * bx[1] + bx[2] * bx[2] * bx[2] * bx[2]) * fabs(lw[0]) * lw[0] / 64.0;
            lt[1] -= m->density * bx[1] * (bx[0] * bx[0] * bx[0] * bx[0] + bx[2] * bx[2] * bx[2] * bx[2]) * fabs(lw[1]) * lw[1] / 64.0;
            lt[2] -= m->density * bx[2] * (bx[0] * bx[0] * bx[0] * bx[0] + bx[1] * bx[1] * bx[1] * bx[1]) * fabs(lw[2]) * lw[2] / 64.0;
        }
        double gt[3], gf[3]; /* back to the world frame; torque about the tree com = torque + off x force */
        for (int k = 0; k < 3; k++)
            gt[k] = R[3 * k] * lt[0] + R[3 * k + 1] * lt[1] + R[3 * k + 2] * lt[2], gf[k] = R[3 * k] * lf[0] + R[3 * k + 1] * lf[1] + R[3 * k + 2] * lf[2];
        const double tq[3] = {gt[0] + off[1] * gf[2] - off[2] * gf[1], gt[1] + off[2] * gf[0] - off[0] * gf[2], gt[2] + off[0] * gf[1] - off[1] * gf[0]};
        int bb = b;
        while (bb > 0 && m->body_dofnum[bb] == 0) bb = m->body_parentid[bb];
        if (bb <= 0) continue;
        for (int i = m->body_dofadr[bb] + m->body_dofnum[bb] - 1; i >= 0; i = m->dof_parentid[i])
            d->qfrc_passive[i] += d->cdof[i][0] * tq[0] + d->cdof[i][1] * tq[1] + d->cdof[i][2] * tq[2] + d->cdof[i][3] * gf[0] + d->cdof[i][4] * gf[1] +
                                  d->cdof[i][5] * gf[2];
    }
}

static void passive(const mjo_model *m, mjo_data *d) {
    for (int i = 0; i < m->nv; i++) d->qfrc_passive[i] = -m->dof_damping[i] * d->qvel[i];
    for (int j = 0; j < m->njnt; j++)
        if (m->jnt_type[j] == MJO_HINGE || m->jnt_type[j] == MJO_SLIDE) {
            int qa = m->jnt_qposadr[j];
            d->qfrc_passive[m->jnt_dofadr[j]] -= m->jnt_stiffness[j] * (d->qpos[qa] - m->qpos_spring[qa]);
        }
    fluid(m, d);
}

/* recursive Newton-Euler with qacc = 0: Coriolis, centrifugal and gravitational forces */
static void rne_bias(const mjo_model *m, mjo_data *d) {
    double cacc[MJO_MAXB][6], cfrc[MJO_MAXB][6];
    memset(cacc[0], 0, 48);
    for (int k = 0; k < 3; k++) cacc[0][3 + k] = -m->gravity[k];
    memset(cfrc[0], 0, 48);
    for (int b = 1; b < m->nbody; b++) {
        memcpy(cacc[b], cacc[m->body_parentid[b]], 48);
        for (int i = m->body_dofadr[b]; i < m->body_dofadr[b] + m->body_dofnum[b]; i++)
            for (int c = 0; c < 6; c++) cacc[b][c] += d->cdof_dot[i][c] * d->qvel[i];
        double Ia[6], Iv[6], vxIv[6];
        inert_mul(Ia, d->cinert[b], cacc[b]);
        inert_mul(Iv, d->cinert[b], d->cvel[b]);
        cross_force(vxIv, d->cvel[b], Iv);
        for (int c = 0; c < 6; c++) cfrc[b][c] = Ia[c] + vxIv[c];
    }
    for (int b = m->nbody - 1; b > 0; b--) {
        int p = m->body_parentid[b];
        if (p > 0)
            for (int c = 0; c < 6; c++) cfrc[p][c] += cfrc[b][c];
    }
    for (int i = 0; i < m->nv; i++) {
        double v = 0;
        for (int c = 0; c < 6; c++) v += d->cdof[i][c] * cfrc[m->dof_bodyid[i]][c];
        d->qfrc_bias[i] = v;
    }
}

static void actuation(const mjo_model *m, mjo_data *d) {
    memset(d->qfrc_actuator, 0, sizeof(double) * m->nv);
    for (int u = 0; u < m->nu; u++) {
        double c = d->ctrl[u], lo = m->actuator_ctrlrange[u][0], hi = m->actuator_ctrlrange[u][1];
        c = c < lo ? lo : (c > hi ? hi : c);
        d->qfrc_actuator[m->actuator_dofadr[u]] += m->actuator_gear[u] * c;
    }
}

/* ---- constraint solvers --------------------------------------------------------------------------------------- */
static void constraint_update(const mjo_model *m, mjo_data *d, const double *qacc) {
    /* forces for a given acceleration (all rows are unilateral: limit, frictionless contact, pyramid edge) */
    for (int i = 0; i < d->nefc; i++) {
        double jar = -d->efc_aref[i];
        for (int k = 0; k < m->nv; k++) jar += d->efc_J[i][k] * qacc[k];
        d->efc_force[i] = jar < 0 ? -d->efc_D[i] * jar : 0.0;
    }
    for (int k = 0; k < m->nv; k++) {
        double s = 0;
        for (int i = 0; i < d->nefc; i++) s += d->efc_J[i][k] * d->efc_force[i];
        d->qfrc_constraint[k] = s;
    }
}

static void solve_newton(const mjo_model *m, mjo_data *d) {
    int nv = m->nv, ne = d->nefc;
    double qacc[MJO_MAXV], grad[MJO_MAXV], dir[MJO_MAXV], jar[MJO_MAXEFC], jd[MJO_MAXEFC];
    static __thread double H[MJO_MAXV][MJO_MAXV], HL[MJO_MAXV][MJO_MAXV];
    /* warm start: whichever of qacc_warmstart / qacc_smooth has the lower cost */
    double cost_best = 0;
    for (int pass = 0; pass < 2; pass++) {
        const double *x = pass == 0 ? d->qacc_smooth : d->qacc_warmstart;
        double cost = 0;
        for (int i = 0; i < nv; i++) {
            double s = 0;
            for (int k = 0; k < nv; k++) s += d->qM[i][k] * (x[k] - d->qacc_smooth[k]);
            cost += 0.5 * s * (x[i] - d->qacc_smooth[i]);
        }
        for (int i = 0; i < ne; i++) {
            double r = -d->efc_aref[i];
            for (int k = 0; k < nv; k++) r += d->efc_J[i][k] * x[k];
            if (r < 0) cost += 0.5 * d->efc_D[i] * r * r;
        }
        if (pass == 0 || cost < cost_best) {
            cost_best = cost;
            memcpy(qacc, x, sizeof(double) * nv);
        }
    }
    const double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
    int it;
    for (it = 0; it < 100; it++) {
        for (int i = 0; i < ne; i++) {
            double r = -d->efc_aref[i];
            for (int k = 0; k < nv; k++) r += d->efc_J[i][k] * qacc[k];
            jar[i] = r;
        }
        double gn = 0;
        for (int i = 0; i < nv; i++) {
            double s = 0;
            for (int k = 0; k < nv; k++) s += d->qM[i][k] * (qacc[k] - d->qacc_smooth[k]);
            for (int r = 0; r < ne; r++)
                if (jar[r] < 0) s += d->efc_J[r][i] * d->efc_D[r] * jar[r];
            grad[i] = s, gn += s * s;
        }
        if (sqrt(gn) * scale < 1e-10) break; /* MuJoCo's own tolerance is 1e-8 on the same scaled quantity */
        for (int i = 0; i < nv; i++)
            for (int k = 0; k < nv; k++) {
                double s = d->qM[i][k];
                for (int r = 0; r < ne; r++)
                    if (jar[r] < 0) s += d->efc_J[r][i] * d->efc_D[r] * d->efc_J[r][k];
                H[i][k] = s;
            }
        chol_factor(nv, H, HL);
        for (int i = 0; i < nv; i++) dir[i] = -grad[i];
        chol_solve(nv, HL, dir);
        /* exact line search on the convex piecewise-quadratic phi(alpha) = cost(qacc + alpha dir) */
        double g0 = 0, h0 = 0;
        for (int i = 0; i < nv; i++) {
            double s = 0, t = 0;
            for (int k = 0; k < nv; k++) s += d->qM[i][k] * dir[k], t += d->qM[i][k] * (qacc[k] - d->qacc_smooth[k]);
            h0 += dir[i] * s, g0 += dir[i] * t;
        }
        for (int r = 0; r < ne; r++) {
            double s = 0;
            for (int k = 0; k < nv; k++) s += d->efc_J[r][k] * dir[k];
            jd[r] = s;
        }
        /* phi'(alpha) = c0 + c1 alpha on each segment between breakpoints t_r = -jar_r / jd_r (row r switches there) */
        double c0 = g0, c1 = h0, bp[MJO_MAXEFC];
        int order[MJO_MAXEFC], nbp = 0;
        for (int r = 0; r < ne; r++) {
            int active = jar[r] < 0 || (jar[r] == 0 && jd[r] < 0);
            if (active) c0 += d->efc_D[r] * jar[r] * jd[r], c1 += d->efc_D[r] * jd[r] * jd[r];
            if (jd[r] != 0) {
                double t = -jar[r] / jd[r];
                if (t > 0 && ((active && jd[r] > 0) || (!active && jd[r] < 0))) {
                    int k = nbp++;
                    while (k > 0 && bp[order[k - 1]] > t) order[k] = order[k - 1], k--;
                    order[k] = r, bp[r] = t;
                }
            }
        }
        double alpha = 0;
        for (int k = 0; k <= nbp; k++) {
            double hi = k < nbp ? bp[order[k]] : INFINITY;
            double root = c1 > MINVAL ? -c0 / c1 : hi;
            if (root <= hi) {
                alpha = root;
                break;
            }
            int r = order[k]; /* cross the breakpoint: toggle row r */
            double sg = jd[r] > 0 ? -1.0 : 1.0; /* active -> inactive when moving up out of violation, else inactive -> active */
            c0 += sg * d->efc_D[r] * jar[r] * jd[r], c1 += sg * d->efc_D[r] * jd[r] * jd[r];
            alpha = hi;
        }
        if (!(alpha > 0)) break; /* no descent possible: converged to rounding */
        double move = 0;
        for (int i = 0; i < nv; i++) qacc[i] += alpha * dir[i], move += fabs(alpha * dir[i]);
        if (move * scale < 1e-16) break;
    }
    d->solver_iter = it;
    memcpy(d->qacc, qacc, sizeof(double) * nv);
    constraint_update(m, d, qacc);
}

static void solve_pgs(const mjo_model *m, mjo_data *d) {
    int nv = m->nv, ne = d->nefc;
    static __thread double AR[MJO_MAXEFC][MJO_MAXEFC], MiJT[MJO_MAXEFC][MJO_MAXV];
    double b[MJO_MAXEFC], f[MJO_MAXEFC];
    for (int r = 0; r < ne; r++) {
        memcpy(MiJT[r], d->efc_J[r], sizeof(double) * nv);
        chol_solve(nv, d->qL, MiJT[r]);
    }
    for (int r = 0; r < ne; r++) {
        for (int c = 0; c < ne; c++) {
            double s = 0;
            for (int k = 0; k < nv; k++) s += d->efc_J[r][k] * MiJT[c][k];
            AR[r][c] = s;
        }
        AR[r][r] += d->efc_R[r];
        double s = -d->efc_aref[r];
        for (int k = 0; k < nv; k++) s += d->efc_J[r][k] * d->qacc_smooth[k];
        b[r] = s;
    }
    /* warm start from the previous acceleration; fall back to zero forces if that is not an improvement */
    constraint_update(m, d, d->qacc_warmstart);
    memcpy(f, d->efc_force, sizeof(double) * ne);
    double cost = 0;
    for (int r = 0; r < ne; r++) {
        double s = 0;
        for (int c = 0; c < ne; c++) s += AR[r][c] * f[c];
        cost += f[r] * (0.5 * s + b[r]);
    }
    if (cost > 0) memset(f, 0, sizeof(double) * ne);
    const double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
    int it;
    for (it = 0; it < m->iterations; it++) {
        double improvement = 0;
        for (int r = 0; r < ne; r++) {
            double res = b[r];
            for (int c = 0; c < ne; c++) res += AR[r][c] * f[c];
            double old = f[r], nw = old - res / AR[r][r];
            if (nw < 0) nw = 0;
            double delta = nw - old;
            f[r] = nw;
            improvement -= delta * (0.5 * delta * AR[r][r] + res);
        }
        if (improvement * scale < g_pgs_tolerance) {
            it++;
            break;
        }
    }
    d->solver_iter = it;
    memcpy(d->efc_force, f, sizeof(double) * ne);
    for (int k = 0; k < nv; k++) {
        double s = 0;
        for (int r = 0; r < ne; r++) s += d->efc_J[r][k] * f[r];
        d->qfrc_constraint[k] = s;
    }
    double t[MJO_MAXV];
    memcpy(t, d->qfrc_constraint, sizeof(double) * nv);
    chol_solve(nv, d->qL, t);
    for (int k = 0; k < nv; k++) d->qacc[k] = d->qacc_smooth[k] + t[k];
}

/* ---- forward -------------------------------------------------------------------------------------------------- */
/* fixed tendons: length = sum coef qpos[joint], velocity = sum coef qvel[joint] (mjWRAP_JOINT wraps of mj_tendon; ten_J qvel) */
static void tendons(const mjo_model *m, mjo_data *d) {
    for (int t = 0; t < m->ntendon; t++) {
        double l = 0, v = 0;
        for (int k = 0; k < m->tendon_num[t]; k++)
            l += m->wrap_coef[t][k] * d->qpos[m->wrap_qposadr[t][k]], v += m->wrap_coef[t][k] * d->qvel[m->wrap_dofadr[t][k]];
        d->ten_length[t] = l, d->ten_velocity[t] = v;
    }
}

void mjo_forward(const mjo_model *m, mjo_data *d) {
    int nv = m->nv;
    kinematics(m, d);
    tendons(m, d);
    com_pos(m, d);
    crb(m, d);
    chol_factor(nv, d->qM, d->qL);
    collision(m, d);
    make_constraint(m, d);
    com_vel(m, d);
    passive(m, d);
    rne_bias(m, d);
    actuation(m, d);
    for (int i = 0; i < nv; i++) d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i];
    memcpy(d->qacc_smooth, d->qfrc_smooth, sizeof(double) * nv);
    chol_solve(nv, d->qL, d->qacc_smooth);
    for (int r = 0; r < d->nefc; r++) {
        double v = 0;
        for (int k = 0; k < nv; k++) v += d->efc_J[r][k] * d->qvel[k];
        d->efc_vel[r] = v;
        d->efc_aref[r] = -d->efc_KBIP[r][1] * v - d->efc_KBIP[r][0] * d->efc_KBIP[r][2] * (d->efc_pos[r] - d->efc_margin[r]);
    }
    if (d->nefc == 0) {
        memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv);
        memset(d->qfrc_constraint, 0, sizeof(double) * nv);
        d->solver_iter = 0;
    } else if (m->solver == MJO_PGS) {
        solve_pgs(m, d);
    } else {
        solve_newton(m, d);
    }
    /* Newton (converged): the next pass starts from this one's solution.  PGS is truncated, so its iterates depend on the start: there
     * MuJoCo's rule is kept exactly -- qacc_warmstart is saved once per mj_step, after the integrator (mj_advance), and every forward
     * pass of an RK4 step (and a bare mj_forward) starts from that same vector. */
    if (m->solver != MJO_PGS) memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * nv);
}

/* ---- integration ---------------------------------------------------------------------------------------------- */
static void integrate_pos(const mjo_model *m, double *qpos, const double *qvel, double h) {
    for (int j = 0; j < m->njnt; j++) {
        int qa = m->jnt_qposadr[j], va = m->jnt_dofadr[j];
        if (m->jnt_type[j] == MJO_FREE) {
            for (int k = 0; k < 3; k++) qpos[qa + k] += h * qvel[va + k];
            double w[3] = {qvel[va + 3], qvel[va + 4], qvel[va + 5]}, qrot[4];
            double ang = h * normalize3(w);
            axis_angle_to_quat(qrot, w, ang);
            quat_normalize(qpos + qa + 3);
            quat_mul(qpos + qa + 3, qpos + qa + 3, qrot);
        } else {
            qpos[qa] += h * qvel[va];
        }
    }
}

static void euler(const mjo_model *m, mjo_data *d) {
    int nv = m->nv, damped = 0;
    double qacc[MJO_MAXV];
    for (int i = 0; i < nv; i++) damped |= m->dof_damping[i] > 0;
    if (damped) { /* implicit in the joint damping: (M + h D) qacc = qfrc_smooth + qfrc_constraint */
        static __thread double A[MJO_MAXV][MJO_MAXV], L[MJO_MAXV][MJO_MAXV];
        memcpy(A, d->qM, sizeof A);
        for (int i = 0; i < nv; i++) A[i][i] += m->timestep * m->dof_damping[i], qacc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
        chol_factor(nv, A, L);
        chol_solve(nv, L, qacc);
    } else {
        memcpy(qacc, d->qacc, sizeof(double) * nv);
    }
    for (int i = 0; i < nv; i++) d->qvel[i] += m->timestep * qacc[i];
    integrate_pos(m, d->qpos, d->qvel, m->timestep);
}

static void rk4(const mjo_model *m, mjo_data *d) {
    static const double A[3][3] = {{0.5, 0, 0}, {0, 0.5, 0}, {0, 0, 1.0}}, B[4] = {1.0 / 6, 1.0 / 3, 1.0 / 3, 1.0 / 6};
    int nv = m->nv, nq = m->nq;
    double h = m->timestep, q0[MJO_MAXQ], v0[MJO_MAXV], Fv[4][MJO_MAXV], Fa[4][MJO_MAXV], dv[MJO_MAXV], da[MJO_MAXV];
    memcpy(q0, d->qpos, sizeof(double) * nq), memcpy(v0, d->qvel, sizeof(double) * nv);
    memcpy(Fv[0], d->qvel, sizeof(double) * nv), memcpy(Fa[0], d->qacc, sizeof(double) * nv);
    for (int i = 1; i < 4; i++) {
        for (int k = 0; k < nv; k++) {
            dv[k] = da[k] = 0;
            for (int j = 0; j < i; j++) dv[k] += A[i - 1][j] * Fv[j][k], da[k] += A[i - 1][j] * Fa[j][k];
        }
        memcpy(d->qpos, q0, sizeof(double) * nq);
        integrate_pos(m, d->qpos, dv, h);
        for (int k = 0; k < nv; k++) d->qvel[k] = v0[k] + h * da[k];
        mjo_forward(m, d);
        memcpy(Fv[i], d->qvel, sizeof(double) * nv), memcpy(Fa[i], d->qacc, sizeof(double) * nv);
    }
    for (int k = 0; k < nv; k++) {
        dv[k] = da[k] = 0;
        for (int j = 0; j < 4; j++) dv[k] += B[j] * Fv[j][k], da[k] += B[j] * Fa[j][k];
    }
    memcpy(d->qpos, q0, sizeof(double) * nq);
    for (int k = 0; k < nv; k++) d->qvel[k] = v0[k] + h * da[k];
    integrate_pos(m, d->qpos, dv, h);
}

void mjo_step(const mjo_model *m, mjo_data *d, int nstep) {
    for (int s = 0; s < nstep; s++) {
        mjo_forward(m, d);
        if (m->integrator == MJO_RK4)
            rk4(m, d);
        else
            euler(m, d);
        if (m->solver == MJO_PGS) memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * m->nv); /* mj_advance: save qacc for the next step's warm start */
    }
}

/* external (contact) forces on every body as com-based spatial forces [torque; force], from the LAST forward pass */
void mjo_rne_post_constraint(const mjo_model *m, mjo_data *d) {
    memset(d->cfrc_ext, 0, sizeof d->cfrc_ext);
    for (int c = 0; c < d->ncon; c++) {
        const mjo_contact *con = &d->contact[c];
        if (con->efc_address < 0) continue;
        const double *f = d->efc_force + con->efc_address;
        double lf[3] = {0, 0, 0};
        if (con->dim == 1) {
            lf[0] = f[0];
        } else {
            lf[0] = f[0] + f[1] + f[2] + f[3];
            lf[1] = (f[0] - f[1]) * con->friction, lf[2] = (f[2] - f[3]) * con->friction;
        }
        double F[3];
        for (int k = 0; k < 3; k++) F[k] = con->frame[k] * lf[0] + con->frame[3 + k] * lf[1] + con->frame[6 + k] * lf[2];
        int bodies[2] = {m->geom_bodyid[con->geom1], m->geom_bodyid[con->geom2]};
        for (int s = 0; s < 2; s++) {
            int b = bodies[s];
            if (b <= 0) continue;
            double sign = s == 0 ? -1.0 : 1.0;
            const double *O = d->subtree_com[m->body_rootid[b]];
            double r[3] = {con->pos[0] - O[0], con->pos[1] - O[1], con->pos[2] - O[2]}, tq[3];
            cross3(tq, r, F);
            for (int k = 0; k < 3; k++) d->cfrc_ext[b][k] += sign * tq[k], d->cfrc_ext[b][3 + k] += sign * F[k];
        }
    }
}
