/*
 * mujoco_envs.c -- TEST INFRASTRUCTURE (CPU oracle).  The reference's Python glue around `mujoco` for
 * HalfCheetah-v5 / Ant-v5 / Humanoid-v5 restated in C on top of mujoco_core.c (PARITY UNPINNED for the physics, see
 * mujoco_core.h; the glue arithmetic -- NumPy reductions, float32 promotion, reset-noise streams -- is pinned on NumPy by
 * tests/test_mujoco_oracle.py).
 *
 *   step / reward / termination / obs / reset_model:
 *     gymnasium/envs/mujoco/half_cheetah_v5.py:220-281   ant_v5.py:327-428   humanoid_v5.py:413-541 (mass_center :17-21)
 *   do_simulation = ctrl copy -> mj_step(frame_skip) -> mj_rnePostConstraint:  mujoco_env.py:144-155,193-202
 *   reset = mj_resetData -> reset_model (noise) -> set_state -> mj_forward:     mujoco_env.py:132-142,172-187
 */
#include "mujoco_envs.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../gymnasium_amd/csrc/ziggurat_tables.h"

static const uint64_t KI[256] = {MI_ZIG_KI_VALUES};
static const double WI[256] = {MI_ZIG_WI_VALUES};
static const double FI[256] = {MI_ZIG_FI_VALUES};

/* numpy/random/src/distributions/distributions.c random_standard_normal (ziggurat, 256 layers) */
double orc_standard_normal(orc_pcg64 *rng) {
    for (;;) {
        uint64_t r = orc_pcg64_next64(rng);
        int idx = (int)(r & 0xff);
        r >>= 8;
        int sign = (int)(r & 0x1);
        uint64_t rabs = (r >> 1) & 0x000fffffffffffffULL;
        double x = (double)rabs * WI[idx];
        if (sign) x = -x;
        if (rabs < KI[idx]) return x;
        if (idx == 0) {
            for (;;) {
                double xx = -MI_ZIG_INV_R * log1p(-orc_pcg64_double(rng));
                double yy = -log1p(-orc_pcg64_double(rng));
                if (yy + yy > xx * xx) return ((rabs >> 8) & 0x1) ? -(MI_ZIG_R + xx) : MI_ZIG_R + xx;
            }
        } else {
            if ((FI[idx - 1] - FI[idx]) * orc_pcg64_double(rng) + FI[idx] < exp(-0.5 * x * x)) return x;
        }
    }
}

/* np.sum over a contiguous array = NumPy's pairwise_sum over all n elements (numpy/_core/src/umath/loops_utils.h.src:
 * plain loop below 8 elements, 8 interleaved accumulators up to 128, recursion above); order determined empirically
 * against NumPy 2.2 in tests/test_mujoco_oracle.py */
static double pairwise_f64(const double *a, int n) {
    if (n < 8) {
        double res = 0.;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        int i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_f64(a, n2) + pairwise_f64(a + n2, n - n2);
    }
}
double orc_np_sum_f64(const double *a, int n) { return pairwise_f64(a, n); }
/* np.linalg.norm of a short 1-D float64 vector (ord None or 2): numpy/linalg/_linalg.py norm() takes sqrt(x.dot(x)), and the BLAS dot of the
   reference environment (OpenBLAS, FMA kernels) accumulates x0 x0, then fma(x1, x1, acc), ... -- NOT the separately rounded sum of squares
   (they differ in the last bit for ~8 % of 2-vectors).  Pinned on NumPy by tests/test_mujoco_oracle.py::test_norm_matches_numpy; found by
   running the reference's own env classes against this glue (tests/test_mujoco_fixture_pipeline.py). */
double orc_np_norm(const double *v, int n) {
    double acc = v[0] * v[0];
    for (int k = 1; k < n; k++) acc = fma(v[k], v[k], acc);
    return sqrt(acc);
}
static float pairwise_f32(const float *a, int n) {
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    } else {
        float r[8];
        int i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
}
float orc_np_sum_f32(const float *a, int n) { return pairwise_f32(a, n); }

/* ---- model registry (filled by oracle/oracle.py at load time) -------------------------------------------------- */
static mjo_model *g_models[ORC_MJ_COUNT];
__attribute__((visibility("default"))) int orc_mj_register_model(int which, const double *blob, int n) {
    if (which < 0 || which >= ORC_MJ_COUNT) return -1;
    if (!g_models[which]) g_models[which] = (mjo_model *)malloc(sizeof(mjo_model));
    return mjo_model_from_blob(g_models[which], blob, n);
}
const mjo_model *orc_mj_model(int which) { return which >= 0 && which < ORC_MJ_COUNT ? g_models[which] : 0; }

/* ---- layout ----------------------------------------------------------------------------------------------------- */
static int is_planar_walker(int which) { return which == ORC_MJ_HOPPER || which == ORC_MJ_WALKER2D; }
static int is_humanoid(int which) { return which == ORC_MJ_HUMANOID || which == ORC_MJ_HUMANOID_STANDUP; }
static int is_pendulum(int which) { return which == ORC_MJ_INVERTED_PENDULUM || which == ORC_MJ_INVERTED_DOUBLE_PENDULUM; }

int orc_mjenv_obs_dim(int which, const double *P) {
    const mjo_model *m = g_models[which];
    int excl = P[3] != 0.0;
    if (which == ORC_MJ_INVERTED_PENDULUM) return m->nq + m->nv;
    if (which == ORC_MJ_INVERTED_DOUBLE_PENDULUM) return 1 + 2 * (m->nq - 1) + m->nv + 1;
    if (which == ORC_MJ_REACHER) return 10;
    if (which == ORC_MJ_PUSHER) return 23;
    if (is_planar_walker(which)) return m->nq - (excl ? 1 : 0) + m->nv;
    if (which == ORC_MJ_HALF_CHEETAH) return m->nq - (excl ? 1 : 0) + m->nv;
    if (which == ORC_MJ_SWIMMER) return m->nq - (excl ? 2 : 0) + m->nv; /* swimmer_v5.py:206-213 */
    if (which == ORC_MJ_ANT) return m->nq - (excl ? 2 : 0) + m->nv + (P[12] != 0.0 ? 6 * (m->nbody - 1) : 0);
    return m->nq - (excl ? 2 : 0) + m->nv + (P[12] != 0.0 ? 10 * (m->nbody - 1) : 0) + (P[13] != 0.0 ? 6 * (m->nbody - 1) : 0) +
           (P[14] != 0.0 ? m->nv - 6 : 0) + (P[15] != 0.0 ? 6 * (m->nbody - 1) : 0);
}
int orc_mjenv_info_dim(int which) {
    if (is_planar_walker(which)) return 6;
    if (which == ORC_MJ_INVERTED_PENDULUM) return 1;
    if (which == ORC_MJ_INVERTED_DOUBLE_PENDULUM) return 3;
    if (which == ORC_MJ_REACHER) return 2;
    if (which == ORC_MJ_HUMANOID_STANDUP) return 6 + 2 * g_models[which]->ntendon; /* + tendon_length, tendon_velocity (humanoidstandup_v5.py:433-434) */
    if (which == ORC_MJ_SWIMMER) return 7;
    if (which == ORC_MJ_PUSHER) return 3;
    if (which == ORC_MJ_HUMANOID) return 9 + 2 * g_models[which]->ntendon; /* humanoid_v5.py:486-487 */
    return which == ORC_MJ_HALF_CHEETAH ? 4 : 9;
}
/* the tendon columns of an info row: data.ten_length, data.ten_velocity of the last forward pass */
static void tendon_info(const orc_mjenv *e, double *cols) {
    const int nt = e->m->ntendon;
    for (int t = 0; t < nt; t++) cols[t] = e->d.ten_length[t], cols[nt + t] = e->d.ten_velocity[t];
}
int orc_mjenv_state_dim(int which) { return g_models[which]->nq + 2 * g_models[which]->nv + 2; }

/* ---- per-env glue ----------------------------------------------------------------------------------------------- */
orc_mjenv *orc_mjenv_create(int which, int force_newton) {
    orc_mjenv *e = (orc_mjenv *)calloc(1, sizeof *e);
    e->which = which, e->m = g_models[which];
    if (force_newton && g_models[which]->solver != MJO_NEWTON) { /* the humanoid envs' solver="Newton" opt-in */
        e->own = (mjo_model *)malloc(sizeof(mjo_model));
        memcpy(e->own, g_models[which], sizeof(mjo_model));
        e->own->solver = MJO_NEWTON, e->m = e->own;
    }
    mjo_reset_data(e->m, &e->d);
    return e;
}

static void mass_center_xy(const orc_mjenv *e, double out[2]) { /* humanoid_v5.py:17-21 (einsum then divide) */
    const mjo_model *m = e->m;
    double num[2] = {0, 0}, den = 0;
    for (int b = 0; b < m->nbody; b++) num[0] += m->body_mass[b] * e->d.xipos[b][0], num[1] += m->body_mass[b] * e->d.xipos[b][1];
    den = orc_np_sum_f64(m->body_mass, m->nbody);
    out[0] = num[0] / den, out[1] = num[1] / den;
}

/* the position the env differentiates to get its velocity reward, read from the LAST forward pass (the reference reads
 * data.qpos / data.body().xpos / data.xipos after mj_step, and those Cartesian quantities lag qpos by one sub-step) */
static void tracked_xy(const orc_mjenv *e, double out[2]) {
    if (e->which == ORC_MJ_REACHER || e->which == ORC_MJ_HUMANOID_STANDUP || e->which == ORC_MJ_PUSHER) {
        out[0] = out[1] = 0;
    } else if (e->which == ORC_MJ_INVERTED_DOUBLE_PENDULUM) { /* data.site_xpos[0]: x and height of the tip (inverted_double_pendulum_v5.py:189) */
        const double tip[3] = {0, 0, 0.6}; /* <site name="tip" pos="0 0 .6"/> on pole2 = the last body */
        const int b = e->m->nbody - 1;
        const double *R = e->d.xmat[b];
        out[0] = e->d.xpos[b][0] + R[0] * tip[0] + R[1] * tip[1] + R[2] * tip[2];
        out[1] = e->d.xpos[b][2] + R[6] * tip[0] + R[7] * tip[1] + R[8] * tip[2];
    } else if (e->which == ORC_MJ_HALF_CHEETAH || is_planar_walker(e->which) || e->which == ORC_MJ_INVERTED_PENDULUM)
        out[0] = e->d.qpos[0], out[1] = 0;
    else if (e->which == ORC_MJ_SWIMMER)
        out[0] = e->d.qpos[0], out[1] = e->d.qpos[1]; /* data.qpos[0:2] (swimmer_v5.py:226-228) */
    else if (e->which == ORC_MJ_ANT)
        out[0] = e->d.xpos[1][0], out[1] = e->d.xpos[1][1]; /* main_body = 1 (torso) */
    else
        mass_center_xy(e, out);
}

void orc_mjenv_obs(const orc_mjenv *e, const double *P, double *o) {
    const mjo_model *m = e->m;
    const mjo_data *d = &e->d;
    int skip = P[3] != 0.0 ? ((e->which == ORC_MJ_HALF_CHEETAH || is_planar_walker(e->which)) ? 1 : 2) : 0, n = 0;
    if (e->which == ORC_MJ_PUSHER) { /* pusher_v5.py:317-326; get_body_com = data.body(name).xpos of tips_arm, object, goal (the last three bodies) */
        for (int k = 0; k < 7; k++) o[k] = d->qpos[k], o[7 + k] = d->qvel[k];
        for (int b = 0; b < 3; b++)
            for (int k = 0; k < 3; k++) o[14 + 3 * b + k] = d->xpos[m->nbody - 3 + b][k];
        return;
    }
    if (e->which == ORC_MJ_REACHER) { /* reacher_v5.py:232-245; get_body_com = data.body(name).xpos: fingertip = body 3, target = body 4 */
        o[0] = cos(d->qpos[0]), o[1] = cos(d->qpos[1]), o[2] = sin(d->qpos[0]), o[3] = sin(d->qpos[1]);
        o[4] = d->qpos[2], o[5] = d->qpos[3], o[6] = d->qvel[0], o[7] = d->qvel[1];
        o[8] = d->xpos[3][0] - d->xpos[4][0], o[9] = d->xpos[3][1] - d->xpos[4][1];
        return;
    }
    if (e->which == ORC_MJ_INVERTED_DOUBLE_PENDULUM) { /* inverted_double_pendulum_v5.py:217-226 */
        o[n++] = d->qpos[0];
        for (int k = 1; k < m->nq; k++) o[n++] = sin(d->qpos[k]);
        for (int k = 1; k < m->nq; k++) o[n++] = cos(d->qpos[k]);
        for (int k = 0; k < m->nv; k++) o[n++] = d->qvel[k] < -10.0 ? -10.0 : (d->qvel[k] > 10.0 ? 10.0 : d->qvel[k]);
        const double f = d->qfrc_constraint[0];
        o[n++] = f < -10.0 ? -10.0 : (f > 10.0 ? 10.0 : f);
        return;
    }
    if (is_pendulum(e->which)) skip = 0;
    for (int k = skip; k < m->nq; k++) o[n++] = d->qpos[k];
    for (int k = 0; k < m->nv; k++) {
        const double v = d->qvel[k];
        o[n++] = is_planar_walker(e->which) ? (v < -10.0 ? -10.0 : (v > 10.0 ? 10.0 : v)) : v; /* np.clip(qvel, -10, 10) hopper_v5.py:262 */
    }
    if (e->which == ORC_MJ_ANT && P[12] != 0.0) {
        for (int b = 1; b < m->nbody; b++)
            for (int k = 0; k < 6; k++) {
                double f = d->cfrc_ext[b][k];
                o[n++] = f < P[10] ? P[10] : (f > P[11] ? P[11] : f); /* np.clip(cfrc_ext, lo, hi) ant_v5.py:327-332 */
            }
    } else if (is_humanoid(e->which)) {
        if (P[12] != 0.0)
            for (int b = 1; b < m->nbody; b++)
                for (int k = 0; k < 10; k++) o[n++] = d->cinert[b][k];
        if (P[13] != 0.0)
            for (int b = 1; b < m->nbody; b++)
                for (int k = 0; k < 6; k++) o[n++] = d->cvel[b][k];
        if (P[14] != 0.0)
            for (int k = 6; k < m->nv; k++) o[n++] = d->qfrc_actuator[k];
        if (P[15] != 0.0)
            for (int b = 1; b < m->nbody; b++)
                for (int k = 0; k < 6; k++) o[n++] = d->cfrc_ext[b][k];
    }
}

void orc_mjenv_reset(orc_mjenv *e, orc_pcg64 *rng, const double *P) {
    const mjo_model *m = e->m;
    double scale = P[2], qpos[MJO_MAXQ], qvel[MJO_MAXV];
    mjo_reset_data(m, &e->d); /* mj_resetData */
    if (e->which == ORC_MJ_PUSHER) { /* pusher_v5.py:293-315 */
        for (int k = 0; k < m->nq; k++) qpos[k] = m->qpos0[k];
        for (;;) { /* the object is re-drawn until it is farther than 0.17 from the goal at the origin of its sliders (y slider first) */
            const double c0 = -0.3 + (0.0 - (-0.3)) * orc_pcg64_double(rng), c1 = -0.2 + (0.2 - (-0.2)) * orc_pcg64_double(rng);
            qpos[m->nq - 4] = c0, qpos[m->nq - 3] = c1;
            const double cg[2] = {c0, c1};
            if (orc_np_norm(cg, 2) > 0.17) break;
        }
        qpos[m->nq - 2] = 0.0, qpos[m->nq - 1] = 0.0;
        for (int k = 0; k < m->nv; k++) qvel[k] = 0.0 + (-0.005 + (0.005 - (-0.005)) * orc_pcg64_double(rng));
        for (int k = m->nv - 4; k < m->nv; k++) qvel[k] = 0.0;
        memcpy(e->d.qpos, qpos, sizeof(double) * m->nq), memcpy(e->d.qvel, qvel, sizeof(double) * m->nv);
        mjo_forward(m, &e->d);
        e->has_override = 0;
        return;
    }
    if (e->which == ORC_MJ_REACHER) { /* reacher_v5.py:209-226 */
        for (int k = 0; k < m->nq; k++) qpos[k] = (-0.1 + (0.1 - (-0.1)) * orc_pcg64_double(rng)) + m->qpos0[k];
        for (;;) { /* the goal is re-drawn until it lies inside the 0.2 disc (np.linalg.norm of 2 elements = sqrt(g0 g0 + g1 g1)) */
            const double g0 = -0.2 + (0.2 - (-0.2)) * orc_pcg64_double(rng), g1 = -0.2 + (0.2 - (-0.2)) * orc_pcg64_double(rng);
            qpos[2] = g0, qpos[3] = g1;
            const double gg[2] = {g0, g1};
            if (orc_np_norm(gg, 2) < 0.2) break;
        }
        for (int k = 0; k < m->nv; k++) qvel[k] = 0.0 + (-0.005 + (0.005 - (-0.005)) * orc_pcg64_double(rng));
        qvel[2] = 0.0, qvel[3] = 0.0;
        memcpy(e->d.qpos, qpos, sizeof(double) * m->nq), memcpy(e->d.qvel, qvel, sizeof(double) * m->nv);
        mjo_forward(m, &e->d);
        e->has_override = 0;
        return;
    }
    /* qpos = init_qpos + uniform(-s, s, nq): Generator.uniform = low + (high - low) * next_double */
    for (int k = 0; k < m->nq; k++) qpos[k] = m->qpos0[k] + (-scale + (scale - (-scale)) * orc_pcg64_double(rng));
    if (is_humanoid(e->which) || is_planar_walker(e->which) || e->which == ORC_MJ_INVERTED_PENDULUM || e->which == ORC_MJ_SWIMMER) /* swimmer_v5.py:279-294,  humanoid_v5.py:526-528, hopper_v5.py:318-331, inverted_pendulum_v5.py:178-190: uniform noise on the velocities too */
        for (int k = 0; k < m->nv; k++) qvel[k] = 0.0 + (-scale + (scale - (-scale)) * orc_pcg64_double(rng));
    else /* init_qvel + scale * standard_normal(nv) */
        for (int k = 0; k < m->nv; k++) qvel[k] = 0.0 + scale * orc_standard_normal(rng);
    memcpy(e->d.qpos, qpos, sizeof(double) * m->nq), memcpy(e->d.qvel, qvel, sizeof(double) * m->nv);
    mjo_forward(m, &e->d); /* set_state -> mj_forward */
    e->has_override = 0;
    /* cfrc_ext stays zero after mj_resetData until the first mj_rnePostConstraint (the reset observation shows zeros) */
}

/* weight * np.sum(np.square(action)) in the action's OWN dtype (half_cheetah_v5.py:216-218 control_cost and its siblings): NEP 50 keeps a
 * float32 array's reduction and its product with the Python-float weight in float32; a float64 action row makes all of it float64.  The
 * value is returned widened to double (exact).  -np.square(action).sum() * w (reacher_v5.py:201, pusher_v5.py:281) is its negation bit for bit. */
static double control_cost(const double *a, int nu, int act_f64, double weight) {
    if (act_f64) {
        double sq[MJO_MAXU];
        for (int u = 0; u < nu; u++) sq[u] = a[u] * a[u];
        return weight * orc_np_sum_f64(sq, nu);
    }
    float sq[MJO_MAXU];
    for (int u = 0; u < nu; u++) sq[u] = (float)a[u] * (float)a[u];
    return (double)((float)weight * orc_np_sum_f32(sq, nu));
}

void orc_mjenv_step(orc_mjenv *e, const void *action_row, int act_f64, const double *P, double *reward, int *terminated, double *info) {
    double action[MJO_MAXU]; /* the row widened to double: exact for float32 values */
    for (int u = 0; u < e->m->nu; u++) action[u] = act_f64 ? ((const double *)action_row)[u] : (double)((const float *)action_row)[u];
    const mjo_model *m = e->m;
    mjo_data *d = &e->d;
    const int nu = m->nu, frame_skip = (int)P[4];
    double before[2], after[2];
    if (e->has_override)
        before[0] = e->track_override[0], before[1] = e->track_override[1], e->has_override = 0;
    else
        tracked_xy(e, before);
    for (int u = 0; u < nu; u++) d->ctrl[u] = action[u];
    mjo_step(m, d, frame_skip);
    mjo_rne_post_constraint(m, d);
    tracked_xy(e, after);
    const double dt = m->timestep * frame_skip;
    const double xv = (after[0] - before[0]) / dt, yv = (after[1] - before[1]) / dt;
    const double forward_reward = e->which == ORC_MJ_ANT ? xv * P[0] : P[0] * xv;
    if (e->which == ORC_MJ_HUMANOID_STANDUP) { /* humanoidstandup_v5.py:423-462 (uph_cost_weight is never applied there) */
        const double uph_cost = (d->qpos[2] - 0) / m->timestep;
        double sq[MJO_MAXU], c2[6 * MJO_MAXB];
        for (int u = 0; u < nu; u++) sq[u] = d->ctrl[u] * d->ctrl[u];
        const double quad_ctrl_cost = P[1] * orc_np_sum_f64(sq, nu);
        for (int b = 0; b < m->nbody; b++)
            for (int k = 0; k < 6; k++) c2[6 * b + k] = d->cfrc_ext[b][k] * d->cfrc_ext[b][k];
        double quad_impact_cost = P[5] * orc_np_sum_f64(c2, 6 * m->nbody);
        quad_impact_cost = quad_impact_cost < P[10] ? P[10] : (quad_impact_cost > P[11] ? P[11] : quad_impact_cost);
        *reward = uph_cost - quad_ctrl_cost - quad_impact_cost + 1;
        *terminated = 0;
        info[0] = d->qpos[0], info[1] = d->qpos[1], info[2] = d->qpos[2] - m->qpos0[2], info[3] = uph_cost, info[4] = -quad_ctrl_cost,
        info[5] = -quad_impact_cost;
        tendon_info(e, info + 6);
        return;
    }
    if (e->which == ORC_MJ_PUSHER) { /* pusher_v5.py:266-291 */
        const double *tip = d->xpos[m->nbody - 3], *ob = d->xpos[m->nbody - 2], *goal = d->xpos[m->nbody - 1];
        const double v1[3] = {ob[0] - tip[0], ob[1] - tip[1], ob[2] - tip[2]}, v2[3] = {ob[0] - goal[0], ob[1] - goal[1], ob[2] - goal[2]};
        const double reward_near = -orc_np_norm(v1, 3) * P[0];
        const double reward_dist = -orc_np_norm(v2, 3) * P[5];
        const double reward_ctrl = -control_cost(action, nu, act_f64, P[1]); /* a float32 row: float32 sum times a Python float -> float32 */
        *reward = (reward_dist + (double)reward_ctrl) + reward_near;
        *terminated = 0;
        info[0] = reward_dist, info[1] = (double)reward_ctrl, info[2] = reward_near;
        return;
    }
    if (e->which == ORC_MJ_REACHER) { /* reacher_v5.py:188-207 */
        const double v[3] = {d->xpos[3][0] - d->xpos[4][0], d->xpos[3][1] - d->xpos[4][1], d->xpos[3][2] - d->xpos[4][2]};
        const double reward_dist = -orc_np_norm(v, 3) * P[0];
        const double reward_ctrl = -control_cost(action, nu, act_f64, P[1]); /* a float32 row: float32 sum times a Python float -> float32 */
        *reward = reward_dist + (double)reward_ctrl;
        *terminated = 0;
        info[0] = reward_dist, info[1] = (double)reward_ctrl;
        return;
    }
    if (e->which == ORC_MJ_INVERTED_PENDULUM) { /* inverted_pendulum_v5.py:160-176 */
        int finite = 1;
        for (int k = 0; k < m->nq; k++) finite &= isfinite(d->qpos[k]) != 0;
        for (int k = 0; k < m->nv; k++) finite &= isfinite(d->qvel[k]) != 0;
        *terminated = !finite || fabs(d->qpos[1]) > 0.2;
        *reward = *terminated ? 0.0 : 1.0;
        info[0] = *reward;
        return;
    }
    if (e->which == ORC_MJ_INVERTED_DOUBLE_PENDULUM) { /* inverted_double_pendulum_v5.py:186-215 */
        const double x = after[0], y = after[1], v1 = d->qvel[1], v2 = d->qvel[2];
        *terminated = y <= 1.0;
        /* x, y, v1, v2 are np.float64 SCALARS there: `** 2` is NumPy's scalar power = libm pow(x, 2.0), which is not the correctly rounded x * x for
         * ~0.09 % of arguments (the same fact as pendulum.py:131, oracle/classic_control.c pendulum_step) */
        const double dist_penalty = 0.01 * pow(x, 2.0) + pow(y - 2, 2.0), vel_penalty = 1e-3 * pow(v1, 2.0) + 5e-3 * pow(v2, 2.0);
        const double alive_bonus = P[6] * (*terminated ? 0.0 : 1.0);
        *reward = alive_bonus - dist_penalty - vel_penalty;
        info[0] = alive_bonus, info[1] = -dist_penalty, info[2] = -vel_penalty;
        return;
    }
    if (is_planar_walker(e->which)) { /* hopper_v5.py:240-309, walker2d_v5.py:245-311 */
        const double ctrl_cost = control_cost(action, nu, act_f64, P[1]);
        const double z = d->qpos[1], angle = d->qpos[2];
        int healthy = P[8] < z && z < P[9] && P[10] < angle && angle < P[11];
        if (e->which == ORC_MJ_HOPPER) { /* healthy_state_range on state_vector()[2:] */
            for (int k = 2; k < m->nq; k++) healthy = healthy && P[12] < d->qpos[k] && d->qpos[k] < P[13];
            for (int k = 0; k < m->nv; k++) healthy = healthy && P[12] < d->qvel[k] && d->qvel[k] < P[13];
        }
        const double healthy_reward = healthy ? P[6] : 0.0;
        *reward = (forward_reward + healthy_reward) - (double)ctrl_cost;
        *terminated = !healthy && P[7] != 0.0;
        info[0] = d->qpos[0], info[1] = d->qpos[1] - m->qpos0[1], info[2] = xv, info[3] = forward_reward, info[4] = -(double)ctrl_cost,
        info[5] = healthy_reward;
        return;
    }
    if (e->which == ORC_MJ_SWIMMER) { /* swimmer_v5.py:225-263: never terminates; the control cost is float32 like HalfCheetah's */
        const double ctrl_cost = control_cost(action, nu, act_f64, P[1]);
        *reward = forward_reward - (double)ctrl_cost;
        *terminated = 0;
        info[0] = after[0], info[1] = after[1], info[2] = orc_np_norm(after, 2), info[3] = xv, info[4] = yv;
        info[5] = forward_reward, info[6] = -(double)ctrl_cost;
        return;
    }
    if (e->which == ORC_MJ_HALF_CHEETAH) {
        /* control_cost: weight * np.sum(np.square(action)) with a float32 action -> float32 (NEP 50) */
        const double ctrl_cost = control_cost(action, nu, act_f64, P[1]);
        *reward = forward_reward - (double)ctrl_cost;
        *terminated = 0;
        info[0] = d->qpos[0], info[1] = xv, info[2] = forward_reward, info[3] = -(double)ctrl_cost;
        return;
    }
    int healthy;
    double ctrl_cost, contact_cost;
    if (e->which == ORC_MJ_ANT) {
        int finite = 1;
        for (int k = 0; k < m->nq; k++) finite &= isfinite(d->qpos[k]) != 0;
        for (int k = 0; k < m->nv; k++) finite &= isfinite(d->qvel[k]) != 0;
        healthy = finite && P[8] <= d->qpos[2] && d->qpos[2] <= P[9];
        ctrl_cost = control_cost(action, nu, act_f64, P[1]);
        double c2[6 * MJO_MAXB];
        for (int b = 0; b < m->nbody; b++)
            for (int k = 0; k < 6; k++) {
                double f = d->cfrc_ext[b][k];
                f = f < P[10] ? P[10] : (f > P[11] ? P[11] : f);
                c2[6 * b + k] = f * f;
            }
        contact_cost = P[5] * orc_np_sum_f64(c2, 6 * m->nbody);
    } else {
        healthy = P[8] < d->qpos[2] && d->qpos[2] < P[9];
        double sq[MJO_MAXU], c2[6 * MJO_MAXB];
        for (int u = 0; u < nu; u++) sq[u] = d->ctrl[u] * d->ctrl[u]; /* np.square(self.data.ctrl): float64 */
        ctrl_cost = P[1] * orc_np_sum_f64(sq, nu);
        for (int b = 0; b < m->nbody; b++)
            for (int k = 0; k < 6; k++) c2[6 * b + k] = d->cfrc_ext[b][k] * d->cfrc_ext[b][k];
        contact_cost = P[5] * orc_np_sum_f64(c2, 6 * m->nbody);
        contact_cost = contact_cost < P[10] ? P[10] : (contact_cost > P[11] ? P[11] : contact_cost);
    }
    const double healthy_reward = healthy ? P[6] : 0.0; /* is_healthy * healthy_reward */
    const double rewards = forward_reward + healthy_reward, costs = ctrl_cost + contact_cost;
    *reward = rewards - costs;
    *terminated = !healthy && P[7] != 0.0;
    info[0] = d->qpos[0], info[1] = d->qpos[1], info[2] = orc_np_norm(d->qpos, 2);
    info[3] = xv, info[4] = yv, info[5] = forward_reward, info[6] = -ctrl_cost, info[7] = -contact_cost, info[8] = healthy_reward;
    if (e->which == ORC_MJ_HUMANOID) tendon_info(e, info + 9);
}

/* _get_reset_info of the scalar env: positions only */
void orc_mjenv_reset_info(const orc_mjenv *e, double *row) {
    if (is_pendulum(e->which) || e->which == ORC_MJ_REACHER || e->which == ORC_MJ_PUSHER) return; /* {} (inverted_pendulum_v5.py:198-199) */
    row[0] = e->d.qpos[0];
    if (is_humanoid(e->which)) tendon_info(e, row + (e->which == ORC_MJ_HUMANOID ? 9 : 6)); /* humanoid_v5.py:534-541 */
    if (e->which == ORC_MJ_HUMANOID_STANDUP)
        row[1] = e->d.qpos[1], row[2] = e->d.qpos[2] - e->m->qpos0[2]; /* humanoidstandup_v5.py:479-486 */
    else if (is_planar_walker(e->which))
        row[1] = e->d.qpos[1] - e->m->qpos0[1]; /* z_distance_from_origin, hopper_v5.py:338-342 */
    else if (e->which != ORC_MJ_HALF_CHEETAH)
        row[1] = e->d.qpos[1], row[2] = orc_np_norm(row, 2);
}

/* checkpoint row: qpos, qvel, qacc_warmstart, tracked xy of the last forward pass */
void orc_mjenv_get_state(const orc_mjenv *e, double *s) {
    const mjo_model *m = e->m;
    memcpy(s, e->d.qpos, sizeof(double) * m->nq), memcpy(s + m->nq, e->d.qvel, sizeof(double) * m->nv);
    memcpy(s + m->nq + m->nv, e->d.qacc_warmstart, sizeof(double) * m->nv);
    if (e->has_override)
        s[m->nq + 2 * m->nv] = e->track_override[0], s[m->nq + 2 * m->nv + 1] = e->track_override[1];
    else
        tracked_xy(e, s + m->nq + 2 * m->nv);
}
void orc_mjenv_set_state(orc_mjenv *e, const double *s) {
    const mjo_model *m = e->m;
    memcpy(e->d.qpos, s, sizeof(double) * m->nq), memcpy(e->d.qvel, s + m->nq, sizeof(double) * m->nv);
    mjo_forward(m, &e->d);
    memcpy(e->d.qacc_warmstart, s + m->nq + m->nv, sizeof(double) * m->nv);
    /* the Cartesian position the next step differentiates against is the checkpointed (lagging) one, not the fresh one */
    e->track_override[0] = s[m->nq + 2 * m->nv], e->track_override[1] = s[m->nq + 2 * m->nv + 1], e->has_override = 1;
}
