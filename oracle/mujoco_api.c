/*
 * mujoco_api.c -- TEST INFRASTRUCTURE: flat C entry points of the MuJoCo-pipeline oracle for ctypes (oracle/mujoco.py).
 * PARITY UNPINNED (see mujoco_core.h).
 */
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include "mujoco_core.h"

#define API __attribute__((visibility("default")))

API mjo_model *orc_mj_model_create(const double *blob, int n) {
    mjo_model *m = (mjo_model *)malloc(sizeof(mjo_model));
    if (!m) return 0;
    if (mjo_model_from_blob(m, blob, n) != 0) {
        free(m);
        return 0;
    }
    return m;
}
API void orc_mj_model_destroy(mjo_model *m) { free(m); }
API mjo_data *orc_mj_data_create(const mjo_model *m) {
    mjo_data *d = (mjo_data *)malloc(sizeof(mjo_data));
    if (d) mjo_reset_data(m, d);
    return d;
}
API void orc_mj_data_destroy(mjo_data *d) { free(d); }
API void orc_mj_reset(const mjo_model *m, mjo_data *d) { mjo_reset_data(m, d); }
API void orc_mj_set_state(const mjo_model *m, mjo_data *d, const double *qpos, const double *qvel, const double *ctrl) {
    if (qpos) memcpy(d->qpos, qpos, sizeof(double) * m->nq);
    if (qvel) memcpy(d->qvel, qvel, sizeof(double) * m->nv);
    if (ctrl) memcpy(d->ctrl, ctrl, sizeof(double) * m->nu);
}
API void orc_mj_forward(const mjo_model *m, mjo_data *d) { mjo_forward(m, d); }
API void orc_mj_step(const mjo_model *m, mjo_data *d, int nstep) { mjo_step(m, d, nstep); }
API void orc_mj_rne_post_constraint(const mjo_model *m, mjo_data *d) { mjo_rne_post_constraint(m, d); }

typedef struct {
    const char *name;
    size_t offset;
    int rows_kind; /* 0: fixed count, 1: nq, 2: nv, 3: nbody, 4: nu, 5: nefc, 6: njnt, 7: ngeom */
    int rows, cols, stride;
} field_t;
#define F(nm, kind, rows, cols, stride) {#nm, offsetof(mjo_data, nm), kind, rows, cols, stride}
static const field_t FIELDS[] = {
    F(qpos, 1, 0, 1, 1), F(qvel, 2, 0, 1, 1), F(ctrl, 4, 0, 1, 1), F(xpos, 3, 0, 3, 3), F(xquat, 3, 0, 4, 4), F(xmat, 3, 0, 9, 9),
    F(xipos, 3, 0, 3, 3), F(xanchor, 6, 0, 3, 3), F(xaxis, 6, 0, 3, 3), F(geom_xpos, 7, 0, 3, 3), F(geom_xmat, 7, 0, 9, 9),
    F(subtree_com, 3, 0, 3, 3), F(cinert, 3, 0, 10, 10), F(cdof, 2, 0, 6, 6), F(cvel, 3, 0, 6, 6), F(cdof_dot, 2, 0, 6, 6),
    F(qM, 2, 0, -2, MJO_MAXV), F(qfrc_passive, 2, 0, 1, 1), F(qfrc_bias, 2, 0, 1, 1), F(qfrc_actuator, 2, 0, 1, 1),
    F(qfrc_smooth, 2, 0, 1, 1), F(qacc_smooth, 2, 0, 1, 1), F(qfrc_constraint, 2, 0, 1, 1), F(qacc, 2, 0, 1, 1),
    F(qacc_warmstart, 2, 0, 1, 1), F(cfrc_ext, 3, 0, 6, 6), F(efc_J, 5, 0, -2, MJO_MAXV), F(efc_pos, 5, 0, 1, 1),
    F(efc_margin, 5, 0, 1, 1), F(efc_D, 5, 0, 1, 1), F(efc_R, 5, 0, 1, 1), F(efc_vel, 5, 0, 1, 1), F(efc_aref, 5, 0, 1, 1),
    F(efc_force, 5, 0, 1, 1), F(efc_KBIP, 5, 0, 4, 4), F(ten_length, 0, MJO_MAXT, 1, 1), F(ten_velocity, 0, MJO_MAXT, 1, 1),
};

/* Copies field `name` as a dense row-major [rows][cols] array into out; returns rows*cols, or -1 for an unknown name. */
API int orc_mj_get(const mjo_model *m, const mjo_data *d, const char *name, double *out, int max) {
    if (!strcmp(name, "ncon")) { out[0] = d->ncon; return 1; }
    if (!strcmp(name, "nefc")) { out[0] = d->nefc; return 1; }
    if (!strcmp(name, "solver_iter")) { out[0] = d->solver_iter; return 1; }
    if (!strcmp(name, "contact")) { /* rows: dist, pos3, frame9, geom1, geom2, dim, efc_address */
        int w = 17;
        if (d->ncon * w > max) return -2;
        for (int c = 0; c < d->ncon; c++) {
            const mjo_contact *k = &d->contact[c];
            double *o = out + c * w;
            o[0] = k->dist, memcpy(o + 1, k->pos, 24), memcpy(o + 4, k->frame, 72);
            o[13] = k->geom1, o[14] = k->geom2, o[15] = k->dim, o[16] = k->efc_address;
        }
        return d->ncon * w;
    }
    for (size_t f = 0; f < sizeof FIELDS / sizeof FIELDS[0]; f++) {
        if (strcmp(name, FIELDS[f].name)) continue;
        int counts[8] = {FIELDS[f].rows, m->nq, m->nv, m->nbody, m->nu, d->nefc, m->njnt, m->ngeom};
        int rows = counts[FIELDS[f].rows_kind], cols = FIELDS[f].cols == -2 ? m->nv : FIELDS[f].cols;
        if (rows * cols > max) return -2;
        const double *src = (const double *)((const char *)d + FIELDS[f].offset);
        for (int r = 0; r < rows; r++) memcpy(out + r * cols, src + r * FIELDS[f].stride, sizeof(double) * cols);
        return rows * cols;
    }
    return -1;
}

/* test hooks for the glue arithmetic (pinned on NumPy by tests/test_mujoco_oracle.py) */
#include "mujoco_envs.h"
API double orc_test_np_sum_f64(const double *a, int n) { return orc_np_sum_f64(a, n); }
API double orc_test_np_norm(const double *a, int n) { return orc_np_norm(a, n); }
API float orc_test_np_sum_f32(const float *a, int n) { return orc_np_sum_f32(a, n); }
API void orc_test_standard_normal(const uint64_t pcg[4], int n, double *out, uint64_t pcg_out[4]) {
    orc_pcg64 r;
    r.state = ((orc_u128)pcg[0] << 64) | pcg[1], r.inc = ((orc_u128)pcg[2] << 64) | pcg[3];
    for (int i = 0; i < n; i++) out[i] = orc_standard_normal(&r);
    pcg_out[0] = (uint64_t)(r.state >> 64), pcg_out[1] = (uint64_t)r.state, pcg_out[2] = (uint64_t)(r.inc >> 64), pcg_out[3] = (uint64_t)r.inc;
}
