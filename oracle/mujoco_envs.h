/* mujoco_envs.h -- TEST INFRASTRUCTURE (CPU oracle): env glue of HalfCheetah-v5 / Ant-v5 / Humanoid-v5, see mujoco_envs.c. */
#ifndef ORACLE_MUJOCO_ENVS_H
#define ORACLE_MUJOCO_ENVS_H
#include "mujoco_core.h"
#include "pcg64.h"

enum { ORC_MJ_HALF_CHEETAH = 0, ORC_MJ_ANT = 1, ORC_MJ_HUMANOID = 2, ORC_MJ_HOPPER = 3, ORC_MJ_WALKER2D = 4, ORC_MJ_INVERTED_PENDULUM = 5,
       ORC_MJ_INVERTED_DOUBLE_PENDULUM = 6, ORC_MJ_REACHER = 7, ORC_MJ_HUMANOID_STANDUP = 8, ORC_MJ_SWIMMER = 9, ORC_MJ_PUSHER = 10, ORC_MJ_COUNT = 11 };

typedef struct orc_mjenv {
    int which;
    const mjo_model *m;
    mjo_model *own; /* private copy of the model when the env overrides the solver (solver="Newton" opt-in), else NULL */
    mjo_data d;
    double track_override[2];
    int has_override;
} orc_mjenv;

double orc_standard_normal(orc_pcg64 *rng);
double orc_np_sum_f64(const double *a, int n);
double orc_np_norm(const double *v, int n);
float orc_np_sum_f32(const float *a, int n);
const mjo_model *orc_mj_model(int which);
int orc_mjenv_obs_dim(int which, const double *params);
int orc_mjenv_info_dim(int which);
int orc_mjenv_state_dim(int which);
orc_mjenv *orc_mjenv_create(int which, int force_newton);
void orc_mjenv_obs(const orc_mjenv *e, const double *params, double *obs);
void orc_mjenv_reset(orc_mjenv *e, orc_pcg64 *rng, const double *params);
/* action: nu float32 values, or (act_f64) nu float64 values taken un-rounded (mujoco_env.py:148 data.ctrl[:] = ctrl) */
void orc_mjenv_step(orc_mjenv *e, const void *action, int act_f64, const double *params, double *reward, int *terminated, double *info);
void orc_mjenv_reset_info(const orc_mjenv *e, double *row);
void orc_mjenv_get_state(const orc_mjenv *e, double *s);
void orc_mjenv_set_state(orc_mjenv *e, const double *s);
#endif
