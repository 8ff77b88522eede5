// mjx_kernels.h -- the MuJoCo-family environments as lockstep kernels (included by engine.hip, which owns the C ABI).
//
// Per-env glue restated from the reference's Python (one lane = one scalar env):
//   gymnasium/envs/mujoco/mujoco_env.py:132-155,172-187   set_state / do_simulation / reset
//   gymnasium/envs/mujoco/half_cheetah_v5.py:220-281      step, _get_rew, _get_obs, reset_model
//   gymnasium/envs/mujoco/ant_v5.py:327-428               contact_forces, is_healthy, step, _get_rew, _get_obs, reset_model
//   gymnasium/envs/mujoco/hopper_v5.py:236-343, walker2d_v5.py:241-345            (planar walkers: same skeleton)
//   gymnasium/envs/mujoco/inverted_pendulum_v5.py:160-196, inverted_double_pendulum_v5.py:186-246, reacher_v5.py:188-245
// and the vectoriser semantics shared with the classic-control kernels (TimeLimit, autoreset modes, episode statistics).
// Physics: mjx_core.h.  NumPy arithmetic that the reference inherits (np.sum pairwise order, float32 promotion of the
// control cost, Generator.uniform / standard_normal streams) is reproduced bit for bit.
#pragma once
#include "mjx_core.h"
#include "ziggurat_tables.h"
#include "pow_exact.h"

namespace mjx {

__device__ const uint64_t kZigKi[256] = {MI_ZIG_KI_VALUES};
__device__ const double kZigWi[256] = {MI_ZIG_WI_VALUES};
__device__ const double kZigFi[256] = {MI_ZIG_FI_VALUES};

// numpy random_standard_normal (256-layer ziggurat) on the lane's own PCG64 stream
MJX_DEV double standard_normal(mi::Pcg64 &rng) {
    for (;;) {
        uint64_t r = rng.next64();
        const int idx = (int)(r & 0xff);
        r >>= 8;
        const int sign = (int)(r & 1);
        const uint64_t rabs = (r >> 1) & 0x000fffffffffffffULL;
        double x = (double)rabs * kZigWi[idx];
        if (sign) x = -x;
        if (rabs < kZigKi[idx]) return x;
        if (idx == 0) {
            for (;;) {
                const double xx = -MI_ZIG_INV_R * log1p(-rng.next_double());
                const double yy = -log1p(-rng.next_double());
                if (yy + yy > xx * xx) return ((rabs >> 8) & 1) ? -(MI_ZIG_R + xx) : MI_ZIG_R + xx;
            }
        } else if ((kZigFi[idx - 1] - kZigFi[idx]) * rng.next_double() + kZigFi[idx] < exp(-0.5 * x * x)) {
            return x;
        }
    }
}

// np.sum of a contiguous array: NumPy's pairwise_sum (plain loop < 8 elements, 8 interleaved accumulators up to 128)
// np.linalg.norm of a 2- / 3-vector of float64: sqrt(x.dot(x)) with the BLAS dot's accumulation -- x0 x0, then fused multiply-adds (oracle/mujoco_envs.c
// orc_np_norm, pinned on NumPy by tests/test_mujoco_oracle.py); spelled with the builtin because this glue is compiled with contraction off
MJX_DEV double np_norm(double a, double b) { return sqrt(__builtin_fma(b, b, a * a)); }
MJX_DEV double np_norm(double a, double b, double c) { return sqrt(__builtin_fma(c, c, __builtin_fma(b, b, a * a))); }

template <class T, int N>
MJX_DEV T np_sum(const T *a) {
    static_assert(N <= 128, "block recursion not needed for these envs");
    if (N < 8) {
        T res = 0;
        for (int i = 0; i < N; i++) res += a[i];
        return res;
    }
    T r[8];
    for (int i = 0; i < 8; i++) r[i] = a[i];
    int i = 8;
    for (; i < N - (N % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < N; i++) res += a[i];
    return res;
}

// One action row as the caller gave it: float32 values, or float64 values taken un-rounded (mi_step_io.actions_dtype; mujoco_env.py:148
// `data.ctrl[:] = ctrl`).  Read where it is used -- no per-lane copy: the glue kernels are register-bound.
struct ActRow {
    const void *p;
    bool f64;
    MJX_DEV double operator[](int u) const { return f64 ? static_cast<const double *>(p)[u] : (double)static_cast<const float *>(p)[u]; }
};

enum MjKind { kHalfCheetah = 0, kAnt = 1, kHumanoid = 2, kHopper = 3, kWalker2d = 4, kInvertedPendulum = 5, kInvertedDoublePendulum = 6, kReacher = 7, kHumanoidStandup = 8, kSwimmer = 9, kPusher = 10 };

// quantities of the last forward pass that the observations read besides qpos / qvel (null pointer = zeros, which is
// what mj_resetData leaves in cfrc_ext / qfrc_actuator)
struct ObsExtras {
    const double (*cfrc)[6];
    const double (*cinert)[10];
    const double (*cvel)[6];
    const double *qfrc_actuator;
    const double *qfrc_constraint = nullptr;  // [NV], InvertedDoublePendulum's observation
    const double *vec = nullptr;              // Reacher: [3] fingertip - target; Pusher: [9] positions of tips_arm, object, goal
};

template <class M, int KIND>
struct MjEnv {
    typedef M Model;
    static constexpr int KIND_ID = KIND;
    static constexpr int NQ = M::NQ, NV = M::NV, NU = M::NU, NB = M::NBODY;
    static constexpr int S = NQ + 2 * NV + 2;  // state row: qpos, qvel, (warm-start slot, unused by the Newton solver), tracked xy
    static constexpr bool PLANAR_WALKER = KIND == kHopper || KIND == kWalker2d;
    static constexpr bool HUMANOID_LIKE = KIND == kHumanoid || KIND == kHumanoidStandup;  // same model family, same observation
    static constexpr bool PENDULUM = KIND == kInvertedPendulum || KIND == kInvertedDoublePendulum;
    // info row = the scalar entries of the env's info dict, then (humanoids) tendon_length[NTENDON], tendon_velocity[NTENDON]
    // (humanoid_v5.py:486-487, humanoidstandup_v5.py:433-434)
    static constexpr int INFO_SCALARS =
        KIND == kHalfCheetah ? 4 : (PLANAR_WALKER ? 6 : (KIND == kInvertedPendulum ? 1 : (KIND == kInvertedDoublePendulum ? 3 : (KIND == kReacher ? 2 : (KIND == kHumanoidStandup ? 6 : (KIND == kSwimmer ? 7 : (KIND == kPusher ? 3 : 9)))))));
    static constexpr int INFO = INFO_SCALARS + 2 * M::NTENDON;
    static constexpr bool HAS_COOP = KIND == kHalfCheetah || KIND == kAnt || HUMANOID_LIKE || PLANAR_WALKER;  // the other small robots: one-lane kernel only
    static constexpr int COOP_G = (NV > 16 || NB - 1 > 16) ? 32 : 16;  // lanes per sub-environment in the cooperative kernel (mjx_coop.h)
    static constexpr int SKIP = (KIND == kHalfCheetah || PLANAR_WALKER) ? 1 : ((PENDULUM || KIND == kReacher || KIND == kPusher) ? 0 : 2);
    static constexpr int MAX_OBS = NQ + NV + (KIND == kAnt ? 6 * (NB - 1) : 0) + (HUMANOID_LIKE ? 22 * (NB - 1) + NV - 6 : 0) +
                                   (KIND == kInvertedDoublePendulum ? NQ : 0) + (KIND == kReacher ? 2 : 0) + (KIND == kPusher ? 9 : 0);

    static int obs_dim_host(const mi::EnvParams &P) {  // the same rule, host side (mi_create)
        if (KIND == kReacher) return 10;
        if (KIND == kPusher) return 23;
        if (KIND == kInvertedPendulum) return NQ + NV;
        if (KIND == kInvertedDoublePendulum) return 1 + 2 * (NQ - 1) + NV + 1;
        int n = NQ + NV - (P.p[3] != 0.0 ? SKIP : 0);
        if (KIND == kAnt && P.p[12] != 0.0) n += 6 * (NB - 1);
        if (HUMANOID_LIKE)
            n += (P.p[12] != 0.0 ? 10 * (NB - 1) : 0) + (P.p[13] != 0.0 ? 6 * (NB - 1) : 0) + (P.p[14] != 0.0 ? NV - 6 : 0) +
                 (P.p[15] != 0.0 ? 6 * (NB - 1) : 0);
        return n;
    }
    static MJX_DEV int obs_dim(const mi::EnvParams &P) {
        if (KIND == kReacher) return 10;
        if (KIND == kPusher) return 23;
        if (KIND == kInvertedPendulum) return NQ + NV;
        if (KIND == kInvertedDoublePendulum) return 1 + 2 * (NQ - 1) + NV + 1;
        int n = NQ + NV - (P.p[3] != 0.0 ? SKIP : 0);
        if (KIND == kAnt && P.p[12] != 0.0) n += 6 * (NB - 1);
        if (HUMANOID_LIKE)
            n += (P.p[12] != 0.0 ? 10 * (NB - 1) : 0) + (P.p[13] != 0.0 ? 6 * (NB - 1) : 0) + (P.p[14] != 0.0 ? NV - 6 : 0) +
                 (P.p[15] != 0.0 ? 6 * (NB - 1) : 0);
        return n;
    }

    // ant_v5.py:393-404, half_cheetah_v5.py:248-257, humanoid_v5.py:430-466
    static MJX_DEV void write_obs(const double *s, const ObsExtras &x, const mi::EnvParams &P, double *o) {
        int n = 0;
        if (KIND == kPusher) {
            // pusher_v5.py:317-326: arm qpos[:7], arm qvel[:7], get_body_com of tips_arm, object, goal (= data.body(name).xpos)
            for (int k = 0; k < 7; k++) o[k] = s[k], o[7 + k] = s[NQ + k];
            for (int k = 0; k < 9; k++) o[14 + k] = x.vec ? x.vec[k] : 0.0;
            return;
        }
        if (KIND == kReacher) {
            // reacher_v5.py:232-245: cos(theta), sin(theta), target qpos, arm qvel, (fingertip - target)[:2]
            o[0] = cos(s[0]), o[1] = cos(s[1]), o[2] = sin(s[0]), o[3] = sin(s[1]), o[4] = s[2], o[5] = s[3], o[6] = s[NQ], o[7] = s[NQ + 1];
            o[8] = x.vec ? x.vec[0] : 0.0, o[9] = x.vec ? x.vec[1] : 0.0;
            return;
        }
        if (KIND == kInvertedDoublePendulum) {
            // inverted_double_pendulum_v5.py:217-226: x, sin(angles), cos(angles), clip(qvel, -10, 10), clip(qfrc_constraint, -10, 10)[:1]
            o[n++] = s[0];
            for (int k = 1; k < NQ; k++) o[n++] = sin(s[k]);
            for (int k = 1; k < NQ; k++) o[n++] = cos(s[k]);
            for (int k = 0; k < NV; k++) o[n++] = s[NQ + k] < -10.0 ? -10.0 : (s[NQ + k] > 10.0 ? 10.0 : s[NQ + k]);
            const double f = x.qfrc_constraint ? x.qfrc_constraint[0] : 0.0;
            o[n++] = f < -10.0 ? -10.0 : (f > 10.0 ? 10.0 : f);
            return;
        }
        for (int k = ((P.p[3] != 0.0 && !PENDULUM) ? SKIP : 0); k < NQ; k++) o[n++] = s[k];
        for (int k = 0; k < NV; k++) {
            const double v = s[NQ + k];
            o[n++] = PLANAR_WALKER ? (v < -10.0 ? -10.0 : (v > 10.0 ? 10.0 : v)) : v;  // np.clip(qvel, -10, 10): hopper_v5.py:262
        }
        if (KIND == kAnt && P.p[12] != 0.0)
            for (int b = 1; b < NB; b++)
                for (int k = 0; k < 6; k++) {
                    const double f = x.cfrc ? x.cfrc[b][k] : 0.0;
                    o[n++] = f < P.p[10] ? P.p[10] : (f > P.p[11] ? P.p[11] : f);  // np.clip(cfrc_ext, lo, hi)
                }
        if (HUMANOID_LIKE) {
            if (P.p[12] != 0.0)
                for (int b = 1; b < NB; b++)
                    for (int k = 0; k < 10; k++) o[n++] = x.cinert ? x.cinert[b][k] : 0.0;
            if (P.p[13] != 0.0)
                for (int b = 1; b < NB; b++)
                    for (int k = 0; k < 6; k++) o[n++] = x.cvel ? x.cvel[b][k] : 0.0;
            if (P.p[14] != 0.0)
                for (int k = 6; k < NV; k++) o[n++] = x.qfrc_actuator ? x.qfrc_actuator[k] : 0.0;
            if (P.p[15] != 0.0)
                for (int b = 1; b < NB; b++)
                    for (int k = 0; k < 6; k++) o[n++] = x.cfrc ? x.cfrc[b][k] : 0.0;
        }
    }

    // mass_center (humanoid_v5.py:17-21): einsum("b,bj->j", body_mass, xipos) / body_mass.sum()
    static MJX_DEV void mass_center_xy(const Data<M> &d, double *out) {
        double nx = 0, ny = 0, mass[NB];
        for (int b = 0; b < NB; b++) nx += M::body_mass[b] * d.xipos[b][0], ny += M::body_mass[b] * d.xipos[b][1], mass[b] = M::body_mass[b];
        const double den = np_sum<double, NB>(mass);
        out[0] = nx / den, out[1] = ny / den;
    }

    // reset_model + set_state (-> mj_forward); writes the reset observation when obs != nullptr
    static MJX_DEV void reset(mi::Pcg64 &rng, double *s, const mi::EnvParams &P, double *obs) {
        if (KIND == kPusher) {
            // pusher_v5.py:293-315: the arm starts at init_qpos; the object's sliders (y first, then x: joint order of the XML) are re-drawn
            // until the object is farther than 0.17 from the goal; small arm velocities, object and goal at rest
            for (int k = 0; k < NQ; k++) s[k] = M::qpos0[k];
            for (;;) {
                const double c0 = -0.3 + (0.0 - (-0.3)) * rng.next_double(), c1 = -0.2 + (0.2 - (-0.2)) * rng.next_double();
                s[NQ - 4] = c0, s[NQ - 3] = c1;
                if (np_norm(c0, c1) > 0.17) break;
            }
            s[NQ - 2] = 0.0, s[NQ - 1] = 0.0;
            for (int k = 0; k < NV; k++) s[NQ + k] = 0.0 + (-0.005 + (0.005 - (-0.005)) * rng.next_double());
            for (int k = NV - 4; k < NV; k++) s[NQ + k] = 0.0;
            for (int k = 0; k < NV; k++) s[NQ + NV + k] = 0.0;
            s[NQ + 2 * NV] = 0.0, s[NQ + 2 * NV + 1] = 0.0;
            if (obs) {
                Data<M> d;
                for (int k = 0; k < NQ; k++) d.qpos[k] = s[k];
                kinematics<M>(d);
                double vec[9];
                for (int b = 0; b < 3; b++)
                    for (int k = 0; k < 3; k++) vec[3 * b + k] = d.xpos[NB - 3 + b][k];
                ObsExtras x = {nullptr, nullptr, nullptr, nullptr};
                x.vec = vec;
                write_obs(s, x, P, obs);
            }
            return;
        }
        if (KIND == kReacher) {
            // reacher_v5.py:209-226: arm + target noise, the goal re-drawn until it lies inside the 0.2 disc, small arm velocities
            for (int k = 0; k < NQ; k++) s[k] = (-0.1 + (0.1 - (-0.1)) * rng.next_double()) + M::qpos0[k];
            for (;;) {
                const double g0 = -0.2 + (0.2 - (-0.2)) * rng.next_double(), g1 = -0.2 + (0.2 - (-0.2)) * rng.next_double();
                s[2] = g0, s[3] = g1;
                if (np_norm(g0, g1) < 0.2) break;
            }
            for (int k = 0; k < NV; k++) s[NQ + k] = 0.0 + (-0.005 + (0.005 - (-0.005)) * rng.next_double());
            s[NQ + 2] = 0.0, s[NQ + 3] = 0.0;
            for (int k = 0; k < NV; k++) s[NQ + NV + k] = 0.0;
            s[NQ + 2 * NV] = 0.0, s[NQ + 2 * NV + 1] = 0.0;
            if (obs) {  // the observation shows fingertip - target of the forward pass at the reset state
                Data<M> d;
                for (int k = 0; k < NQ; k++) d.qpos[k] = s[k];
                kinematics<M>(d);
                const double vec[3] = {d.xpos[3][0] - d.xpos[4][0], d.xpos[3][1] - d.xpos[4][1], d.xpos[3][2] - d.xpos[4][2]};
                ObsExtras x = {nullptr, nullptr, nullptr, nullptr};
                x.vec = vec;
                write_obs(s, x, P, obs);
            }
            return;
        }
        const double scale = P.p[2];
        for (int k = 0; k < NQ; k++) s[k] = M::qpos0[k] + (-scale + (scale - (-scale)) * rng.next_double());
        if (HUMANOID_LIKE || PLANAR_WALKER || KIND == kInvertedPendulum || KIND == kSwimmer)  // swimmer_v5.py:279-294, humanoid_v5.py:526-528, hopper_v5.py:318-331: uniform noise on the velocities as well
            for (int k = 0; k < NV; k++) s[NQ + k] = 0.0 + (-scale + (scale - (-scale)) * rng.next_double());
        else
            for (int k = 0; k < NV; k++) s[NQ + k] = 0.0 + scale * standard_normal(rng);
        for (int k = 0; k < NV; k++) s[NQ + NV + k] = 0.0;
        if (HUMANOID_LIKE) {
            // the observation shows cinert / cvel of the forward pass at the reset state, and the tracked point is the
            // whole-body centre of mass
            Data<M> d;
            for (int k = 0; k < NQ; k++) d.qpos[k] = s[k];
            for (int k = 0; k < NV; k++) d.qvel[k] = s[NQ + k];
            kinematics<M>(d);
            com_pos<M>(d);
            double bias[NV];
            com_vel_and_bias<M>(d, bias);
            mass_center_xy(d, s + NQ + 2 * NV);
            if (obs) {
                const ObsExtras x = {nullptr, d.cinert, d.cvel, nullptr};
                write_obs(s, x, P, obs);
            }
            return;
        }
        // free joint / slider: the tracked Cartesian position is the joint's own coordinate
        s[NQ + 2 * NV] = s[0], s[NQ + 2 * NV + 1] = (KIND == kHalfCheetah || PLANAR_WALKER || PENDULUM) ? 0.0 : s[1];
        if (obs) {
            const ObsExtras x = {nullptr, nullptr, nullptr, nullptr};
            write_obs(s, x, P, obs);
        }
    }

    // What the reward / observation code reads from the last forward pass of the physics (mj_rnePostConstraint included).
    struct StepExtras {
        double after[2];                 // tracked Cartesian point (x, y) after the step
        const double (*cfrc)[6];         // cfrc_ext[NB][6]
        const double (*cinert)[10];      // cinert[NB][10] (per body, not composite)
        const double (*cvel)[6];         // cvel[NB][6]
        const double *qfrc_actuator;     // [NV]
        const double *qfrc_constraint;   // [NV]
        double vec[9];                   // Reacher: fingertip - target (body frames 3 and 4) of the last forward pass; Pusher: xpos of the last three bodies
        const double *ten = nullptr;     // [2 NTENDON] data.ten_length, data.ten_velocity of the last forward pass
    };
    // the tendon columns of an info row (data.ten_length / data.ten_velocity are views of the LAST forward pass's values)
    static MJX_DEV void write_tendon_info(const double *ten, double *info) {
#pragma unroll
        for (int k = 0; k < 2 * M::NTENDON; k++) info[INFO_SCALARS + k] = ten[k];
    }

    // One env.step() with the one-lane simulator (mjx_core.h): physics, then finish().
    static MJX_DEV void step(double *s, const ActRow action, const mi::EnvParams &P, double *obs, double &reward, bool &terminated,
                             double *info, bool newton = false) {
        Data<M> d;
        for (int k = 0; k < NQ; k++) d.qpos[k] = s[k];
        for (int k = 0; k < NV; k++) d.qvel[k] = s[NQ + k];
        for (int u = 0; u < NU; u++) d.ctrl[u] = action[u];
        for (int k = 0; k < NV; k++) d.qacc_warm[k] = s[NQ + NV + k];  // the qacc_warmstart slot of the state row
        const double before[2] = {s[NQ + 2 * NV], s[NQ + 2 * NV + 1]};
        const int frame_skip = (int)P.p[4];
        if constexpr (M::SOLVER == 1) {  // the MJCF's PGS / 50, or the opt-in Newton solver (MI_CFG_SOLVER_NEWTON)
            if (newton)
                for (int f = 0; f < frame_skip; f++) mjx::step<M, false>(d);
            else
                for (int f = 0; f < frame_skip; f++) mjx::step<M, true>(d);
        } else {
            for (int f = 0; f < frame_skip; f++) mjx::step<M>(d);
        }
        for (int k = 0; k < NV; k++) s[NQ + NV + k] = d.qacc_warm[k];
        // Cartesian quantities of the LAST forward pass (they lag qpos by one sub-step, as in the reference)
        StepExtras x;
        if (KIND == kHumanoidStandup) {
            x.after[0] = x.after[1] = 0.0;
        } else if (KIND == kPusher) {
            x.after[0] = x.after[1] = 0.0;
            for (int b = 0; b < 3; b++)
                for (int k = 0; k < 3; k++) x.vec[3 * b + k] = d.xpos[NB - 3 + b][k];  // tips_arm, object, goal
        } else if (KIND == kReacher) {
            x.after[0] = x.after[1] = 0.0;
            for (int k = 0; k < 3; k++) x.vec[k] = d.xpos[3][k] - d.xpos[4][k];
        } else if (KIND == kInvertedDoublePendulum) {  // the tip site of the LAST forward pass: x and z (the reference's `x, _, y = site_xpos[0]`)
            const int sb = M::site_bodyid[0];
            double t[3];
            rot_vec(t, d.xmat[sb], M::site_pos[0]);
            x.after[0] = d.xpos[sb][0] + t[0], x.after[1] = d.xpos[sb][2] + t[2];
        } else if (KIND == kHalfCheetah || PLANAR_WALKER || KIND == kInvertedPendulum)
            x.after[0] = d.qpos[0], x.after[1] = 0.0;
        else if (KIND == kSwimmer)  // data.qpos[0:2] (swimmer_v5.py:226-228)
            x.after[0] = d.qpos[0], x.after[1] = d.qpos[1];
        else if (KIND == kAnt)
            x.after[0] = d.xpos[1][0], x.after[1] = d.xpos[1][1];
        else
            mass_center_xy(d, x.after);
        for (int k = 0; k < NQ; k++) s[k] = d.qpos[k];
        for (int k = 0; k < NV; k++) s[NQ + k] = d.qvel[k];
        double cfrc[NB][6];
        if (KIND == kAnt || HUMANOID_LIKE) contact_forces<M>(d, cfrc);
        if (HUMANOID_LIKE) com_pos<M>(d);  // crb() folded the per-body inertias into composites: restore them for the obs
        x.cfrc = cfrc, x.cinert = d.cinert, x.cvel = d.cvel, x.qfrc_actuator = d.qfrc_actuator, x.qfrc_constraint = d.qfrc_constraint;
        double ten[M::NTENDON > 0 ? 2 * M::NTENDON : 1];
        for (int t = 0; t < M::NTENDON; t++) ten[t] = d.ten_length[t], ten[M::NTENDON + t] = d.ten_velocity[t];
        x.ten = ten;
        finish(s, before, x, action, P, obs, reward, terminated, info);
    }

    // The tracked point from the cooperative kernel's extras row (coop::Sim::write_extras): body-1 position for Ant, the
    // mass-weighted sum of xipos over np.sum(body_mass) for Humanoid (humanoid_v5.py:17-21), the root slider for HalfCheetah, Hopper and Walker2d.
    static MJX_DEV void after_from_extras(const double *s, const double *ex, double *after) {
        if (KIND == kHumanoidStandup) {
            after[0] = after[1] = 0.0;
        } else if (KIND == kHalfCheetah || PLANAR_WALKER) {
            after[0] = s[0], after[1] = 0.0;
        } else if (KIND == kAnt) {
            after[0] = ex[0], after[1] = ex[1];
        } else {
            double mass[NB];
            for (int b = 0; b < NB; b++) mass[b] = M::body_mass[b];
            const double den = np_sum<double, NB>(mass);
            after[0] = ex[2] / den, after[1] = ex[3] / den;
        }
    }

    // Everything of env.step() after do_simulation: s holds the NEW qpos / qvel, `before` the tracked point before the step.
    // weight * np.sum(np.square(action)) in the action row's OWN dtype (half_cheetah_v5.py:216-218 control_cost and its siblings): NEP 50 keeps a
    // float32 row's reduction and its product with the Python-float weight in float32; a float64 row makes all of it float64.  Returned widened to
    // double (exact).  -np.square(action).sum() * w (reacher_v5.py:201, pusher_v5.py:281) is its negation bit for bit.
    static MJX_DEV double control_cost(const ActRow action, double weight) {
        if (!action.f64) {
            float sq[NU];
            for (int u = 0; u < NU; u++) sq[u] = static_cast<const float *>(action.p)[u] * static_cast<const float *>(action.p)[u];
            return (double)((float)weight * np_sum<float, NU>(sq));
        }
        double sq[NU];
        for (int u = 0; u < NU; u++) sq[u] = action[u] * action[u];
        return weight * np_sum<double, NU>(sq);
    }
    static MJX_DEV void finish(double *s, const double *before, const StepExtras &x, const ActRow action, const mi::EnvParams &P, double *obs,
                               double &reward, bool &terminated, double *info) {
        const int frame_skip = (int)P.p[4];
        const double *after = x.after;
        s[NQ + 2 * NV] = after[0], s[NQ + 2 * NV + 1] = after[1];
        const double dt = M::TIMESTEP * frame_skip;
        const double xv = (after[0] - before[0]) / dt, yv = (after[1] - before[1]) / dt;
        const double ctrl_cost_f = control_cost(action, P.p[1]);  // weight * np.sum(np.square(action)): float32 arithmetic for a float32 row
        if (KIND == kHumanoidStandup) {
            // humanoidstandup_v5.py:423-462: reward = z / opt.timestep - w_ctrl sum(ctrl^2) - clip(w_impact sum(cfrc_ext^2)) + 1 (the
            // `uph_cost_weight` argument is stored but never applied there); never terminates
            const double uph_cost = (s[2] - 0) / M::TIMESTEP;
            double sqd[NU], c2s[6 * NB];
            for (int u = 0; u < NU; u++) sqd[u] = action[u] * action[u];
            const double quad_ctrl_cost = P.p[1] * np_sum<double, NU>(sqd);
            for (int b = 0; b < NB; b++)
                for (int k = 0; k < 6; k++) c2s[6 * b + k] = x.cfrc[b][k] * x.cfrc[b][k];
            double quad_impact_cost = P.p[5] * np_sum<double, 6 * NB>(c2s);
            quad_impact_cost = quad_impact_cost < P.p[10] ? P.p[10] : (quad_impact_cost > P.p[11] ? P.p[11] : quad_impact_cost);
            reward = uph_cost - quad_ctrl_cost - quad_impact_cost + 1;
            terminated = false;
            const ObsExtras ox = {x.cfrc, x.cinert, x.cvel, x.qfrc_actuator};
            write_obs(s, ox, P, obs);
            if (info) {
                info[0] = s[0], info[1] = s[1], info[2] = s[2] - M::qpos0[2], info[3] = uph_cost, info[4] = -quad_ctrl_cost, info[5] = -quad_impact_cost;
                write_tendon_info(x.ten, info);
            }
            return;
        }
        if (KIND == kPusher) {
            // pusher_v5.py:266-291: reward = -|object - goal| w_dist + (-sum(a^2) w_ctrl, float32) + (-|object - tips_arm| w_near); never terminates
            const double *tip = x.vec, *ob = x.vec + 3, *goal = x.vec + 6;
            const double v1[3] = {ob[0] - tip[0], ob[1] - tip[1], ob[2] - tip[2]}, v2[3] = {ob[0] - goal[0], ob[1] - goal[1], ob[2] - goal[2]};
            const double reward_near = -np_norm(v1[0], v1[1], v1[2]) * P.p[0];
            const double reward_dist = -np_norm(v2[0], v2[1], v2[2]) * P.p[5];
            const double reward_ctrl = -ctrl_cost_f;
            reward = (reward_dist + (double)reward_ctrl) + reward_near;
            terminated = false;
            ObsExtras ox = {nullptr, nullptr, nullptr, nullptr};
            ox.vec = x.vec;
            write_obs(s, ox, P, obs);
            if (info) info[0] = reward_dist, info[1] = (double)reward_ctrl, info[2] = reward_near;
            return;
        }
        if (KIND == kReacher) {
            // reacher_v5.py:188-207: reward = -|fingertip - target| w_dist - sum(a^2) w_ctrl (the control term in float32); never terminates
            const double reward_dist = -np_norm(x.vec[0], x.vec[1], x.vec[2]) * P.p[0];
            const double reward_ctrl = -ctrl_cost_f;
            reward = reward_dist + (double)reward_ctrl;
            terminated = false;
            ObsExtras ox = {nullptr, nullptr, nullptr, nullptr};
            ox.vec = x.vec;
            write_obs(s, ox, P, obs);
            if (info) info[0] = reward_dist, info[1] = (double)reward_ctrl;
            return;
        }
        if (KIND == kInvertedPendulum) {
            // inverted_pendulum_v5.py:160-176: terminated = not isfinite(obs).all() or |angle| > 0.2; reward = int(not terminated)
            bool finite = true;
            for (int k = 0; k < NQ + NV; k++) finite &= isfinite(s[k]);
            terminated = !finite || fabs(s[1]) > 0.2;
            reward = terminated ? 0.0 : 1.0;
            const ObsExtras ox = {nullptr, nullptr, nullptr, nullptr};
            write_obs(s, ox, P, obs);
            if (info) info[0] = reward;
            return;
        }
        if (KIND == kInvertedDoublePendulum) {
            // inverted_double_pendulum_v5.py:186-215: tip site (x, y = height), penalties on the distance from upright and the joint speeds
            const double tx = after[0], ty = after[1];
            terminated = ty <= 1.0;
            const double v1 = s[NQ + 1], v2 = s[NQ + 2];
            // np.float64 scalars `** 2`: libm pow(x, 2.0), restated bit for bit (pow_exact.h; its tables read from global memory here: four squares a step)
            auto sq = [](double v) { return mi_pow::square<false>(mi_pow::kLogTab, mi_pow::kExpTab, v); };
            const double dist_penalty = 0.01 * sq(tx) + sq(ty - 2);
            const double vel_penalty = 1e-3 * sq(v1) + 5e-3 * sq(v2);
            const double alive_bonus = P.p[6] * (terminated ? 0.0 : 1.0);
            reward = alive_bonus - dist_penalty - vel_penalty;
            ObsExtras ox = {nullptr, nullptr, nullptr, nullptr};
            ox.qfrc_constraint = x.qfrc_constraint;
            write_obs(s, ox, P, obs);
            if (info) info[0] = alive_bonus, info[1] = -dist_penalty, info[2] = -vel_penalty;
            return;
        }
        if (PLANAR_WALKER) {
            // hopper_v5.py:240-257,266-309 / walker2d_v5.py:245-259,268-311
            const double forward_reward = P.p[0] * xv;
            const double z = s[1], angle = s[2];
            bool healthy = P.p[8] < z && z < P.p[9] && P.p[10] < angle && angle < P.p[11];
            if (KIND == kHopper)  // healthy_state_range on state_vector()[2:]
                for (int k = 2; k < NQ + NV; k++) healthy = healthy && P.p[12] < s[k] && s[k] < P.p[13];
            const double healthy_reward = healthy ? P.p[6] : 0.0;
            reward = (forward_reward + healthy_reward) - (double)ctrl_cost_f;
            terminated = !healthy && P.p[7] != 0.0;
            const ObsExtras ox = {nullptr, nullptr, nullptr, nullptr};
            write_obs(s, ox, P, obs);
            if (info)
                info[0] = s[0], info[1] = s[1] - M::qpos0[1], info[2] = xv, info[3] = forward_reward, info[4] = -(double)ctrl_cost_f,
                info[5] = healthy_reward;
            return;
        }
        if (KIND == kSwimmer) {
            // swimmer_v5.py:225-263: forward velocity of the root sliders, float32 control cost, never terminates
            const double forward_reward = P.p[0] * xv;
            reward = forward_reward - (double)ctrl_cost_f;
            terminated = false;
            const ObsExtras ox = {nullptr, nullptr, nullptr, nullptr};
            write_obs(s, ox, P, obs);
            if (info) {
                info[0] = after[0], info[1] = after[1], info[2] = np_norm(after[0], after[1]), info[3] = xv, info[4] = yv;
                info[5] = forward_reward, info[6] = -(double)ctrl_cost_f;
            }
            return;
        }
        if (KIND == kHalfCheetah) {
            const double forward_reward = P.p[0] * xv;
            reward = forward_reward - (double)ctrl_cost_f;
            terminated = false;
            const ObsExtras ox = {nullptr, nullptr, nullptr, nullptr};
            write_obs(s, ox, P, obs);
            if (info) info[0] = s[0], info[1] = xv, info[2] = forward_reward, info[3] = -(double)ctrl_cost_f;
            return;
        }
        const double (*cfrc)[6] = x.cfrc;
        double c2[6 * NB];
        bool healthy;
        double ctrl_cost, contact_cost;
        if (KIND == kAnt) {
            bool finite = true;
            for (int k = 0; k < NQ + NV; k++) finite &= isfinite(s[k]);
            healthy = finite && P.p[8] <= s[2] && s[2] <= P.p[9];
            for (int b = 0; b < NB; b++)
                for (int k = 0; k < 6; k++) {
                    double f = cfrc[b][k];
                    f = f < P.p[10] ? P.p[10] : (f > P.p[11] ? P.p[11] : f);
                    c2[6 * b + k] = f * f;
                }
            contact_cost = P.p[5] * np_sum<double, 6 * NB>(c2), ctrl_cost = (double)ctrl_cost_f;
        } else {
            healthy = P.p[8] < s[2] && s[2] < P.p[9];
            double sqd[NU];
            for (int u = 0; u < NU; u++) sqd[u] = action[u] * action[u];  // np.square(self.data.ctrl): float64
            ctrl_cost = P.p[1] * np_sum<double, NU>(sqd);
            for (int b = 0; b < NB; b++)
                for (int k = 0; k < 6; k++) c2[6 * b + k] = cfrc[b][k] * cfrc[b][k];
            contact_cost = P.p[5] * np_sum<double, 6 * NB>(c2);
            contact_cost = contact_cost < P.p[10] ? P.p[10] : (contact_cost > P.p[11] ? P.p[11] : contact_cost);
        }
        const double forward_reward = KIND == kAnt ? xv * P.p[0] : P.p[0] * xv, healthy_reward = healthy ? P.p[6] : 0.0;
        const double rewards = forward_reward + healthy_reward, costs = ctrl_cost + contact_cost;
        reward = rewards - costs;
        terminated = !healthy && P.p[7] != 0.0;
        const ObsExtras ox = {cfrc, x.cinert, x.cvel, x.qfrc_actuator};
        write_obs(s, ox, P, obs);
        if (info) {
            info[0] = s[0], info[1] = s[1], info[2] = np_norm(s[0], s[1]), info[3] = xv, info[4] = yv;
            info[5] = forward_reward, info[6] = -ctrl_cost, info[7] = -contact_cost, info[8] = healthy_reward;
            if (HUMANOID_LIKE) write_tendon_info(x.ten, info);
        }
    }
    static MJX_DEV void reset_info(const double *s, double *info) {
        for (int k = 0; k < INFO; k++) info[k] = 0.0;
        if (PENDULUM || KIND == kReacher || KIND == kPusher) return;  // _get_reset_info is empty (inverted_pendulum_v5.py:198-199)
        info[0] = s[0];
        if constexpr (M::NTENDON > 0) {  // mj_forward at the reset state: humanoid_v5.py:534-541
            double ten[2 * M::NTENDON];
            tendons<M>(s, s + NQ, ten, ten + M::NTENDON);
            write_tendon_info(ten, info);
        }
        if (KIND == kHumanoidStandup) {
            info[1] = s[1], info[2] = s[2] - M::qpos0[2];  // humanoidstandup_v5.py:479-486
            return;
        }
        if (PLANAR_WALKER) {
            info[1] = s[1] - M::qpos0[1];  // z_distance_from_origin (hopper_v5.py:338-342)
            return;
        }
        if (KIND != kHalfCheetah) info[1] = s[1], info[2] = np_norm(s[0], s[1]);
    }
};

}  // namespace mjx
