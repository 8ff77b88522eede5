// mjx_coop.h -- cooperative articulated-body simulator for gfx950: G lanes of one wavefront advance ONE sub-environment.
//
// Why: the one-env-per-lane simulator (mjx_core.h) needs ~22 KB of private state per lane, which
// lives in scratch: 1.2 MB of HBM traffic per Ant env-step, one wavefront per SIMD, 86 % of the cycles in s_waitcnt.
// Here a group of G = 16 (HalfCheetah, Ant) or 32 (Humanoid) lanes shares one environment:
//   * shared quantities (poses, motion subspaces, twists, contact frames, solver vectors) sit on a per-environment
//     BLACKBOARD in LDS (struct Board),
//   * every lane plays three roles -- body `lane + 1`, dof `lane`, contacts `lane + k G` -- and keeps the row of the mass
//     matrix / Newton Hessian of its dof, the constants of its body and the state of its contacts in REGISTERS,
//   * tree recursions run level by level (depth <= 6), matrix factorisations column by column with the pivot column
//     exchanged through LDS, reductions as wavefront butterflies (__shfl_xor inside the group),
//   * nothing is ever private-indexed dynamically, so nothing spills to scratch.
// The same pipeline as mjx_core.h (which documents the formulation and stays as the one-lane cross-check): kinematics ->
// com-based spatial quantities -> RNE bias -> CRB mass matrix -> Cholesky -> collision -> soft constraints (joint limits,
// pyramidal / frictionless contacts) -> primal Newton with exact line search -> semi-implicit Euler (implicit joint damping)
// or RK4.  Reference call sites replaced: gymnasium/envs/mujoco/mujoco_env.py:142 (mj_forward), :150 (mj_step(nstep)),
// :155 (mj_rnePostConstraint).  Parity with `mujoco` itself: UNPINNED (see mjx_core.h).
//
// Synchronisation model: the G lanes of a group always sit in the same wavefront, so they run in lockstep; coop_sync() only
// orders LDS traffic (a compiler + memory fence, no s_barrier).  Discipline (bulk-synchronous): between two coop_sync()
// calls no LDS location is written by one lane and read or written by another.  Under MJX_HOST_EMU (tests/coop_emu, test
// infrastructure only) the identical source runs on the CPU with one fiber per lane and coop_sync() = round-robin yield,
// which is how the algorithm is validated against the C oracle without a GPU.
#pragma once
#include "mjx_core.h"

#if !defined(MJX_HOST_EMU)
#pragma clang fp contract(fast)
#endif

#if defined(MJX_HOST_EMU)
#define MJX_SCHED_FENCE() ((void)0)
#else
#define MJX_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

namespace mjx {
namespace coop {

#if defined(MJX_HOST_EMU)
void coop_sync();  // provided by the harness: yields to the next lane's fiber
extern long g_stat[8];  // harness counters: [0] forwards, [1] forwards with constraint rows, [2] Newton iterations, [3] line-search iterations
static inline unsigned lds_or(unsigned *p, unsigned v) {
    const unsigned o = *p;
    *p = o | v;
    return o;
}
static inline double rsq(double x) { return 1.0 / sqrt(x); }
static inline int popc(unsigned x) { return __builtin_popcount(x); }
template <int G>
static inline double group_sum(double v, double (*red)[32], int lane) {
    for (int off = G / 2; off > 0; off >>= 1) {
        red[0][lane] = v;
        coop_sync();
        v = v + red[0][lane ^ off];
        coop_sync();
    }
    return v;
}
#else
MJX_DEV void coop_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
MJX_DEV unsigned lds_or(unsigned *p, unsigned v) { return atomicOr(p, v); }
// 1/sqrt(x): hardware estimate + two Newton-Raphson steps in FMA arithmetic (full double precision for the pivots)
MJX_DEV double rsq(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-x * y, y, 1.0);
    y = fma(y * 0.5, e, y);
    e = fma(-x * y, y, 1.0);
    return fma(y * 0.5, e, y);
}
MJX_DEV int popc(unsigned x) { return __popc(x); }
// Data-parallel-primitive move of a double inside each row of 16 lanes (two 32-bit v_mov_b32_dpp: a VALU operand modifier, no LDS
// round trip, no s_waitcnt).  CTRL: 0x120 + n = row_ror:n (rotate right by n lanes), 0x150 + k = row_newbcast:k (lane k to all).
template <int CTRL>
MJX_DEV double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// All-reduce over the G lanes of a group: rotations by 8, 4, 2, 1 inside the 16-lane DPP row (every lane ends with the same
// bits: the partial sums of one level are identical on the lanes that get combined at the next, the same tree as the xor
// butterfly of the host emulation); a 32-lane group first folds its two rows with one ds_bpermute.
// v_permlane16_swap (new on gfx950) exchanges the odd 16-lane rows of one register with the even rows of another; fed the same
// value twice it returns {even row's data in both rows, odd row's data in both rows}: the cross-row step of a 32-lane group.
MJX_DEV void row_pair(double v, double &even_rows, double &odd_rows) {
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    even_rows = __hiloint2double((int)hi[0], (int)lo[0]), odd_rows = __hiloint2double((int)hi[1], (int)lo[1]);
}
template <int G>
MJX_DEV double group_sum(double v, decltype(nullptr), int) {
    if (G == 32) {
        double a, b;
        row_pair(v, a, b);
        v = a + b;
    }
    v = v + dpp_mov<0x128>(v);
    v = v + dpp_mov<0x124>(v);
    v = v + dpp_mov<0x122>(v);
    v = v + dpp_mov<0x121>(v);
    return v;
}
#endif

// Blackboard of one sub-environment (LDS).  Arrays whose lifetimes inside one forward pass do not overlap share storage:
//   A: kinematics / collision (xquat, xmat)          | solver (packed Cholesky factor, search direction)
//   B: RNE pass (cacc, cfrc)                         | CRB pass (per-body inertias of all bodies)
//   C: CRB pass (composite inertias, I * cdof)       | solver (twists of the search direction, contact Jacobian exchange, forces)
// Phase order in forward(): kinematics -> com_pos -> collision -> com_vel_and_bias -> crb -> constraint rows -> solver.
// Phase timing (scripts/coop_phase_bench.hip only): MJX_PHASE(r, k) adds the shader cycles since the previous mark to slot k.
#if defined(MJX_PHASE_TIMING) && !defined(MJX_HOST_EMU)
#define MJX_PHASE(r, k)                                         \
    do {                                                        \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        (r).tphase[k] += now_ - (r).tmark, (r).tmark = now_;    \
    } while (0)
#else
#define MJX_PHASE(r, k) ((void)0)
#endif
// finer marks inside one phase, slots 12 .. 15: only the group selected by -DMJX_DETAIL=<tag> is compiled in (1: PGS factor, 2: crb, 3: RNE, 4: com_pos / kinematics)
#ifndef MJX_DETAIL
#define MJX_DETAIL 1
#endif
#define MJX_PHASE_X(r, tag, k)                \
    do {                                      \
        if (MJX_DETAIL == (tag)) MJX_PHASE(r, k); \
    } while (0)

template <class M, int G_>
struct Board {
    static constexpr int G = G_, NQ = M::NQ, NV = M::NV, NB = M::NBODY, NU = M::NU, NTRI = M::NV * (M::NV + 1) / 2;
    static constexpr int MAXCON = M::MAXCON, NSLOT = M::NSLOT, KC = (MAXCON + G - 1) / G, KS = (NSLOT + G - 1) / G;
    static_assert(NB - 1 <= G && NV <= G && G <= 32, "one lane per body and per dof");
    double qpos[NQ], qvel[NV], ctrl[NU];
    double q0[NQ], dv[NV];                                    // RK4: base position, stage velocity
    double xpos[NB][3], xipos[NB][3];
    double com[3];
    double cdof[NV][6];
    double cvel[NB][6];
    double con_r[MAXCON][3], con_frame[MAXCON][9], con_dist[MAXCON];
    union {
        struct {
            double xquat[NB][4], xmat[NB][9];
        } kin;
        struct {
            double L[NTRI];                                   // packed lower Cholesky factor (column access in back substitution)
            double col[2][(G_ == 16 && !0) ? 1 : NV];                 // pivot column / substitution exchange of the 32-lane variant (double buffered)
            double vdir[NV];                                  // solution of the last solve (qacc_smooth, then the search directions)
        } sol;
    } A;
    union {
        struct {
            double cacc[NB][6], cfrc[NB][6];
        } rne;
        double cinert[NB][10];
        double geo[(M::NSLOT > 2 * G_) ? M::NGEOM : 1][6];  // collision (Sim::COLLIDE_TABLES): world position and axis of every geom (dead before the RNE pass starts)
    } Bu;
    union {
        struct {
            double cinertc[NB][10], buf[NV][6];
        } crb;
        struct {
            double tw[NB][6];                                 // per-body twist of a dof-space vector (J v without the frames)
            double jc[2][3][NV];                              // contact-frame Jacobian of the contact being assembled
            double gw[2][8];                                  // its owner's sum_active D jar e (3) and sum_active D e e^T (5)
            double con_F[MAXCON][3];                          // world-frame contact forces of the last forward pass
        } sol;
    } C;
#if defined(MJX_HOST_EMU)
    double red[1][32];
#endif
    // Mass matrix rows: in registers for 16-lane robots; for NV > 16 (Humanoid: 23 doubles per lane on top of the 23 of the
    // Hessian row) they live here, stored column-wise (M is symmetric: lane i reads M[j][i] at [j][i], consecutive lanes ->
    // consecutive addresses), which is what keeps that kernel from spilling.
    static constexpr bool M_IN_LDS = NV > 16;
    double Mt[M_IN_LDS ? NV : 1][M_IN_LDS ? NV : 1];
    // PGS (Sim::bblock): M^-1 J_c^T blocks beyond the ones that fit the dead storage of M, of the Cholesky factor and of union Bu -- as many as keep
    // the 32-lane robots at four wavefronts' worth of LDS per CU (2 x 20 064 B per wavefront for the Humanoid)
    static constexpr int NBX = M_IN_LDS ? 2 : 0;
    // One word that nothing reads: until round 6 the (switched-off) blocked MFMA Cholesky kept its 8 x 8 Schur tile here, one double for every robot
    // without it.  The word stays because a Board that grows or shrinks by a single double re-lays every LDS offset behind it, and these kernels'
    // code generation is sensitive to that (measured in round 3: Ant 5.02 -> 4.83 M env-steps/s for one word more, Humanoid 1.26 -> 1.23 M when the
    // word was folded into a union; scripts/r03/gpu_call37.sh, gpu_call38.sh): removing the dead code must not move the shipped kernels.
    double layout_word[1][1];
    double bx[NBX][NBX ? 3 * NV : 1];  // live from the row assembly to the end of the sweeps
    int con_pair[MAXCON];             // geom pair of the contact | body of its first geom << 16 | body of its second geom << 24 (the bodies
                                      // ride along because pair -> geom -> body is two dependent table loads from global memory per use;
                                      // no room for more: the 16-lane robots sit exactly at four workgroups' worth of LDS per CU)
    unsigned cmask[KS], anyrow;
    unsigned limmask[2];  // dofs whose lower / upper joint-limit row is active in this forward pass (PGS visits them in dof order)
    int ncon;
    double ten[M::NTENDON > 0 ? 2 * M::NTENDON : 1];  // fixed tendons at this forward pass: lengths, then velocities
};

// Everything a lane keeps in registers across phases.
template <class M, int G>
struct Lane {
    static constexpr int NV = M::NV, KC = Board<M, G>::KC;
    // body role
    double anchor[M::MAXJPB][3], axis[M::MAXJPB][3];
    double xmat[9], cinert[10];
    // dof role
    double cdof[6], Mrow[Board<M, G>::M_IN_LDS ? 1 : NV], Hrow[NV], idiag;
    double bias, qfrc_smooth, qfrc_actuator, qacc_smooth, qacc, qacc_int, qfrc_constraint;
#if defined(MJX_PHASE_TIMING) && !defined(MJX_HOST_EMU)
    unsigned long long tphase[16], tmark;
#endif
#ifdef MJX_COUNT_WORK
    int work, work_wave;  // harness statistics: passes of the solver's state machine (factor / solve / assembly rounds) of this sub-environment
                          // since they were cleared, and the same for the slowest sub-environment of the wavefront in every forward pass
#endif
    double warm;  // qacc of the previous forward pass (mj qacc_warmstart): where the next constrained solve starts
    bool lim_on[2];
    double lim_D[2], lim_aref[2], lim_sign[2];
    // contact role
    bool c_on[KC];
    int c_b1[KC], c_b2[KC], c_dim[KC];
    double c_mu[KC], c_D[KC], c_kterm[KC], c_b[KC], c_jv[KC][3], c_jx[KC][3], c_jd[KC][3], c_gw[KC][8];
    // dual PGS (Sim<.., PGS = true>): forces of this lane's limit rows and contact rows, the 3x3 contact-frame block A = J_c M^-1 J_c^T
    // (A00 A01 A02 A11 A12 A22), the reciprocals 1 / (E A E^T + R) of its rows, J_c qacc_smooth, and where contacts beyond the LDS
    // capacity keep their M^-1 J_c^T block (global memory, per environment)
    double p_lf[2], p_lari[2], p_f[KC][4], p_A[KC][6], p_ari[KC][4], p_js[KC][3];
    int grp;  // which of the wavefront's sub-environments this lane belongs to (its blackboard is boards[grp])
};

template <class M, int G, bool PGS = (M::SOLVER == 1)>
struct Sim {
    typedef Board<M, G> B;
    typedef Lane<M, G> R;
    static constexpr int NQ = M::NQ, NV = M::NV, NB = M::NBODY, NU = M::NU, KC = B::KC, KS = B::KS, MAXCON = B::MAXCON;
#if defined(MJX_HOST_EMU)
#define MJX_RED(bb) (bb).red
#else
#define MJX_RED(bb) nullptr
#endif

    // entry j of this lane's mass-matrix row (registers or LDS, see Board::M_IN_LDS); j is always a compile-time constant
    static MJX_DEV double mrow(const B &bb, const R &r, int lane, int j) { return B::M_IN_LDS ? bb.Mt[j][lane < NV ? lane : 0] : r.Mrow[B::M_IN_LDS ? 0 : j]; }
    static MJX_DEV void set_mrow(B &bb, R &r, int lane, int j, double v) {
        if (B::M_IN_LDS) {
            if (lane < NV) bb.Mt[j][lane] = v;
        } else {
            r.Mrow[B::M_IN_LDS ? 0 : j] = v;
        }
    }

    // ---- one-time initialisation of the constant rows of the blackboard (world body) ----------------------------------
    static MJX_DEV void init(B &bb, int lane) {
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 3; k++) bb.xpos[0][k] = 0, bb.xipos[0][k] = 0;
#pragma unroll
            for (int k = 0; k < 6; k++) bb.cvel[0][k] = 0;
        }
        coop_sync();
    }

    // Everything static about a body's joints in ONE record (see SlotTab: body -> jntadr -> type / qposadr / dofadr -> qpos0 is a chain of
    // dependent per-lane loads from global memory in every phase that walks the joints; this is one level)
    struct BodyTab {
        struct Rec {
            int jn, jtype[M::MAXJPB], qadr[M::MAXJPB], dadr[M::MAXJPB];
            double jpos[M::MAXJPB][3], jaxis[M::MAXJPB][3], q0[M::MAXJPB];
        } b[NB];
    };
    static constexpr BodyTab make_bodies() {
        BodyTab t{};
        for (int b = 0; b < NB; b++) {
            const int ja = M::body_jntadr[b], jn = M::body_jntnum[b];
            t.b[b].jn = jn;
            for (int jj = 0; jj < M::MAXJPB; jj++) {
                const int j = jj < jn ? ja + jj : 0;
                t.b[b].jtype[jj] = jj < jn ? M::jnt_type[j] : -1;
                t.b[b].qadr[jj] = M::jnt_qposadr[j], t.b[b].dadr[jj] = M::jnt_dofadr[j], t.b[b].q0[jj] = M::qpos0[M::jnt_qposadr[j]];
                for (int k = 0; k < 3; k++) t.b[b].jpos[jj][k] = M::jnt_pos[j][k], t.b[b].jaxis[jj][k] = M::jnt_axis[j][k];
            }
        }
        return t;
    }
    static constexpr BodyTab kBody = make_bodies();
    // ... and about a dof: its actuator, passive terms, joint limit
    struct DofTab {
        struct Rec {
            int act, scalar_joint, limited, qadr, jnt;  // actuator (-1: none), hinge / slide, limited hinge / slide, qpos address, joint
            double gear, clo, chi, damping, stiffness, q0, range[2], invweight0;
        } d[NV];
    };
    static constexpr DofTab make_dofs() {
        DofTab t{};
        for (int i = 0; i < NV; i++) {
            const int u = M::dof_actuator[i], j = M::dof_jntid[i];
            const bool sc = M::jnt_type[j] == HINGE || M::jnt_type[j] == SLIDE;
            t.d[i].act = u, t.d[i].scalar_joint = sc, t.d[i].limited = sc && M::jnt_limited[j], t.d[i].qadr = M::jnt_qposadr[j], t.d[i].jnt = j;
            t.d[i].gear = u >= 0 ? M::actuator_gear[u] : 0.0, t.d[i].clo = u >= 0 ? M::actuator_ctrlrange[u][0] : 0.0, t.d[i].chi = u >= 0 ? M::actuator_ctrlrange[u][1] : 0.0;
            t.d[i].damping = M::dof_damping[i], t.d[i].stiffness = M::jnt_stiffness[j], t.d[i].q0 = M::qpos0[M::jnt_qposadr[j]];
            t.d[i].range[0] = M::jnt_range[j][0], t.d[i].range[1] = M::jnt_range[j][1], t.d[i].invweight0 = M::dof_invweight0[i];
        }
        return t;
    }
    static constexpr DofTab kDof = make_dofs();
    // joint jj of body b (1 = 0: through the model tables, for A/B runs)
    static MJX_DEV int jnum(int b) { return 1 ? kBody.b[b].jn : M::body_jntnum[b]; }
    static MJX_DEV int jtype(int b, int jj) { return 1 ? kBody.b[b].jtype[jj] : M::jnt_type[M::body_jntadr[b] + jj]; }
    static MJX_DEV int jqadr(int b, int jj) { return 1 ? kBody.b[b].qadr[jj] : M::jnt_qposadr[M::body_jntadr[b] + jj]; }
    static MJX_DEV int jdadr(int b, int jj) { return 1 ? kBody.b[b].dadr[jj] : M::jnt_dofadr[M::body_jntadr[b] + jj]; }
    static MJX_DEV double jq0(int b, int jj) { return 1 ? kBody.b[b].q0[jj] : M::qpos0[M::jnt_qposadr[M::body_jntadr[b] + jj]]; }
    static MJX_DEV const double *jpos(int b, int jj) { return 1 ? kBody.b[b].jpos[jj] : M::jnt_pos[M::body_jntadr[b] + jj]; }
    static MJX_DEV const double *jaxis(int b, int jj) { return 1 ? kBody.b[b].jaxis[jj] : M::jnt_axis[M::body_jntadr[b] + jj]; }

    // ancestors at distance 1, 2, 4, ... of every body (0 = the world: nothing left to compose), for the pointer-jumping kinematics
    struct AncTab {
        static constexpr int MAXR = 4;
        int ROUNDS;
        int a[MAXR][NB];
    };
    static constexpr AncTab make_anc() {
        AncTab t{};
        t.ROUNDS = 0;
        while ((1 << t.ROUNDS) < M::MAXDEPTH) t.ROUNDS++;
        for (int b = 0; b < NB; b++) {
            for (int rd = 0; rd < AncTab::MAXR; rd++) {
                int x = b;
                for (int s = 0; s < (1 << rd) && x > 0; s++) x = M::body_parentid[x];
                t.a[rd][b] = x;
            }
        }
        return t;
    }
    static constexpr AncTab kAnc = make_anc();
    static_assert(kAnc.ROUNDS <= AncTab::MAXR, "body tree deeper than 16 levels");

    // ---- position stage ---------------------------------------------------------------------------------------------------
    static MJX_DEV void kinematics(B &bb, R &r, int lane) {
        const int b = lane + 1;
        const bool isbody = b < NB;
        const int bi = isbody ? b : 1;
        const int depth = M::body_depth[bi], p = M::body_parentid[bi], ja = M::body_jntadr[bi], jn = jnum(bi);
        // (round 3) The joints of a body act in the body's own static frame F0 (parent pose o body_pos / body_quat): with pose = F0 o (pl, ql),
        //   anchor_j = F0 (pl + R(ql) jnt_pos_j),  axis_j = F0 R(ql) jnt_axis_j,  hinge: ql <- ql o q(axis_j, theta_j), pl <- anchor_j^loc - R(ql) jnt_pos_j,
        //   slide: pl += axis_j^loc theta_j
        // -- the same recursion mj_kinematics runs in world coordinates, factored.  (pl, ql) and the local anchors / axes depend on this body's
        // qpos only, so EVERY body evaluates its joint chain at once, before the level loop; a level then only composes the parent's pose with
        // F0 and maps the locals to the world.  Before, the whole joint chain (a sincos, two quaternion products and three rotations per
        // hinge, up to three hinges) sat inside the level loop, where a wavefront pays it once per LEVEL (six for the Humanoid) although each
        // body uses it once.  Rounding differs from the world-frame order at the 1e-16 level (tests: HIP / emulation vs oracle tolerances).
        const bool free_root = jn == 1 && jtype(bi, 0) == FREE;
        double ql[4] = {1, 0, 0, 0}, pl[3] = {0, 0, 0};
        if (isbody && !free_root) {
#pragma unroll
            for (int jj = 0; jj < M::MAXJPB; jj++) {
                if (jj < jn) {
                    const int j = ja + jj;
                    double Rm[9], qq[4], t[3];
                    quat_to_mat(Rm, ql);
                    rot_vec(t, Rm, jpos(bi, jj));
                    r.anchor[jj][0] = pl[0] + t[0], r.anchor[jj][1] = pl[1] + t[1], r.anchor[jj][2] = pl[2] + t[2];
                    rot_vec(r.axis[jj], Rm, jaxis(bi, jj));
                    const double q = bb.qpos[jqadr(bi, jj)] - jq0(bi, jj);
                    if (jtype(bi, jj) == HINGE) {
                        axis_angle_quat(qq, jaxis(bi, jj), q);
                        quat_mul(ql, ql, qq);
                        quat_to_mat(Rm, ql);
                        rot_vec(t, Rm, jpos(bi, jj));
                        pl[0] = r.anchor[jj][0] - t[0], pl[1] = r.anchor[jj][1] - t[1], pl[2] = r.anchor[jj][2] - t[2];
                    } else {
                        pl[0] += r.axis[jj][0] * q, pl[1] += r.axis[jj][1] * q, pl[2] += r.axis[jj][2] * q;
                    }
                }
            }
        }
        MJX_PHASE_X(r, 4, 12);
        // World poses by pointer jumping.  T_b starts as the body's pose in its parent's frame (static frame o joint chain; the free root: its
        // qpos) and a_b as the parent; a round replaces T_b by T_{a_b} o T_b and a_b by a_{a_b}, so after ceil(log2(MAXDEPTH)) rounds every T_b
        // is a world pose -- 3 rounds of one pose product for the Humanoid's 6 levels (2 for Ant's 4) instead of one pass over the whole
        // per-body work per LEVEL, which a lone wavefront pays in full each time however few bodies the level has.  The joints' anchors and
        // axes need the body's static frame in world coordinates: F0 = T_b o (pl, ql)^-1.  The product order differs from mj_kinematics'
        // (associativity): 1e-16-level rounding, and quaternions are normalised once at the end instead of per level.
        double pos[3] = {0, 0, 0}, quat[4] = {1, 0, 0, 0};
        if (isbody) {
            if (free_root) {
                const int qa = jqadr(bi, 0);
                pos[0] = bb.qpos[qa], pos[1] = bb.qpos[qa + 1], pos[2] = bb.qpos[qa + 2];
                quat[0] = bb.qpos[qa + 3], quat[1] = bb.qpos[qa + 4], quat[2] = bb.qpos[qa + 5], quat[3] = bb.qpos[qa + 6];
                quat_normalize(quat);
            } else {
                double RS[9], t[3];
                quat_to_mat(RS, M::body_quat[bi]);
                rot_vec(t, RS, pl);
                pos[0] = M::body_pos[bi][0] + t[0], pos[1] = M::body_pos[bi][1] + t[1], pos[2] = M::body_pos[bi][2] + t[2];
                quat_mul(quat, M::body_quat[bi], ql);
            }
        }
#pragma unroll
        for (int rd = 0; rd < kAnc.ROUNDS; rd++) {
            if (isbody) {
#pragma unroll
                for (int k = 0; k < 3; k++) bb.xpos[b][k] = pos[k];
#pragma unroll
                for (int k = 0; k < 4; k++) bb.A.kin.xquat[b][k] = quat[k];
            }
            coop_sync();
            const int a = kAnc.a[rd][bi];
            if (isbody && a != 0) {
                double Ra[9], t[3], qa[4];
#pragma unroll
                for (int k = 0; k < 4; k++) qa[k] = bb.A.kin.xquat[a][k];
                quat_to_mat(Ra, qa);
                rot_vec(t, Ra, pos);
                pos[0] = bb.xpos[a][0] + t[0], pos[1] = bb.xpos[a][1] + t[1], pos[2] = bb.xpos[a][2] + t[2];
                quat_mul(quat, qa, quat);
            }
            coop_sync();  // every read of the old poses before the next round (or the final values) overwrites them
        }
        MJX_PHASE_X(r, 4, 13);
        if (isbody) {
            quat_normalize(quat);
            double t[3];
            if (free_root) {
                r.anchor[0][0] = pos[0], r.anchor[0][1] = pos[1], r.anchor[0][2] = pos[2];
                r.axis[0][0] = 0, r.axis[0][1] = 0, r.axis[0][2] = 1;
            } else {
                double posS[3], quatS[4], RS[9];
                const double qc[4] = {ql[0], -ql[1], -ql[2], -ql[3]};
                quat_mul(quatS, quat, qc);
                quat_to_mat(RS, quatS);
                rot_vec(t, RS, pl);
                posS[0] = pos[0] - t[0], posS[1] = pos[1] - t[1], posS[2] = pos[2] - t[2];
#pragma unroll
                for (int jj = 0; jj < M::MAXJPB; jj++) {
                    if (jj < jn) {
                        rot_vec(t, RS, r.anchor[jj]);
                        r.anchor[jj][0] = posS[0] + t[0], r.anchor[jj][1] = posS[1] + t[1], r.anchor[jj][2] = posS[2] + t[2];
                        rot_vec(t, RS, r.axis[jj]);
                        r.axis[jj][0] = t[0], r.axis[jj][1] = t[1], r.axis[jj][2] = t[2];
                    }
                }
            }
            quat_to_mat(r.xmat, quat);
            rot_vec(t, r.xmat, M::body_ipos[bi]);
#pragma unroll
            for (int k = 0; k < 3; k++) bb.xpos[b][k] = pos[k], bb.xipos[b][k] = pos[k] + t[k];
#pragma unroll
            for (int k = 0; k < 4; k++) bb.A.kin.xquat[b][k] = quat[k];
#pragma unroll
            for (int k = 0; k < 9; k++) bb.A.kin.xmat[b][k] = r.xmat[k];
        }
        coop_sync();
    }

    // subtree centre of mass of the tree (every lane computes it: same summation order as the one-lane code), the body's
    // com-based inertia, the motion subspaces of the body's joints
    static MJX_DEV void com_pos(B &bb, R &r, int lane) {
        double mass = 0, c[3] = {0, 0, 0};
#pragma unroll
        for (int b = 1; b < NB; b++) {
            mass += M::body_mass[b];
            c[0] += M::body_mass[b] * bb.xipos[b][0], c[1] += M::body_mass[b] * bb.xipos[b][1], c[2] += M::body_mass[b] * bb.xipos[b][2];
        }
        const double com[3] = {c[0] / mass, c[1] / mass, c[2] / mass};
        if (lane == 0) bb.com[0] = com[0], bb.com[1] = com[1], bb.com[2] = com[2];
        const int b = lane + 1;
        if (b < NB) {
            const double *Rm = r.xmat, *I = M::body_inertia[b];
            const double off[3] = {bb.xipos[b][0] - com[0], bb.xipos[b][1] - com[1], bb.xipos[b][2] - com[2]};
            double T[9], W[9];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) T[3 * i + j] = Rm[3 * i] * I[j] + Rm[3 * i + 1] * I[3 + j] + Rm[3 * i + 2] * I[6 + j];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) W[3 * i + j] = T[3 * i] * Rm[3 * j] + T[3 * i + 1] * Rm[3 * j + 1] + T[3 * i + 2] * Rm[3 * j + 2];
            const double mm = M::body_mass[b], dd = dot3(off, off);
            double *ci = r.cinert;
            ci[0] = W[0] + mm * (dd - off[0] * off[0]), ci[1] = W[4] + mm * (dd - off[1] * off[1]), ci[2] = W[8] + mm * (dd - off[2] * off[2]);
            ci[3] = W[1] - mm * off[0] * off[1], ci[4] = W[2] - mm * off[0] * off[2], ci[5] = W[5] - mm * off[1] * off[2];
            ci[6] = mm * off[0], ci[7] = mm * off[1], ci[8] = mm * off[2], ci[9] = mm;
            const int jn = jnum(b);
#pragma unroll
            for (int jj = 0; jj < M::MAXJPB; jj++) {
                if (jj < jn) {
                    const int a = jdadr(b, jj), jt = jtype(b, jj);
                    const double o[3] = {com[0] - r.anchor[jj][0], com[1] - r.anchor[jj][1], com[2] - r.anchor[jj][2]};
                    if (jt == FREE) {
#pragma unroll
                        for (int k = 0; k < 3; k++) {
#pragma unroll
                            for (int c6 = 0; c6 < 6; c6++) bb.cdof[a + k][c6] = 0;
                            bb.cdof[a + k][3 + k] = 1.0;
                            const double ax[3] = {Rm[k], Rm[3 + k], Rm[6 + k]};
                            double cr[3];
                            cross3(cr, ax, o);
                            bb.cdof[a + 3 + k][0] = ax[0], bb.cdof[a + 3 + k][1] = ax[1], bb.cdof[a + 3 + k][2] = ax[2];
                            bb.cdof[a + 3 + k][3] = cr[0], bb.cdof[a + 3 + k][4] = cr[1], bb.cdof[a + 3 + k][5] = cr[2];
                        }
                    } else if (jt == HINGE) {
                        double cr[3];
                        cross3(cr, r.axis[jj], o);
                        bb.cdof[a][0] = r.axis[jj][0], bb.cdof[a][1] = r.axis[jj][1], bb.cdof[a][2] = r.axis[jj][2];
                        bb.cdof[a][3] = cr[0], bb.cdof[a][4] = cr[1], bb.cdof[a][5] = cr[2];
                    } else {
                        bb.cdof[a][0] = bb.cdof[a][1] = bb.cdof[a][2] = 0;
                        bb.cdof[a][3] = r.axis[jj][0], bb.cdof[a][4] = r.axis[jj][1], bb.cdof[a][5] = r.axis[jj][2];
                    }
                }
            }
        }
        coop_sync();
        if (lane < NV) {
#pragma unroll
            for (int k = 0; k < 6; k++) r.cdof[k] = bb.cdof[lane][k];
        }
    }

    // ---- velocity stage + RNE bias ----------------------------------------------------------------------------------------
    static MJX_DEV void com_vel_and_bias(B &bb, R &r, int lane) {
        const int b = lane + 1;
        const bool isbody = b < NB;
        const int bi = isbody ? b : 1;
        const int depth = M::body_depth[bi], p = M::body_parentid[bi], ja = M::body_jntadr[bi], jn = jnum(bi);
        // cvel and cacc live in ONE frame (the com-based world frame), so a body's value is its parent's plus the contributions of its own
        // joints: two prefix sums over the tree, done by pointer jumping (see kinematics()) -- first the velocities (a joint's cdof_dot needs the
        // velocity just before it), then the accelerations.  Each body walks its own joint chain ONCE instead of the wavefront walking the
        // longest chain once per level.  The sums associate differently from the per-level recursion: rounding only.
        double v[6] = {0, 0, 0, 0, 0, 0}, a[6] = {0, 0, 0, 0, 0, 0};
        if (isbody) {
#pragma unroll
            for (int jj = 0; jj < M::MAXJPB; jj++) {
                if (jj < jn) {
                    const int da = jdadr(bi, jj);
                    if (jtype(bi, jj) == FREE) {
#pragma unroll
                        for (int k = 0; k < 6; k++)
#pragma unroll
                            for (int c = 0; c < 6; c++) v[c] += bb.cdof[da + k][c] * bb.qvel[da + k];
                    } else {
#pragma unroll
                        for (int c = 0; c < 6; c++) v[c] += bb.cdof[da][c] * bb.qvel[da];
                    }
                }
            }
        }
#pragma unroll
        for (int rd = 0; rd < kAnc.ROUNDS; rd++) {
            if (isbody) {
#pragma unroll
                for (int k = 0; k < 6; k++) bb.cvel[b][k] = v[k];
            }
            coop_sync();
            const int an = kAnc.a[rd][bi];
            if (isbody && an != 0) {
#pragma unroll
                for (int k = 0; k < 6; k++) v[k] += bb.cvel[an][k];
            }
            coop_sync();
        }
        if (isbody) {
#pragma unroll
            for (int k = 0; k < 6; k++) bb.cvel[b][k] = v[k];
        }
        coop_sync();
        MJX_PHASE_X(r, 3, 12);
        if (isbody) {
            double vp[6] = {0, 0, 0, 0, 0, 0};  // the velocity the body's first joint sees: its parent's
            if (p == 0) {
                a[3] = -M::gravity[0], a[4] = -M::gravity[1], a[5] = -M::gravity[2];
            } else {
#pragma unroll
                for (int k = 0; k < 6; k++) vp[k] = bb.cvel[p][k];
            }
#pragma unroll
            for (int jj = 0; jj < M::MAXJPB; jj++) {
                if (jj < jn) {
                    const int da = jdadr(bi, jj);
                    if (jtype(bi, jj) == FREE) {
#pragma unroll
                        for (int k = 0; k < 3; k++)
#pragma unroll
                            for (int c = 0; c < 6; c++) vp[c] += bb.cdof[da + k][c] * bb.qvel[da + k];
                        double dd[3][6];
#pragma unroll
                        for (int k = 0; k < 3; k++) cross_motion(dd[k], vp, bb.cdof[da + 3 + k]);
#pragma unroll
                        for (int k = 0; k < 3; k++)
#pragma unroll
                            for (int c = 0; c < 6; c++) a[c] += dd[k][c] * bb.qvel[da + 3 + k];
                    } else {
                        double dd[6];
                        cross_motion(dd, vp, bb.cdof[da]);
#pragma unroll
                        for (int c = 0; c < 6; c++) vp[c] += bb.cdof[da][c] * bb.qvel[da], a[c] += dd[c] * bb.qvel[da];
                    }
                }
            }
        }
#pragma unroll
        for (int rd = 0; rd < kAnc.ROUNDS; rd++) {
            if (isbody) {
#pragma unroll
                for (int k = 0; k < 6; k++) bb.Bu.rne.cacc[b][k] = a[k];
            }
            coop_sync();
            const int an = kAnc.a[rd][bi];
            if (isbody && an != 0) {
#pragma unroll
                for (int k = 0; k < 6; k++) a[k] += bb.Bu.rne.cacc[an][k];
            }
            coop_sync();
        }
        if (isbody) {
            double Ia[6], Iv[6], x[6];
            inert_mul(Ia, r.cinert, a), inert_mul(Iv, r.cinert, v), cross_force(x, v, Iv);
#pragma unroll
            for (int k = 0; k < 6; k++) bb.Bu.rne.cacc[b][k] = a[k], bb.Bu.rne.cfrc[b][k] = Ia[k] + x[k];
        }
        coop_sync();
        MJX_PHASE_X(r, 3, 13);
        // bias force of dof i: its motion subspace against the summed body forces of the subtree it moves
        if (lane < NV) {
            const unsigned desc = (unsigned)M::dof_descbodies[lane];
            double f[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int d = NB - 1; d >= 1; d--) {
                const double w = ((desc >> d) & 1u) ? 1.0 : 0.0;  // predicated, not branched
#pragma unroll
                for (int k = 0; k < 6; k++) f[k] += w * bb.Bu.rne.cfrc[d][k];
            }
            double s = 0;
#pragma unroll
            for (int k = 0; k < 6; k++) s += r.cdof[k] * f[k];
            r.bias = s;
        }
        coop_sync();  // cfrc is dead from here on: crb() reuses its storage
    }

    // ---- composite rigid body: full row `lane` of the mass matrix in registers -----------------------------------------------
    // The blended form of the row loop lets the scheduler overlap the entries (Humanoid rows 8.3 k -> 6.4 k cycles, +2.1 % end to end; HalfCheetah,
    // Walker2d, Hopper +1 %).  The overlap costs registers: while physics16.hip was built with MachineLICM off, the 14-dof Ant kernel went from 21 to
    // 71 spilled VGPRs with it and lost 2.8 %; on the sink flag set (build.py) it has no spills either way and gains 0.6 %.  profiles/r03_collision_tables.txt
    static constexpr bool CRB_BLEND = true;  // (every robot; until round 6 a switch kept the select form for the Ant's A/B runs)
    static MJX_DEV void crb(B &bb, R &r, int lane) {
        const int b = lane + 1;
        if (b < NB) {
#pragma unroll
            for (int k = 0; k < 10; k++) bb.Bu.cinert[b][k] = r.cinert[k];
        }
        coop_sync();
        if (b < NB) {
            const unsigned desc = (unsigned)M::body_descmask[b];
            double c[10];
#pragma unroll
            for (int k = 0; k < 10; k++) c[k] = 0;
#pragma unroll
            for (int d = NB - 1; d >= 1; d--) {
                const double w = ((desc >> d) & 1u) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < 10; k++) c[k] += w * bb.Bu.cinert[d][k];
            }
#pragma unroll
            for (int k = 0; k < 10; k++) bb.C.crb.cinertc[b][k] = c[k];
        }
        coop_sync();
        MJX_PHASE_X(r, 2, 12);
        double mybuf[6] = {0, 0, 0, 0, 0, 0};
        if (lane < NV) {
            inert_mul(mybuf, bb.C.crb.cinertc[M::dof_bodyid[lane]], r.cdof);
#pragma unroll
            for (int k = 0; k < 6; k++) bb.C.crb.buf[lane][k] = mybuf[k];
        }
        coop_sync();
        MJX_PHASE_X(r, 2, 13);
        if (lane < NV) {
            const unsigned anc = (unsigned)M::dof_ancmask[lane];
#pragma unroll
            for (int j = 0; j < NV; j++) {
                // both candidates, then a select: as two branches per entry (2 NV exec-mask regions, each waiting out its own LDS reads before a
                // chain of six multiply-adds) this loop cost a lone wavefront ~9 k cycles for the Humanoid; the selected sum is the same, bit for bit
                double s1 = 0, s2 = 0;
#pragma unroll
                for (int k = 0; k < 6; k++) s1 += bb.cdof[j][k] * mybuf[k];
#pragma unroll
                for (int k = 0; k < 6; k++) s2 += r.cdof[k] * bb.C.crb.buf[j][k];
                // (blended arithmetically: written as selects, LLVM sinks the second dot product back into a conditional block, and a branch per
                //  entry keeps the scheduler from overlapping one entry's LDS reads with the previous entry's multiply-adds; the dot products
                //  are finite, so 1 * s + 0 * t is s)
                double s;
                if constexpr (CRB_BLEND) {
                    const double w1 = ((anc >> j) & 1u) ? 1.0 : 0.0, w2 = ((((unsigned)M::dof_ancmask[j] >> lane) & 1u) && j != lane) ? 1.0 : 0.0;  // j == lane is in both masks
                    s = w1 * s1 + w2 * s2;
                } else {
                    s = ((anc >> j) & 1u) ? s1 : ((((unsigned)M::dof_ancmask[j] >> lane) & 1u) ? s2 : 0.0);
                }
                if (j == lane) s += M::dof_armature[j];
                if (B::M_IN_LDS)
                    r.Hrow[j] = s;  // stored below, all at once: a store into the blackboard between two entries pins the next entry's reads behind it
                else
                    set_mrow(bb, r, lane, j, s);
            }
            if (B::M_IN_LDS) {
#pragma unroll
                for (int j = 0; j < NV; j++) set_mrow(bb, r, lane, j, r.Hrow[j]);
            }
        } else if (!B::M_IN_LDS) {
#pragma unroll
            for (int j = 0; j < NV; j++) set_mrow(bb, r, lane, j, 0.0);
        }
        coop_sync();  // buf is dead from here on: the solver reuses its storage
    }

    // ---- dense Cholesky with one matrix row per lane ---------------------------------------------------------------------
    // Lane-to-group broadcast of lane K's value: a DPP row_newbcast for 16-lane groups (two VALU moves, no LDS traffic, no wait),
    // a ds_bpermute for 32-lane groups; the host emulation goes through the blackboard.
    template <int K>
    static MJX_DEV double bcast(double v, B &bb, int lane) {
#if defined(MJX_HOST_EMU)
        bb.red[0][lane] = v;
        coop_sync();
        const double out = bb.red[0][K];
        coop_sync();
        return out;
#else
        (void)bb, (void)lane;
        if (G == 16) return dpp_mov<0x150 + (K & 15)>(v);
        double a, b;  // 32-lane group: pick the row that holds lane K, then broadcast inside the rows
        row_pair(v, a, b);
        return dpp_mov<0x150 + (K & 15)>(K < 16 ? a : b);
#endif
    }
    // the same with the source lane known only at run time (group-uniform): one ds_bpermute per 32-bit half -- the LDS crossbar without an
    // LDS location, so no write / fence / read round trip
    static MJX_DEV double bcast_from(double v, int src, B &bb, int lane) {
#if defined(MJX_HOST_EMU)
        bb.red[0][lane] = v;
        coop_sync();
        const double out = bb.red[0][src];
        coop_sync();
        return out;
#else
        (void)bb, (void)lane;
        return __shfl(v, src, G);
#endif
    }
    // A: row `lane` (entries j <= lane are used) -> L in place; idiag = 1 / L[lane][lane]; L also stored packed in bb.A.sol.L.
    // Right-looking, column by column: the pivot and the column below it come from the other lanes' registers by broadcast.
    template <int K, int J>
    static MJX_DEV void chol_update(B &bb, double *A, double ak, double t, int lane) {
        const double cj = bcast<J>(ak, bb, lane);  // A'[J][K], held by lane J
        A[J] -= t * cj;  // entries J > lane are never used: unmasked, they carry a finite, meaningless mirror image of the elimination
        if constexpr (J + 1 < NV) chol_update<K, J + 1>(bb, A, ak, t, lane);
    }
    template <int K>
    static MJX_DEV void chol_column(B &bb, double *A, double &idiag, int lane) {
        const double ak = A[K];
        double piv = bcast<K>(ak, bb, lane);
        piv = piv < kMinVal ? kMinVal : piv;
        const double inv = rsq(piv);
        const double lik = ak * inv;
        if (lane >= K) A[K] = lik;
        if (lane == K) idiag = inv;
        if constexpr (K + 1 < NV) {
            chol_update<K, K + 1>(bb, A, ak, lik * inv, lane);
#if !defined(MJX_HOST_EMU)
            __builtin_amdgcn_sched_barrier(0);  // keep the broadcasts of later columns from being hoisted (register pressure)
#endif
            chol_column<K + 1>(bb, A, idiag, lane);
        }
    }
    static MJX_DEV void chol_factor_bcast(B &bb, double *A, double &idiag, int lane) {
        chol_column<0>(bb, A, idiag, lane);
        if (lane < NV) {
#pragma unroll
            for (int j = 0; j < NV; j++)
                if (j <= lane) bb.A.sol.L[tri(lane, 0) + j] = A[j];
        }
        coop_sync();
    }
    // solves L L^T x = rhs (rhs = this lane's component); returns this lane's component, the full solution is left in vdir
    template <int K>
    static MJX_DEV void solve_forward(B &bb, const double *Lrow, double idiag, double &y, int lane) {
        if (lane == K) y = y * idiag;
        const double yk = bcast<K>(y, bb, lane);
        y -= ((lane > K && lane < NV) ? Lrow[K] : 0.0) * yk;
        if constexpr (K + 1 < NV) solve_forward<K + 1>(bb, Lrow, idiag, y, lane);
    }
    template <int K>
    static MJX_DEV void solve_backward(B &bb, double idiag, double &y, int lane) {
        const double lki = bb.A.sol.L[tri(K, 0) + (lane < K ? lane : 0)];  // L[K][lane]: column access through the packed copy
        if (lane == K) y = y * idiag;
        const double zk = bcast<K>(y, bb, lane);
        y -= (lane < K ? lki : 0.0) * zk;
        if constexpr (K > 0) solve_backward<K - 1>(bb, idiag, y, lane);
    }
    static MJX_DEV double chol_solve_bcast(B &bb, const double *Lrow, double idiag, double rhs, int lane) {
        double y = rhs;
        solve_forward<0>(bb, Lrow, idiag, y, lane);
        solve_backward<NV - 1>(bb, idiag, y, lane);
        if (lane < NV) bb.A.sol.vdir[lane] = y;
        coop_sync();
        return y;
    }

    // 32-lane groups (two DPP rows): the pivot column is exchanged through LDS instead -- broadcasting by ds_bpermute keeps ~2 NV values
    // in flight per column and made the Humanoid kernel spill (335 VGPRs against 71 this way).
    // A: row `lane` (entries j <= lane are used) -> L in place; idiag = 1 / L[lane][lane]; L also stored packed in bb.A.sol.L
    static MJX_DEV void chol_factor_lds(B &bb, double *A, double &idiag, int lane) {
        // The entry that becomes the NEXT pivot column is updated and published first, the rest of the row afterwards: the LDS round trip of
        // the exchange overlaps the row update instead of following it.  Entries j > lane are never used, so nothing is masked: they carry
        // the (finite, meaningless) mirror image of the elimination and cost no selects.
        if (lane < NV) bb.A.sol.col[0][lane] = A[0];
        coop_sync();
#pragma unroll
        for (int k = 0; k < NV; k++) {
            double (&col)[NV] = bb.A.sol.col[k & 1];
            double (&nxt)[NV] = bb.A.sol.col[(k + 1) & 1];
            if (lane >= k && lane < NV) {
                double piv = col[k];
                piv = piv < kMinVal ? kMinVal : piv;
                const double inv = rsq(piv);
                const double lik = A[k] * inv;
                A[k] = lik;
                if (lane == k) idiag = inv;
                const double t = lik * inv;
                if (k + 1 < NV) {
                    A[k + 1] -= t * col[k + 1];
                    nxt[lane] = A[k + 1];
                }
#pragma unroll
                for (int j = k + 2; j < NV; j++) A[j] -= t * col[j];
            }
            coop_sync();
        }
        if (lane < NV) {
#pragma unroll
            for (int j = 0; j < NV; j++)
                if (j <= lane) bb.A.sol.L[tri(lane, 0) + j] = A[j];
        }
        coop_sync();
    }
    // solves L L^T x = rhs (rhs = this lane's component); returns this lane's component, the full solution is left in vdir
    static MJX_DEV double chol_solve_lds(B &bb, const double *Lrow, double idiag, double rhs, int lane) {
        double y = rhs;
#pragma unroll
        for (int k = 0; k < NV; k++) {
            double (&col)[NV] = bb.A.sol.col[k & 1];
            if (lane == k) y = y * idiag, col[k] = y;
            coop_sync();
            y -= ((lane > k && lane < NV) ? Lrow[k] : 0.0) * col[k];
        }
#pragma unroll
        for (int k = NV - 1; k >= 0; k--) {
            double (&col)[NV] = bb.A.sol.col[k & 1];
            if (lane == k) y = y * idiag, col[k] = y, bb.A.sol.vdir[k] = y;
            coop_sync();
            y -= (lane < k ? bb.A.sol.L[tri(k, 0) + (lane < k ? lane : 0)] : 0.0) * col[k];
        }
        coop_sync();
        return y;
    }

    static MJX_DEV void chol_factor(B &bb, double *A, double &idiag, int lane) {
        if constexpr ((G == 16 && !0) || (G == 32 && !1))
            chol_factor_bcast(bb, A, idiag, lane);
        else
            chol_factor_lds(bb, A, idiag, lane);
    }
    static MJX_DEV double chol_solve(B &bb, const double *Lrow, double idiag, double rhs, int lane) {
        if constexpr ((G == 16 && !0) || (G == 32 && (!1 || 1)))
            return chol_solve_bcast(bb, Lrow, idiag, rhs, lane);
        else
            return chol_solve_lds(bb, Lrow, idiag, rhs, lane);
    }

    // ---- collision: one candidate slot per lane and round, order-preserving compaction -------------------------------------------
    static MJX_DEV void geom_pose(const B &bb, int g, double *pos, double *axis_z) {
        const int b = M::geom_bodyid[g];
        const double lz[3] = {M::geom_mat[g][2], M::geom_mat[g][5], M::geom_mat[g][8]};
        if (b == 0) {  // the world frame is the identity (its rows of the blackboard are not kept)
#pragma unroll
            for (int k = 0; k < 3; k++) pos[k] = M::geom_pos[g][k], axis_z[k] = lz[k];
            return;
        }
        double t[3];
        rot_vec(t, bb.A.kin.xmat[b], M::geom_pos[g]);
        pos[0] = bb.xpos[b][0] + t[0], pos[1] = bb.xpos[b][1] + t[1], pos[2] = bb.xpos[b][2] + t[2];
        rot_vec(axis_z, bb.A.kin.xmat[b], lz);
    }
    struct Cand {
        bool on;
        double dist, pos[3], frame[9];
    };
    static MJX_DEV void finish(Cand &c, int pair, double dist, const double *pos, const double *n, const double *tangent, bool flip) {
        c.on = dist < M::pair_margin[uniform_pairs() ? 0 : pair];
        c.dist = dist;
        const double sg = flip ? -1.0 : 1.0;
#pragma unroll
        for (int k = 0; k < 3; k++) c.pos[k] = pos[k], c.frame[k] = sg * n[k], c.frame[3 + k] = (tangent && !flip) ? tangent[k] : 0.0;
        make_frame(c.frame);
    }
    static MJX_DEV void sphere_pair(Cand &c, int pair, const double *p1, double r1, const double *p2, double r2, bool flip) {
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
        finish(c, pair, dist - r1 - r2, pos, n, nullptr, flip);
    }
    // which (type1 <= type2) geometry combinations occur among the model's candidate pairs: the others compile to nothing
    static constexpr bool has_pairs(int ta, int tb) {
        for (int p = 0; p < M::NPAIR; p++) {
            const int t1 = M::geom_type[M::pair_geom1[p]], t2 = M::geom_type[M::pair_geom2[p]];
            const int lo = t1 < t2 ? t1 : t2, hi = t1 < t2 ? t2 : t1;
            if (lo == ta && hi == tb) return true;
        }
        return false;
    }
    // Everything static about a candidate slot in ONE record: slot -> pair -> geoms -> type / size / body were four levels of dependent
    // per-lane table loads from global memory (a lone wavefront waits out each level), and every slot recomputed both geoms' world poses.
    struct SlotTab {
        static constexpr int NS = M::NSLOT > 0 ? M::NSLOT : 1;
        int gi[NS][4];     // geom of the lower type, the other geom, type1 | type2 << 4 | flip << 8 | sub << 9, pair | body1 << 16 | body2 << 24
        double sz[NS][4];  // r1, r2, h1, h2 (in the swapped order)
    };
    static constexpr SlotTab make_slots() {
        SlotTab t{};
        for (int s = 0; s < M::NSLOT; s++) {
            const int p = M::slot_pair[s];
            int g1 = M::pair_geom1[p], g2 = M::pair_geom2[p], flip = 0;
            if (M::geom_type[g1] > M::geom_type[g2]) {
                const int x = g1;
                g1 = g2, g2 = x, flip = 1;
            }
            t.gi[s][0] = g1, t.gi[s][1] = g2;
            t.gi[s][2] = M::geom_type[g1] | (M::geom_type[g2] << 4) | (flip << 8) | (M::slot_sub[s] << 9);
            t.gi[s][3] = p | (M::geom_bodyid[M::pair_geom1[p]] << 16) | (M::geom_bodyid[M::pair_geom2[p]] << 24);
            t.sz[s][0] = M::geom_size[g1][0], t.sz[s][1] = M::geom_size[g2][0], t.sz[s][2] = M::geom_size[g1][1], t.sz[s][3] = M::geom_size[g2][1];
        }
        return t;
    }
    static constexpr SlotTab kSlot = make_slots();
    // world pose of every geom, once per forward pass (lane g, in rounds of G)
    static MJX_DEV void geom_poses(B &bb, int lane) {
#pragma unroll
        for (int g0 = 0; g0 < M::NGEOM; g0 += G) {
            const int g = g0 + lane;
            if (g < M::NGEOM) {
                double pos[3], az[3];
                geom_pose(bb, g, pos, az);
#pragma unroll
                for (int k = 0; k < 3; k++) bb.Bu.geo[g][k] = pos[k], bb.Bu.geo[g][3 + k] = az[k];
            }
        }
        coop_sync();
    }
    // Measured (profiles/r03_collision_tables.txt): Humanoid (140 slots, five rounds + the second pass) 18.6 k -> 11.8 k cycles per forward pass;
    // the robots with one or two rounds lose 1 - 3 % to the extra pass and fence, so they keep the direct form.
    static constexpr bool COLLIDE_TABLES = M::NSLOT > 2 * G;
    static MJX_DEV void detect(const B &bb, int slot, Cand &c) {
        int g1, g2, t1, t2, p, sub;
        bool flip = false;
        double r1, r2, h1, h2, p1[3], z1[3], p2[3], z2[3];
        if constexpr (COLLIDE_TABLES) {
            const int fl = kSlot.gi[slot][2];
            g1 = kSlot.gi[slot][0], g2 = kSlot.gi[slot][1], p = kSlot.gi[slot][3] & 0xffff;
            t1 = fl & 15, t2 = (fl >> 4) & 15, sub = (fl >> 9) & 1, flip = (fl >> 8) & 1;
            r1 = kSlot.sz[slot][0], r2 = kSlot.sz[slot][1], h1 = kSlot.sz[slot][2], h2 = kSlot.sz[slot][3];
#pragma unroll
            for (int k = 0; k < 3; k++) p1[k] = bb.Bu.geo[g1][k], z1[k] = bb.Bu.geo[g1][3 + k], p2[k] = bb.Bu.geo[g2][k], z2[k] = bb.Bu.geo[g2][3 + k];
        } else {
            p = M::slot_pair[slot], sub = M::slot_sub[slot];
            g1 = M::pair_geom1[p], g2 = M::pair_geom2[p];
            if (M::geom_type[g1] > M::geom_type[g2]) {
                const int t = g1;
                g1 = g2, g2 = t, flip = true;
            }
            t1 = M::geom_type[g1], t2 = M::geom_type[g2];
            geom_pose(bb, g1, p1, z1), geom_pose(bb, g2, p2, z2);
            r1 = M::geom_size[g1][0], r2 = M::geom_size[g2][0], h1 = M::geom_size[g1][1], h2 = M::geom_size[g2][1];
        }
        c.on = false;
        if (t1 == PLANE) {
            if (has_pairs(PLANE, SPHERE) && (t2 == SPHERE || !has_pairs(PLANE, CAPSULE))) {
                double v[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, pos[3];
                const double dist = dot3(v, z1) - r2;
#pragma unroll
                for (int k = 0; k < 3; k++) pos[k] = p2[k] - z1[k] * (r2 + 0.5 * dist);
                finish(c, p, dist, pos, z1, nullptr, false);
            } else if (has_pairs(PLANE, CAPSULE)) {
                const double s = sub == 0 ? 1.0 : -1.0;
                double cc[3], v[3], pos[3];
#pragma unroll
                for (int k = 0; k < 3; k++) cc[k] = p2[k] + s * h2 * z2[k], v[k] = cc[k] - p1[k];
                const double dist = dot3(v, z1) - r2;
#pragma unroll
                for (int k = 0; k < 3; k++) pos[k] = cc[k] - z1[k] * (r2 + 0.5 * dist);
                finish(c, p, dist, pos, z1, z2, false);
            }
        } else if (has_pairs(SPHERE, SPHERE) && t1 == SPHERE && t2 == SPHERE) {
            sphere_pair(c, p, p1, r1, p2, r2, flip);
        } else if (has_pairs(SPHERE, CAPSULE) && t1 == SPHERE && t2 == CAPSULE) {
            double v[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
            double x = dot3(v, z2);
            x = x > h2 ? h2 : (x < -h2 ? -h2 : x);
            double cc[3] = {p2[0] + x * z2[0], p2[1] + x * z2[1], p2[2] + x * z2[2]};
            sphere_pair(c, p, p1, r1, cc, r2, flip);
        } else if (has_pairs(CAPSULE, CAPSULE) && t1 == CAPSULE && t2 == CAPSULE) {
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
            sphere_pair(c, p, c1, r1, c2, r2, flip);
        }
    }
    // pass 1: every lane tests its candidate slots and publishes a bit per hit; pass 2: the hits are recomputed and written at
    // their rank (slot order = the order in which the serial code emits contacts), truncated at MAXCON
    // which lanes of this group see `pred` (bit i = lane i): the wavefront's ballot, cut down to the group -- no LDS word, no fence
    static MJX_DEV unsigned group_ballot(bool pred, B &bb, int lane, int word) {
#if defined(MJX_HOST_EMU)
        if (pred) lds_or(&bb.cmask[word], 1u << lane);
        coop_sync();
        return bb.cmask[word];
#else
        (void)bb, (void)word;
        const unsigned long long w = __ballot(pred);
        const int first = (int)(threadIdx.x & 63u) - lane;  // wavefront lane of this group's lane 0
        return (unsigned)(w >> first) & (G == 32 ? 0xffffffffu : ((1u << G) - 1u));
#endif
    }
    // Contacts in MuJoCo's order (= slot order).  Two forms:
    //   few candidate slots (Ant 25, HalfCheetah 16: one or two rounds): ONE pass -- a round's active lanes are ranked by the group ballot, the
    //     rounds are sequential, so every contact can be written as soon as it is found (Ant: 5.8 k -> 4.4 k cycles per forward pass);
    //   many slots, few contacts (Humanoid: 140 slots in five rounds): a cheap pass that only asks every slot whether it is active (the compiler
    //     drops the contact point and frame from that call), then the full evaluation for the few active ones -- the one-pass form was measured
    //     at 29 k cycles against 18 k this way.
    static MJX_DEV void write_contact(B &bb, int idx, int slot, const Cand &c) {
        if constexpr (COLLIDE_TABLES) {
            bb.con_pair[idx] = kSlot.gi[slot][3];
        } else {
            const int p = M::slot_pair[slot];
            const int b1 = M::geom_bodyid[M::pair_geom1[p]], b2 = M::geom_bodyid[M::pair_geom2[p]];
            bb.con_pair[idx] = p | (b1 << 16) | (b2 << 24);
        }
        bb.con_dist[idx] = c.dist;
#pragma unroll
        for (int k = 0; k < 3; k++) bb.con_r[idx][k] = c.pos[k] - bb.com[k];
#pragma unroll
        for (int k = 0; k < 9; k++) bb.con_frame[idx][k] = c.frame[k];
    }
    static MJX_DEV void collision(B &bb, int lane) {
        if (lane < KS) bb.cmask[lane] = 0;
        if (lane == 0) bb.anyrow = 0, bb.limmask[0] = 0, bb.limmask[1] = 0;
        if constexpr (COLLIDE_TABLES)
            geom_poses(bb, lane);  // (ends with the fence the two stores above need)
        else
            coop_sync();
        int base = 0;
        if constexpr (M::NSLOT <= 2 * G) {
#pragma unroll 1
            for (int rd = 0; rd < KS; rd++) {
                const int slot = rd * G + lane;
                Cand c;
                c.on = false;
                if (slot < M::NSLOT) detect(bb, slot, c);
                const unsigned m = group_ballot(c.on, bb, lane, rd);
                if (c.on) {
                    const int idx = base + popc(m & ((1u << lane) - 1u));
                    if (idx < MAXCON) write_contact(bb, idx, slot, c);
                }
                base += popc(m);
            }
        } else {
            unsigned mine = 0;
#pragma unroll 1
            for (int rd = 0; rd < KS; rd++) {
                const int slot = rd * G + lane;
                if (slot < M::NSLOT) {
                    Cand c;
                    detect(bb, slot, c);
                    if (c.on) lds_or(&bb.cmask[rd], 1u << lane), mine |= 1u << rd;
                }
            }
            coop_sync();
#pragma unroll 1
            for (int rd = 0; rd < KS; rd++) {
                const unsigned m = bb.cmask[rd];
                if ((mine >> rd) & 1u) {
                    const int idx = base + popc(m & ((1u << lane) - 1u));
                    if (idx < MAXCON) {
                        Cand c;
                        detect(bb, rd * G + lane, c);
                        write_contact(bb, idx, rd * G + lane, c);
                    }
                }
                base += popc(m);
            }
        }
        if (lane == 0) bb.ncon = base < MAXCON ? base : MAXCON;
        coop_sync();
    }

    // velocity of the contact point of contact c under a per-body twist field ([ang; lin] about the tree com), in the contact
    // frame (body2 minus body1); the world body (index 0) does not move
    static MJX_DEV void contact_vel(const B &bb, const double (*field)[6], int c, int b1, int b2, double *v) {
        const double *rr = bb.con_r[c], *F = bb.con_frame[c];
        double w[3] = {0, 0, 0}, t[3];
        if (b2 > 0) {
            cross3(t, field[b2], rr);
            w[0] += field[b2][3] + t[0], w[1] += field[b2][4] + t[1], w[2] += field[b2][5] + t[2];
        }
        if (b1 > 0) {
            cross3(t, field[b1], rr);
            w[0] -= field[b1][3] + t[0], w[1] -= field[b1][4] + t[1], w[2] -= field[b1][5] + t[2];
        }
        v[0] = dot3(F, w), v[1] = dot3(F + 3, w), v[2] = dot3(F + 6, w);
    }
    // bb.C.sol.tw[b] = sum over the dofs on the path to body b of cdof_i vec_i
    static MJX_DEV void twist(B &bb, const double *vec, int lane) {
        const int b = lane + 1;
        if (b < NB) {
            const unsigned mask = (unsigned)M::body_dofmask[b];
            double t[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < NV; i++) {
                const double x = ((mask >> i) & 1u) ? vec[i] : 0.0;
#pragma unroll
                for (int k = 0; k < 6; k++) t[k] += bb.cdof[i][k] * x;
            }
#pragma unroll
            for (int k = 0; k < 6; k++) bb.C.sol.tw[b][k] = t[k];
        }
        coop_sync();
    }

    // ---- constraint rows: joint limits (dof role) and contact parameters (contact role) -----------------------------------------
    // Robots usually give every limited joint (and every contact pair) the same solref / solimp / margin (the MJCF defaults): then
    // the parameters are compile-time constants instead of per-lane table loads, and the stiffness / damping divisions fold away.
    static constexpr int first_limited_joint() {
        for (int j = 0; j < M::NJNT; j++)
            if (M::jnt_limited[j] && (M::jnt_type[j] == HINGE || M::jnt_type[j] == SLIDE)) return j;
        return 0;
    }
    static constexpr bool uniform_limits() {
        const int f = first_limited_joint();
        for (int j = 0; j < M::NJNT; j++) {
            if (!(M::jnt_limited[j] && (M::jnt_type[j] == HINGE || M::jnt_type[j] == SLIDE))) continue;
            if (M::jnt_margin[j] != M::jnt_margin[f] || M::jnt_solref[j][0] != M::jnt_solref[f][0] || M::jnt_solref[j][1] != M::jnt_solref[f][1])
                return false;
            for (int k = 0; k < 5; k++)
                if (M::jnt_solimp[j][k] != M::jnt_solimp[f][k]) return false;
        }
        return true;
    }
    static constexpr bool uniform_pairs() {
        for (int p = 1; p < M::NPAIR; p++) {
            if (M::pair_margin[p] != M::pair_margin[0] || M::pair_friction[p] != M::pair_friction[0] || M::pair_condim[p] != M::pair_condim[0] ||
                M::pair_solref[p][0] != M::pair_solref[0][0] || M::pair_solref[p][1] != M::pair_solref[0][1])
                return false;
            for (int k = 0; k < 5; k++)
                if (M::pair_solimp[p][k] != M::pair_solimp[0][k]) return false;
        }
        return M::NPAIR > 0;
    }
    static MJX_DEV void make_constraint(B &bb, R &r, int lane) {
        bool any = false;
        r.lim_on[0] = r.lim_on[1] = false;
        if (lane < NV) {
            const typename DofTab::Rec &dr = kDof.d[lane];
            const int j = dr.jnt;
            const bool limited = dr.limited;
            const double value = bb.qpos[dr.qadr], rng[2] = {dr.range[0], dr.range[1]}, invw = dr.invweight0;
            if (limited) {
#pragma unroll
                for (int sd = 0; sd < 2; sd++) {
                    const double side = sd == 0 ? -1.0 : 1.0;
                    const double dist = side * (rng[sd] - value);
                    constexpr int JF = first_limited_joint();
                    const int jp = uniform_limits() ? JF : j;  // constant index -> the table reads below fold into immediates
                    const double margin = M::jnt_margin[jp];
                    if (dist < margin) {
                        double k, b, imp, Rr;
                        row_params<M>(M::jnt_solref[jp], M::jnt_solimp[jp], dist, margin, invw, k, b, imp, Rr);
                        r.lim_on[sd] = true, r.lim_sign[sd] = -side, r.lim_D[sd] = 1.0 / Rr;
                        r.lim_aref[sd] = -b * (-side * bb.qvel[lane]) - k * imp * (dist - margin);
                        if (PGS) lds_or(&bb.limmask[sd], 1u << lane);
                        any = true;
                    }
                }
            }
        }
        const int ncon = bb.ncon;
#pragma unroll
        for (int kc = 0; kc < KC; kc++) {
            const int c = kc * G + lane;
            r.c_on[kc] = c < ncon;
            r.c_b1[kc] = r.c_b2[kc] = 0, r.c_dim[kc] = 1, r.c_mu[kc] = 0, r.c_D[kc] = 0, r.c_kterm[kc] = 0, r.c_b[kc] = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) r.c_jv[kc][k] = r.c_jx[kc][k] = r.c_jd[kc][k] = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) r.c_gw[kc][k] = 0;
            if (r.c_on[kc]) {
                const int packed = bb.con_pair[c], p = packed & 0xffff;
                const int pp = uniform_pairs() ? 0 : p;  // constant index -> immediates
                const int b1 = (packed >> 16) & 0xff, b2 = (packed >> 24) & 0xff;
                const double tran = M::body_invweight0[b1][0] + M::body_invweight0[b2][0], mu = M::pair_friction[pp];
                const bool pyramid = M::pair_condim[pp] > 1;
                double k, b, imp, Rr;
                row_params<M>(M::pair_solref[pp], M::pair_solimp[pp], bb.con_dist[c], M::pair_margin[pp], pyramid ? tran + mu * mu * tran : tran, k, b,
                              imp, Rr);
                if (pyramid) {
                    Rr = 2 * mu * mu * Rr;
                    if (Rr < kMinVal) Rr = kMinVal;
                }
                r.c_b1[kc] = b1, r.c_b2[kc] = b2, r.c_dim[kc] = pyramid ? 3 : 1, r.c_mu[kc] = mu;
                r.c_D[kc] = 1.0 / Rr, r.c_kterm[kc] = -k * imp * (bb.con_dist[c] - M::pair_margin[pp]), r.c_b[kc] = b;
                contact_vel(bb, bb.cvel, c, b1, b2, r.c_jv[kc]);  // cvel IS the twist field of qvel
                any = true;
            }
        }
        if (any) lds_or(&bb.anyrow, 1u);
        coop_sync();
    }

    // ---- primal Newton solver ---------------------------------------------------------------------------------------------
    // cost(x) = 1/2 (x - x_s)' M (x - x_s) + sum_rows 1/2 D min(0, J x - aref)^2
    // residual of edge e of a pyramidal contact: J_e x - aref_e, with J_e = J_n + sg J_t
    static MJX_DEV void edge(const R &r, int kc, int e, const double *jq, double &val, double &sg, int &t) {
        sg = (e & 1) ? -r.c_mu[kc] : r.c_mu[kc];
        t = 1 + e / 2;
        val = jq[0] + sg * jq[t];
    }
    static MJX_DEV double edge_aref(const R &r, int kc, double sg, int t) {
        return -r.c_b[kc] * (r.c_jv[kc][0] + sg * r.c_jv[kc][t]) + r.c_kterm[kc];
    }
    // contact role: g = sum_active D jar e (gw[0..2]) and W = sum_active D e e^T (gw[3..7] = W00 W01 W02 W11 W22) of this
    // lane's contacts at the current iterate (registers only)
    static MJX_DEV void contact_state(R &r) {
#pragma unroll
        for (int kc = 0; kc < KC; kc++) {
            if (r.c_on[kc]) {
                double g[3] = {0, 0, 0}, W[5] = {0, 0, 0, 0, 0};
                const double D = r.c_D[kc];
                if (r.c_dim[kc] == 1) {
                    const double jar = r.c_jx[kc][0] - (-r.c_b[kc] * r.c_jv[kc][0] + r.c_kterm[kc]);
                    if (jar < 0) g[0] = D * jar, W[0] = D;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        double v, sg;
                        int t;
                        edge(r, kc, e, r.c_jx[kc], v, sg, t);
                        const double jar = v - edge_aref(r, kc, sg, t);
                        if (jar < 0) {
                            const double Dj = D * jar;
                            g[0] += Dj, g[t] += Dj * sg;
                            W[0] += D, W[t] += D * sg, W[2 + t] += D * sg * sg;
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 3; k++) r.c_gw[kc][k] = g[k];
#pragma unroll
                for (int k = 0; k < 5; k++) r.c_gw[kc][3 + k] = W[k];
            }
        }
    }
    // dof role: column `lane` of the contact-frame Jacobian of contact c
    static MJX_DEV void jac_col(const B &bb, const R &r, int c, int lane, double *jcol) {
        const int packed = bb.con_pair[c], b1 = (packed >> 16) & 0xff, b2 = (packed >> 24) & 0xff;
        const int in1 = ((unsigned)M::body_dofmask[b1] >> lane) & 1u, in2 = ((unsigned)M::body_dofmask[b2] >> lane) & 1u;
        const double sg = (double)(in2 - in1);
        double t[3];
        cross3(t, r.cdof, bb.con_r[c]);
        const double v[3] = {r.cdof[3] + t[0], r.cdof[4] + t[1], r.cdof[5] + t[2]};
        const double *F = bb.con_frame[c];
        jcol[0] = sg * dot3(F, v), jcol[1] = sg * dot3(F + 3, v), jcol[2] = sg * dot3(F + 6, v);
    }
    // gradient (and, if hess, the lower Hessian row in r.Hrow) of the cost at the iterate x; Mdx = this lane's (M (x - x_smooth)).
    // One exchange per contact: its owner publishes (g, W), every dof lane its Jacobian column.
    static MJX_DEV double assemble(B &bb, R &r, double Mdx, double xi, bool hess, int lane) {
        double grad = Mdx;
        if (lane < NV) {
            if (hess) {
#pragma unroll
                for (int j = 0; j < NV; j++) r.Hrow[j] = mrow(bb, r, lane, j);
            }
#pragma unroll
            for (int sd = 0; sd < 2; sd++) {
                if (r.lim_on[sd]) {
                    const double jar = r.lim_sign[sd] * xi - r.lim_aref[sd];
                    if (jar < 0) {
                        grad += r.lim_sign[sd] * r.lim_D[sd] * jar;
                        if (hess) {
                            // NOTE (round 2, open): LLVM turns this chain of tests into a switch over POINTERS into the row (`phi ptr addrspace(5)` in
                            // the IR), which keeps nine elements of the row in scratch memory for the whole Newton solver: 336 scratch instructions in
                            // the 32-lane kernel.  The select form `Hrow[j] += (j == lane) ? D : 0.0` brings that down to 71 -- but together with the
                            // MachineLICM sinking flag of build.py the 32-lane Newton kernel then raised a GPU memory access fault
                            // (scripts/coop_phase_bench.hip humanoid-newton), so the change is NOT in: it needs its own bisection first.
#pragma unroll
                            for (int j = 0; j < NV; j++)
                                if (j == lane) r.Hrow[j] += r.lim_D[sd];
                        }
                    }
                }
            }
        }
        const int ncon = bb.ncon;
#pragma unroll 1
        for (int c = 0; c < ncon; c++) {
            double jcol[3] = {0, 0, 0};
            double (&J)[3][NV] = bb.C.sol.jc[c & 1];
            double (&gw)[8] = bb.C.sol.gw[c & 1];
            if (lane < NV) {
                jac_col(bb, r, c, lane, jcol);
                J[0][lane] = jcol[0], J[1][lane] = jcol[1], J[2][lane] = jcol[2];
            }
            if (lane == (c & (G - 1))) {
                const int kc = c / G;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    double v = r.c_gw[0][k];
#pragma unroll
                    for (int q = 1; q < KC; q++) v = (kc == q) ? r.c_gw[q][k] : v;
                    gw[k] = v;
                }
            }
            coop_sync();
            if (lane < NV) {
                grad += jcol[0] * gw[0] + jcol[1] * gw[1] + jcol[2] * gw[2];
                if (hess) {
                    const double t0 = gw[3] * jcol[0] + gw[4] * jcol[1] + gw[5] * jcol[2], t1 = gw[4] * jcol[0] + gw[6] * jcol[1],
                                 t2 = gw[5] * jcol[0] + gw[7] * jcol[2];
#pragma unroll
                    for (int j = 0; j < NV; j++) r.Hrow[j] += t0 * J[0][j] + t1 * J[1][j] + t2 * J[2][j];  // entries j > lane are never used
                }
            }
        }
        return grad;
    }
    // contact role: world-frame force of this lane's contacts from the current (g, W) registers (force = -sum_active D jar e)
    static MJX_DEV void publish_forces(B &bb, const R &r, int lane) {
#pragma unroll
        for (int kc = 0; kc < KC; kc++) {
            if (r.c_on[kc]) {
                const int c = kc * G + lane;
                const double *F = bb.con_frame[c], *g = r.c_gw[kc];
#pragma unroll
                for (int k = 0; k < 3; k++) bb.C.sol.con_F[c][k] = -(F[k] * g[0] + F[3 + k] * g[1] + F[6 + k] * g[2]);
            }
        }
    }

    // ---- dual projected Gauss-Seidel (humanoid.xml:8 `solver="PGS" iterations="50"`) ---------------------------------------------
    // What mj_solPGS does, laid out for a group of lanes.  Unknowns: forces f >= 0 of the unilateral rows (joint limits, frictionless
    // contacts, the 4 edges of a pyramidal contact), relaxed ONE ROW AT A TIME in MuJoCo's row order (limits in joint order, then
    // contacts in detection order): f_r <- max(0, f_r - res_r / AR_rr), res = (J M^-1 J^T + R) f + J qacc_smooth - aref, for at most
    // M::ITERATIONS sweeps or until a sweep's scaled cost improvement is below the tolerance 1e-8.  The iteration is sequential by
    // definition, so the parallelism is inside a row:
    //   * matrix-free in acceleration space: dof lane i carries a_i, a = qacc_smooth + M^-1 J^T f; res_r = J_r a - aref_r + R_r f_r;
    //   * M^-1 is formed once per forward pass (row i on lane i, in the registers of the Hessian row the Newton solver would use);
    //   * a limit row has J = +-e_i: its residual is local to lane i, its update adds column i of M^-1 times the broadcast step;
    //   * the rows of a contact share the 3 x NV contact-frame Jacobian J_c: v = J_c a is three group reductions, the (up to) four edge
    //     updates then happen on the owner lane alone in the 3-dim contact frame against the block A = J_c M^-1 J_c^T (v += A E^T d),
    //     and a += (M^-1 J_c^T) (sum of E^T d) closes the contact: one exchange per contact and sweep, not one per row;
    //   * M^-1 J_c^T (3 x NV per contact) is computed once per pass; the first BCAP contacts keep it in the LDS storage of M (dead after
    //     the factorisation), for later ones the sweeps apply M^-1 (register rows) to J_c^T dl directly (apply_b).
    // Warm start = mj's dual warmstart: the forces implied by qacc_warmstart (r.warm), dropped for zero if their dual cost is positive.
    // Where the blocks live (all of it storage that is dead while the sweeps run): the NV x NV words of M (NV / 3 blocks), then -- 1 --
    // the packed Cholesky factor (dead once M^-1 is formed: NTRI / (3 NV) blocks), the RNE / CRB scratch of union Bu, and B::NBX blocks of their own.
    typedef decltype(B::Bu) bb_union_b_t;
    static constexpr int BCAP0 = B::M_IN_LDS ? NV / 3 : 0;
    static constexpr bool MORE_BLOCKS = B::M_IN_LDS && PGS;
    static constexpr int BCAP1 = BCAP0 + (MORE_BLOCKS ? B::NTRI / (3 * NV) : 0);
    static constexpr int BCAP2 = BCAP1 + (MORE_BLOCKS ? (int)(sizeof(bb_union_b_t) / sizeof(double)) / (3 * NV) : 0);
    static constexpr int BCAP = BCAP2 + (MORE_BLOCKS ? B::NBX : 0);
    static MJX_DEV double *bblock(B &bb, int c) {  // c < BCAP, group-uniform
        if (c < BCAP0) return &bb.Mt[0][0] + (size_t)c * 3 * NV;
        if (c < BCAP1) return bb.A.sol.L + (size_t)(c - BCAP0) * 3 * NV;
        if (c < BCAP2) return reinterpret_cast<double *>(&bb.Bu) + (size_t)(c - BCAP1) * 3 * NV;
        if constexpr (B::NBX > 0) return bb.bx[c - BCAP2 < B::NBX ? c - BCAP2 : 0];
        return nullptr;
    }
    // (a global-memory overflow store was tried for the contacts beyond BCAP: ~800 cycles per visit; removed)
    static MJX_DEV void store_b(B &bb, int c, int lane, const double *b) {
        if constexpr (B::M_IN_LDS) {
            if (lane < NV && c < BCAP) {
                double *p = bblock(bb, c);
                p[lane] = b[0], p[NV + lane] = b[1], p[2 * NV + lane] = b[2];
            }
        }
    }
    // (M^-1 J_c^T dl)_lane for the frame-space force step dl of contact c: from the stored block, or -- for the contacts beyond the LDS
    // capacity -- as row `lane` of M^-1 (r.Hrow) times J_c^T dl, whose entries the dof lanes exchange through the blackboard.
    static MJX_DEV double apply_b(B &bb, const R &r, int c, int lane, const double *jcol, const double *dl) {
        if constexpr (B::M_IN_LDS) {
            if (c < BCAP) {  // group-uniform
                if (lane >= NV) return 0.0;
                const double *p = bblock(bb, c);
                return p[lane] * dl[0] + p[NV + lane] * dl[1] + p[2 * NV + lane] * dl[2];
            }
        }
        double (&q)[NV] = bb.A.sol.col[0];
        if (lane < NV) q[lane] = jcol[0] * dl[0] + jcol[1] * dl[1] + jcol[2] * dl[2];
        coop_sync();
        double da = 0;
        if (lane < NV) {
#pragma unroll
            for (int j = 0; j < NV; j++) da += r.Hrow[j] * q[j];
        }
        return da;
    }
    // Row `lane` of M^-1 (= column `lane`: solve L L^T x = e_lane) from the packed factor on the blackboard.  Every lane runs its OWN forward
    // and back substitution: the factor entries are read at the same LDS address by all lanes (broadcast reads), the iterate stays in
    // registers, and there is no exchange and no fence between the rows -- 2 x NV (NV - 1) / 2 multiply-adds per lane.  (Round 2 first formed
    // L^-1 cooperatively row by row through the storage of M and multiplied L^-T L^-1: 20 k cycles per forward pass, and inlined it drove the
    // 32-lane kernel from 19 to 540 spilled registers.)
    static MJX_DEV void invert(B &bb, double idiag, int lane, double *x) {
        double (&dg)[NV] = bb.A.sol.col[0];
        if (lane < NV) dg[lane] = idiag;
        coop_sync();
        const int me = lane < NV ? lane : 0;
#pragma unroll
        for (int j = 0; j < NV; j++) {
            double s = (j == me) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < j; k++) s -= bb.A.sol.L[tri(j, 0) + k] * x[k];
            x[j] = s * dg[j];
            MJX_SCHED_FENCE();  // keep the scheduler from hoisting the LDS reads of later rows (hundreds of live values -> spills)
        }
#pragma unroll
        for (int j = NV - 1; j >= 0; j--) {
            double s = x[j];
#pragma unroll
            for (int k = j + 1; k < NV; k++) s -= bb.A.sol.L[tri(k, 0) + j] * x[k];
            x[j] = s * dg[j];
            MJX_SCHED_FENCE();
        }
        coop_sync();  // dg (the solver's exchange column) is reused by the sweeps
    }
    // The rows of contact (kc, owner lane) against the contact-frame acceleration v = J_c a: relax them in order, keep v current, return the
    // frame-space force step dl = sum E^T delta and the cost improvement.  Runs on every lane (SIMD), meaningful on the owner.
    static MJX_DEV void pgs_contact(R &r, int kc, double *v, double *dl, double &impr) {
        const double Rr = 1.0 / r.c_D[kc];
        const double *A = r.p_A[kc];
        dl[0] = dl[1] = dl[2] = 0;
        if (r.c_dim[kc] == 1) {
            const double aref = -r.c_b[kc] * r.c_jv[kc][0] + r.c_kterm[kc];
            const double f0 = r.p_f[kc][0], res = (v[0] - aref) + Rr * f0;
            double nw = f0 - res * r.p_ari[kc][0];
            nw = nw < 0 ? 0.0 : nw;
            const double dd = nw - f0;
            r.p_f[kc][0] = nw, impr -= dd * (0.5 * dd * (A[0] + Rr) + res);
            v[0] += A[0] * dd, v[1] += A[1] * dd, v[2] += A[2] * dd, dl[0] = dd;
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const double sg = (e & 1) ? -r.c_mu[kc] : r.c_mu[kc];
                const int t = 1 + e / 2;
                const double A0t = t == 1 ? A[1] : A[2], Att = t == 1 ? A[3] : A[5], Aot = A[4];  // A[other][t] = A12 either way
                const double AR = (A[0] + 2 * sg * A0t + sg * sg * Att) + Rr;
                const double fe = r.p_f[kc][e], res = ((v[0] + sg * v[t]) - edge_aref(r, kc, sg, t)) + Rr * fe;
                double nw = fe - res * r.p_ari[kc][e];
                nw = nw < 0 ? 0.0 : nw;
                const double dd = nw - fe;
                r.p_f[kc][e] = nw, impr -= dd * (0.5 * dd * AR + res);
                // v += A (e_0 + sg e_t) dd
                v[0] += (A[0] + sg * A0t) * dd;
                v[t] += (A0t + sg * Att) * dd;
                v[3 - t] += ((t == 1 ? A[2] : A[1]) + sg * Aot) * dd;
                dl[0] += dd, dl[t] += sg * dd;
            }
        }
    }
    // limit rows of dof I (static: column I of M^-1 is entry I of every lane's register row), lower side then upper side
    template <int I, bool WARM>
    static MJX_DEV void pgs_limits(B &bb, R &r, int lane, unsigned lm0, unsigned lm1, double &a, double &qf, double &impr) {
        if constexpr (M::jnt_limited[M::dof_jntid[I]] && (M::jnt_type[M::dof_jntid[I]] == HINGE || M::jnt_type[M::dof_jntid[I]] == SLIDE)) {
            if (((lm0 | lm1) >> I) & 1u)  // (one scalar test per limited dof and sweep while neither side is active)
#pragma unroll
            for (int sd = 0; sd < 2; sd++) {
                if (((sd == 0 ? lm0 : lm1) >> I) & 1u) {  // group-uniform
                    double step = 0;  // sign * delta of this row, on lane I
                    if (lane == I) {
                        const double s = r.lim_sign[sd], f0 = r.p_lf[sd];
                        if (WARM) {  // a, qf accumulate M^-1 J^T f and J^T f of the warm-start forces
                            step = s * f0, qf += s * f0;
                        } else {
                            const double Rr = 1.0 / r.lim_D[sd], res = (s * a - r.lim_aref[sd]) + Rr * f0;
                            double nw = f0 - res * r.p_lari[sd];
                            nw = nw < 0 ? 0.0 : nw;
                            const double dd = nw - f0;
                            r.p_lf[sd] = nw, impr -= dd * (0.5 * dd / r.p_lari[sd] + res), step = s * dd;
                        }
                    }
                    const double bs = bcast<I>(step, bb, lane);
                    a += r.Hrow[I] * bs;
                }
            }
        }
        if constexpr (I + 1 < NV) pgs_limits<I + 1, WARM>(bb, r, lane, lm0, lm1, a, qf, impr);
    }
    // in: r.qfrc_smooth, the constraint rows of make_constraint(), r.warm.  out: r.qacc, r.qacc_smooth, contact forces on the blackboard.
    static MJX_DEV void pgs(B &bb, R &r, int lane, bool anyrow) {
        const bool isdof = lane < NV;
#pragma unroll
        for (int j = 0; j < NV; j++) r.Hrow[j] = mrow(bb, r, lane, j);
        MJX_PHASE_X(r, 1, 12);
        chol_factor(bb, r.Hrow, r.idiag, lane);
        MJX_PHASE_X(r, 1, 13);
        const double qs = chol_solve(bb, r.Hrow, r.idiag, isdof ? r.qfrc_smooth : 0.0, lane);
        r.qacc_smooth = qs, r.qfrc_constraint = 0;
        MJX_PHASE(r, 8);
        if (!anyrow) {
            r.qacc = qs;
            return;
        }
        // contact-frame images of qacc_smooth (vdir, left there by the solve) and of the warm start
        double jw[KC][3];
        twist(bb, bb.A.sol.vdir, lane);
#pragma unroll
        for (int kc = 0; kc < KC; kc++) {
            r.p_js[kc][0] = r.p_js[kc][1] = r.p_js[kc][2] = 0;
            if (r.c_on[kc]) contact_vel(bb, bb.C.sol.tw, kc * G + lane, r.c_b1[kc], r.c_b2[kc], r.p_js[kc]);
        }
        coop_sync();
        if (isdof) bb.A.sol.vdir[lane] = r.warm;
        coop_sync();
        twist(bb, bb.A.sol.vdir, lane);
#pragma unroll
        for (int kc = 0; kc < KC; kc++) {
            jw[kc][0] = jw[kc][1] = jw[kc][2] = 0;
            if (r.c_on[kc]) contact_vel(bb, bb.C.sol.tw, kc * G + lane, r.c_b1[kc], r.c_b2[kc], jw[kc]);
        }
        coop_sync();
        MJX_PHASE(r, 9);
        {
            double minv[NV];
            invert(bb, r.idiag, lane, minv);
#pragma unroll
            for (int j = 0; j < NV; j++) r.Hrow[j] = minv[j];  // r.Hrow = row `lane` of M^-1
        }
        MJX_PHASE(r, 7);
        const unsigned lm0 = bb.limmask[0], lm1 = bb.limmask[1];
        const int ncon = bb.ncon;
        // ---- rows: diagonal, warm-start force, dual cost pieces --------------------------------------------------------------------
        double cost = 0;  // this lane's share of sum_r f_r (R_r f_r / 2 + b_r)
        if (isdof) {
            double mii = 0;
#pragma unroll
            for (int j = 0; j < NV; j++) mii = (j == lane) ? r.Hrow[j] : mii;
#pragma unroll
            for (int sd = 0; sd < 2; sd++) {
                r.p_lf[sd] = 0, r.p_lari[sd] = 0;
                if (r.lim_on[sd]) {
                    const double s = r.lim_sign[sd], Rr = 1.0 / r.lim_D[sd];
                    r.p_lari[sd] = 1.0 / (mii + Rr);
                    const double jar = s * r.warm - r.lim_aref[sd];
                    const double f0 = jar < 0 ? -r.lim_D[sd] * jar : 0.0;
                    r.p_lf[sd] = f0, cost += f0 * (0.5 * Rr * f0 + (s * qs - r.lim_aref[sd]));
                }
            }
        } else {
            r.p_lf[0] = r.p_lf[1] = 0, r.p_lari[0] = r.p_lari[1] = 0;
        }
        double w = 0, qf = 0, dummy = 0;  // (M^-1 J^T f)_lane and (J^T f)_lane of the warm-start forces
        pgs_limits<0, true>(bb, r, lane, lm0, lm1, w, qf, dummy);
#pragma unroll
        for (int kc = 0; kc < KC; kc++) {
#pragma unroll 1
            for (int owner = 0; owner < G; owner++) {
                const int c = kc * G + owner;
                if (c >= ncon) break;
                double jcol[3] = {0, 0, 0};
                double (&J)[3][NV] = bb.C.sol.jc[c & 1];
                double (&gw)[8] = bb.C.sol.gw[c & 1];
                if (isdof) {
                    jac_col(bb, r, c, lane, jcol);
                    J[0][lane] = jcol[0], J[1][lane] = jcol[1], J[2][lane] = jcol[2];
                }
                if (lane == owner) {  // warm-start forces of the contact's rows and their frame-space sum
                    const double D = r.c_D[kc], Rr = 1.0 / D;
                    double lam[3] = {0, 0, 0};
#pragma unroll
                    for (int e = 0; e < 4; e++) r.p_f[kc][e] = 0;
                    if (r.c_dim[kc] == 1) {
                        const double aref = -r.c_b[kc] * r.c_jv[kc][0] + r.c_kterm[kc], jar = jw[kc][0] - aref;
                        const double f0 = jar < 0 ? -D * jar : 0.0;
                        r.p_f[kc][0] = f0, lam[0] = f0, cost += f0 * (0.5 * Rr * f0 + (r.p_js[kc][0] - aref));
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const double sg = (e & 1) ? -r.c_mu[kc] : r.c_mu[kc];
                            const int t = 1 + e / 2;
                            const double aref = edge_aref(r, kc, sg, t), jar = (jw[kc][0] + sg * jw[kc][t]) - aref;
                            const double fe = jar < 0 ? -D * jar : 0.0;
                            r.p_f[kc][e] = fe, lam[0] += fe, lam[t] += sg * fe;
                            cost += fe * (0.5 * Rr * fe + ((r.p_js[kc][0] + sg * r.p_js[kc][t]) - aref));
                        }
                    }
                    gw[0] = lam[0], gw[1] = lam[1], gw[2] = lam[2];
                }
                coop_sync();
                double b[3] = {0, 0, 0};
                if (isdof) {
#pragma unroll
                    for (int j = 0; j < NV; j++) b[0] += r.Hrow[j] * J[0][j], b[1] += r.Hrow[j] * J[1][j], b[2] += r.Hrow[j] * J[2][j];
                }
                store_b(bb, c, lane, b);
                // A = J_c (M^-1 J_c^T): six reductions, kept by the owner together with the reciprocal row diagonals
                const double a00 = group_sum<G>(jcol[0] * b[0], MJX_RED(bb), lane), a01 = group_sum<G>(jcol[0] * b[1], MJX_RED(bb), lane),
                             a02 = group_sum<G>(jcol[0] * b[2], MJX_RED(bb), lane), a11 = group_sum<G>(jcol[1] * b[1], MJX_RED(bb), lane),
                             a12 = group_sum<G>(jcol[1] * b[2], MJX_RED(bb), lane), a22 = group_sum<G>(jcol[2] * b[2], MJX_RED(bb), lane);
                if (lane == owner) {
                    double *A = r.p_A[kc];
                    A[0] = a00, A[1] = a01, A[2] = a02, A[3] = a11, A[4] = a12, A[5] = a22;
                    const double Rr = 1.0 / r.c_D[kc];
                    if (r.c_dim[kc] == 1) {
                        r.p_ari[kc][0] = 1.0 / (a00 + Rr);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const double sg = (e & 1) ? -r.c_mu[kc] : r.c_mu[kc];
                            const double A0t = e < 2 ? a01 : a02, Att = e < 2 ? a11 : a22;
                            r.p_ari[kc][e] = 1.0 / ((a00 + 2 * sg * A0t + sg * sg * Att) + Rr);
                        }
                    }
                }
                w += b[0] * gw[0] + b[1] * gw[1] + b[2] * gw[2];
                qf += jcol[0] * gw[0] + jcol[1] * gw[1] + jcol[2] * gw[2];
            }
        }
        // dual cost of the warm start: f'(A + R) f / 2 + f' b = (J^T f)' (M^-1 J^T f) / 2 + sum_r f_r (R_r f_r / 2 + b_r)
        const double wcost = group_sum<G>(cost + 0.5 * qf * w, MJX_RED(bb), lane);
        double a = qs;
        if (wcost > 0) {  // the warm start is worse than zero forces: start from f = 0, a = qacc_smooth
            r.p_lf[0] = r.p_lf[1] = 0;
#pragma unroll
            for (int kc = 0; kc < KC; kc++)
#pragma unroll
                for (int e = 0; e < 4; e++) r.p_f[kc][e] = 0;
        } else {
            a += w;
        }
        MJX_PHASE(r, 6);
        // ---- sweeps ---------------------------------------------------------------------------------------------------------------------
        const double scale = 1.0 / (M::MEANINERTIA * (NV > 1 ? NV : 1));
#pragma unroll 1
        for (int it = 0; it < M::ITERATIONS; it++) {
            double impr = 0, unused = 0;
            pgs_limits<0, false>(bb, r, lane, lm0, lm1, a, unused, impr);
#pragma unroll
            for (int kc = 0; kc < KC; kc++) {
#pragma unroll 1
                for (int owner = 0; owner < G; owner++) {
                    const int c = kc * G + owner;
                    if (c >= ncon) break;
                    double jcol[3] = {0, 0, 0}, v[3], dl[3];
                    if (isdof) jac_col(bb, r, c, lane, jcol);
                    v[0] = group_sum<G>(jcol[0] * a, MJX_RED(bb), lane), v[1] = group_sum<G>(jcol[1] * a, MJX_RED(bb), lane),
                    v[2] = group_sum<G>(jcol[2] * a, MJX_RED(bb), lane);
                    dl[0] = dl[1] = dl[2] = 0;
                    if (lane == owner) pgs_contact(r, kc, v, dl, impr);
                    // the owner's frame-space force step to every lane of the group: a cross-lane read, not an LDS exchange
                    const double step[3] = {bcast_from(dl[0], owner, bb, lane), bcast_from(dl[1], owner, bb, lane), bcast_from(dl[2], owner, bb, lane)};
                    a += apply_b(bb, r, c, lane, jcol, step);
                }
            }
            const double imp = group_sum<G>(impr, MJX_RED(bb), lane);
#if defined(MJX_HOST_EMU)
            if (lane == 0) g_stat[3]++;
#endif
            if (imp * scale < 1e-8) break;
        }
        r.qacc = a;
        MJX_PHASE(r, 10);
        // world-frame contact forces for cfrc_ext: frame^T (sum f, mu (f0 - f1), mu (f2 - f3))  (mj_contactForce for pyramids)
#pragma unroll
        for (int kc = 0; kc < KC; kc++) {
            if (r.c_on[kc]) {
                const int c = kc * G + lane;
                const double *F = bb.con_frame[c], *f = r.p_f[kc], mu = r.c_mu[kc];
                const double lf0 = r.c_dim[kc] == 1 ? f[0] : ((f[0] + f[1]) + f[2]) + f[3], lf1 = r.c_dim[kc] == 1 ? 0.0 : (f[0] - f[1]) * mu,
                             lf2 = r.c_dim[kc] == 1 ? 0.0 : (f[2] - f[3]) * mu;
#pragma unroll
                for (int k = 0; k < 3; k++) bb.C.sol.con_F[c][k] = F[k] * lf0 + F[3 + k] * lf1 + F[6 + k] * lf2;
            }
        }
        coop_sync();
    }

    // ---- forward dynamics -------------------------------------------------------------------------------------------------
    // in: bb.qpos, bb.qvel, bb.ctrl.  out (dof lanes): r.qacc, r.qacc_int (the acceleration the Euler integrator uses: implicit
    // in the joint damping), r.qfrc_actuator; on the blackboard: poses, cvel, contact frames and world-frame contact forces.
    // All linear solves (M for the unconstrained acceleration, the Newton Hessians, M + h B for the damped Euler update) go
    // through ONE factor/solve site driven by a small state machine, which keeps the kernel inside the instruction cache.
    enum { ST_SMOOTH = 0, ST_NEWTON = 1, ST_DAMPED = 2, ST_DONE = 3 };
    static constexpr bool damped_euler() {
        bool d = false;
        for (int i = 0; i < NV; i++) d = d || M::dof_damping[i] > 0;
        return d && M::INTEGRATOR == 0;
    }
    static MJX_DEV void forward(B &bb, R &r, int lane) {
        const bool isdof = lane < NV;
        MJX_PHASE(r, 0);
        if constexpr (M::NTENDON > 0) {  // lane t: tendon t at this pass's qpos / qvel (read by nobody before write_extras)
            if (lane < M::NTENDON) {
                double l = 0, v = 0;
#pragma unroll
                for (int k = 0; k < M::MAXWRAP; k++)
                    if (k < M::tendon_num[lane])
                        l += M::tendon_coef[lane][k] * bb.qpos[M::tendon_qposadr[lane][k]], v += M::tendon_coef[lane][k] * bb.qvel[M::tendon_dofadr[lane][k]];
                bb.ten[lane] = l, bb.ten[M::NTENDON + lane] = v;
            }
        }
        kinematics(bb, r, lane);
        MJX_PHASE(r, 1);
        com_pos(bb, r, lane);
        MJX_PHASE(r, 2);
        collision(bb, lane);
        MJX_PHASE(r, 3);
        com_vel_and_bias(bb, r, lane);
        MJX_PHASE(r, 4);
        crb(bb, r, lane);
        MJX_PHASE(r, 5);
        make_constraint(bb, r, lane);
        MJX_PHASE(r, 6);
        if (isdof) {
            double act = 0.0, passive;
            const typename DofTab::Rec &dr = kDof.d[lane];
            if (dr.act >= 0) {
                double c = bb.ctrl[dr.act];
                c = c < dr.clo ? dr.clo : (c > dr.chi ? dr.chi : c);
                act = dr.gear * c;
            }
            passive = -dr.damping * bb.qvel[lane];
            if (dr.scalar_joint) passive -= dr.stiffness * (bb.qpos[dr.qadr] - dr.q0);
            r.qfrc_actuator = act;
            r.qfrc_smooth = passive - r.bias + act;
        } else {
            r.qfrc_smooth = 0, r.qfrc_actuator = 0;
        }
        const bool anyrow = bb.anyrow != 0;
#if defined(MJX_HOST_EMU)
        if (lane == 0) g_stat[0]++, g_stat[1] += anyrow;
#endif
        if constexpr (PGS) {  // the model's own solver (humanoid.xml:8); r.warm is advanced by step(): MuJoCo saves qacc_warmstart once per mj_step
            static_assert(!damped_euler(), "PGS models of the family integrate with RK4");
            pgs(bb, r, lane, anyrow);
            r.qacc_int = r.qacc;
            coop_sync();
            MJX_PHASE(r, 11);
            return;
        }
        const double scale = 1.0 / (M::MEANINERTIA * (NV > 1 ? NV : 1));
        constexpr double h = M::TIMESTEP;
        int st = ST_SMOOTH, it = 0;
        bool final_pass = false;
        double x = 0, Mdx = 0, grad = 0;
        r.qacc = 0, r.qacc_int = 0, r.qfrc_constraint = 0, r.qacc_smooth = 0;
        // With constraint rows present the unconstrained acceleration is never needed by itself: the gradient of the cost is
        // M x - qfrc_smooth + J^T(...), so Newton starts straight from the warm start (the previous pass's qacc) and the separate
        // factorisation of M is skipped -- one factor/solve less per forward pass (of 2.4), fewer iterations near the solution.
        bool warm_start = anyrow;
        if (warm_start) {
            if (isdof) bb.A.sol.vdir[lane] = r.warm;
            coop_sync();
        }
#ifdef MJX_COUNT_WORK
        int passes = 0, lsn = 0;
#endif
#pragma unroll 1
        for (;;) {
#ifdef MJX_COUNT_WORK
            passes++;
#endif
            double rhs = 0;
            bool solve = true;
            if (st == ST_SMOOTH) {
#pragma unroll
                for (int j = 0; j < NV; j++) r.Hrow[j] = mrow(bb, r, lane, j);
                rhs = r.qfrc_smooth;
                solve = !warm_start;
            } else if (st == ST_NEWTON) {
                MJX_PHASE(r, 11);
                contact_state(r);
                grad = assemble(bb, r, Mdx, x, !final_pass, lane);
                MJX_PHASE(r, 7);
                bool finished = final_pass;
                if (!finished) {
                    const double gn = group_sum<G>(isdof ? grad * grad : 0.0, MJX_RED(bb), lane);
                    finished = sqrt(gn) * scale < 1e-10;
                }
                if (finished) {
                    r.qacc = x, r.qfrc_constraint = Mdx - grad;  // J^T f = -(grad - M dx)
                    publish_forces(bb, r, lane);
                    st = damped_euler() ? ST_DAMPED : ST_DONE;
                    solve = false;
                }
                rhs = isdof ? -grad : 0.0;
            } else if (st == ST_DAMPED) {
#pragma unroll
                for (int j = 0; j < NV; j++) r.Hrow[j] = mrow(bb, r, lane, j) + ((j == lane) ? h * M::dof_damping[j] : 0.0);
                rhs = isdof ? r.qfrc_smooth + r.qfrc_constraint : 0.0;
            } else {
                break;
            }
            if (!solve && !(st == ST_SMOOTH && warm_start)) continue;
            double sol = r.warm;
            MJX_PHASE(r, 11);
            if (solve) {
                chol_factor(bb, r.Hrow, r.idiag, lane);
                sol = chol_solve(bb, r.Hrow, r.idiag, rhs, lane);
            }
            MJX_PHASE(r, 8);
            if (st == ST_DAMPED) {
                r.qacc_int = sol;
                st = ST_DONE;
                continue;
            }
            if (st == ST_SMOOTH && !anyrow) {
                r.qacc_smooth = sol, r.qacc = sol, r.qfrc_constraint = 0;
                st = damped_euler() ? ST_DAMPED : ST_DONE;
                continue;
            }
            // contact-space image of the solution vector (the unconstrained acceleration, or a Newton direction)
            twist(bb, bb.A.sol.vdir, lane);
#pragma unroll
            for (int kc = 0; kc < KC; kc++)
                if (r.c_on[kc]) contact_vel(bb, bb.C.sol.tw, kc * G + lane, r.c_b1[kc], r.c_b2[kc], r.c_jd[kc]);
            MJX_PHASE(r, 9);
            if (st == ST_SMOOTH) {  // the iterate starts at sol (the unconstrained acceleration, or the warm start): x = 0 + 1 * sol
                r.qacc_smooth = sol, x = sol, Mdx = 0;
                if (warm_start) {  // M (x - x_smooth) = M x - qfrc_smooth
                    double Mx = 0;
                    if (isdof) {
#pragma unroll
                        for (int j = 0; j < NV; j++) Mx += mrow(bb, r, lane, j) * bb.A.sol.vdir[j];
                    }
                    Mdx = isdof ? Mx - r.qfrc_smooth : 0.0;
                    warm_start = false;
                }
#pragma unroll
                for (int kc = 0; kc < KC; kc++)
#pragma unroll
                    for (int k = 0; k < 3; k++) r.c_jx[kc][k] = r.c_jd[kc][k];
                st = ST_NEWTON;
                coop_sync();
                continue;
            }
            // line search along dir = sol: phi'(alpha) = g0 + alpha h0 + sum_active D (jar + alpha jd) jd
            const double dir = sol;
            double Md = 0;
            if (isdof) {
#pragma unroll
                for (int j = 0; j < NV; j++) Md += mrow(bb, r, lane, j) * bb.A.sol.vdir[j];
            }
            const double h0 = group_sum<G>(isdof ? dir * Md : 0.0, MJX_RED(bb), lane);
            const double g0 = group_sum<G>(isdof ? dir * Mdx : 0.0, MJX_RED(bb), lane);
            double alpha = 0, lo = 0, hi = INFINITY;
#if defined(MJX_HOST_EMU)
            if (lane == 0) g_stat[2]++;
#endif
#pragma unroll 1
            for (int ls = 0; ls < 40; ls++) {
#ifdef MJX_COUNT_WORK
                lsn++;
#endif
#if defined(MJX_HOST_EMU)
                if (lane == 0) g_stat[3]++;
#endif
                double gl = 0, hl = 0;
                if (isdof) {
#pragma unroll
                    for (int sd = 0; sd < 2; sd++) {
                        if (r.lim_on[sd]) {
                            const double jd = r.lim_sign[sd] * dir;
                            const double v = (r.lim_sign[sd] * x - r.lim_aref[sd]) + alpha * jd;
                            if (v < 0) gl += r.lim_D[sd] * v * jd, hl += r.lim_D[sd] * jd * jd;
                        }
                    }
                }
#pragma unroll
                for (int kc = 0; kc < KC; kc++) {
                    if (r.c_on[kc]) {
                        const double D = r.c_D[kc];
                        if (r.c_dim[kc] == 1) {
                            const double jar = r.c_jx[kc][0] - (-r.c_b[kc] * r.c_jv[kc][0] + r.c_kterm[kc]), jd = r.c_jd[kc][0];
                            const double v = jar + alpha * jd;
                            if (v < 0) gl += D * v * jd, hl += D * jd * jd;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                double jx, jd, sg;
                                int t;
                                edge(r, kc, e, r.c_jx[kc], jx, sg, t);
                                edge(r, kc, e, r.c_jd[kc], jd, sg, t);
                                const double v = (jx - edge_aref(r, kc, sg, t)) + alpha * jd;
                                if (v < 0) gl += D * v * jd, hl += D * jd * jd;
                            }
                        }
                    }
                }
                const double g = (g0 + alpha * h0) + group_sum<G>(gl, MJX_RED(bb), lane);
                const double hh = h0 + group_sum<G>(hl, MJX_RED(bb), lane);
                if (fabs(g) <= 1e-14 * (fabs(g0) + 1e-300)) break;
                if (g < 0)
                    lo = alpha;
                else
                    hi = alpha;
                double next = alpha - g / hh;
                if (!(next > lo && next < hi)) next = hi < INFINITY ? 0.5 * (lo + hi) : 2 * alpha + 1.0;
                if (next == alpha) break;
                alpha = next;
            }
            MJX_PHASE(r, 10);
            if (!(alpha > 0)) {  // no descent possible: the iterate (and the gradient just assembled) is final
                r.qacc = x, r.qfrc_constraint = Mdx - grad;
                publish_forces(bb, r, lane);
                st = damped_euler() ? ST_DAMPED : ST_DONE;
                coop_sync();
                continue;
            }
            x += alpha * dir, Mdx += alpha * Md;
#pragma unroll
            for (int kc = 0; kc < KC; kc++)
#pragma unroll
                for (int k = 0; k < 3; k++) r.c_jx[kc][k] += alpha * r.c_jd[kc][k];
            it++;
            const double move = group_sum<G>(isdof ? fabs(alpha * dir) : 0.0, MJX_RED(bb), lane);
            if (move * scale < 1e-16 || it >= 50) final_pass = true;  // one more assembly for the forces at the final iterate
            coop_sync();
        }
#if defined(MJX_COUNT_WORK) && !defined(MJX_HOST_EMU)
        {
            const int cost = 5 * passes + lsn;  // ~ cycles / 850: a pass of the state machine (assembly, factor / solve, twist) ~ 4 - 5 k cycles, a line-search iteration ~ 0.85 k
            int m = cost;
            m = max(m, __shfl_xor(m, 16, 64)), m = max(m, __shfl_xor(m, 32, 64));  // the other sub-environments of this wavefront
            r.work += cost, r.work_wave += m;
        }
#endif
        if (!damped_euler()) r.qacc_int = r.qacc;
        r.warm = r.qacc;
        coop_sync();
        MJX_PHASE(r, 11);
    }

    // joints of body `lane + 1`: position update of bb.qpos from the dof velocities in `vel` (mj_integratePos)
    static MJX_DEV void integrate_pos(B &bb, const double *vel, double h, int lane) {
        const int b = lane + 1;
        if (b < NB) {
            const int jn = jnum(b);
#pragma unroll
            for (int jj = 0; jj < M::MAXJPB; jj++) {
                if (jj < jn) {
                    const int qa = jqadr(b, jj), va = jdadr(b, jj);
                    if (jtype(b, jj) == FREE) {
                        bb.qpos[qa] += h * vel[va], bb.qpos[qa + 1] += h * vel[va + 1], bb.qpos[qa + 2] += h * vel[va + 2];
                        double w[3] = {vel[va + 3], vel[va + 4], vel[va + 5]}, qr[4], q[4] = {bb.qpos[qa + 3], bb.qpos[qa + 4], bb.qpos[qa + 5], bb.qpos[qa + 6]};
                        const double ang = h * normalize3(w);
                        axis_angle_quat(qr, w, ang);
                        quat_normalize(q);
                        quat_mul(q, q, qr);
                        bb.qpos[qa + 3] = q[0], bb.qpos[qa + 4] = q[1], bb.qpos[qa + 5] = q[2], bb.qpos[qa + 6] = q[3];
                    } else {
                        bb.qpos[qa] += h * vel[va];
                    }
                }
            }
        }
        coop_sync();
    }

    // One stage of the RK4 tableau, as selects on the stage index: sub-diagonal (0.5, 0.5, 1) and weights (1/6, 1/3, 1/3, 1/6).
    // Deliberately OUT OF LINE on the device.  Round 2: inlined into the stage loop, LLVM's iterative schedulers (build.py TU_FLAGS) produced wrong
    // code for exactly this block in the 16-lane instantiation (every environment differed after one sub-step).  Round 3 re-tested it on the
    // rewritten sources (-DMJX_RK4_INLINE=1, scripts/r03/gpu_call28.sh, gpu_call29.sh): the iterative-scheduler build is now bit-identical to the
    // out-of-line default-scheduler build on all four 16-lane robots and 1 - 2 % faster -- but the DEFAULT-scheduler build of the inlined form
    // (libmi355env_ref.so) makes the Ant kernel die with HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION.  So the failure follows the inlining, not the
    // scheduler flag: some address computation of the 16-lane Ant instantiation goes wrong when this block is merged into the stage loop, under either
    // scheduler, and which build shows it changes with the surrounding code.  Stand-alone reproducer with default flags: scripts/repro/README.md -- it goes
    // away with -mllvm -amdgpu-spill-sgpr-to-vgpr=false or -disable-machine-licm: an SGPR spilled into a VGPR lane comes back wrong (profiles/r03_rk4_inline.txt).
    // As a function of its own the kernel is bit-identical across schedulers; four calls per sub-step cost ~1 % next to four forward passes.
#if defined(MJX_HOST_EMU)
    static inline void rk4_stage(B &bb, double qacc, int lane, int i, double &v0, double &sumv, double &suma) {
#elif 0
    static MJX_DEV void rk4_stage(B &bb, double qacc, int lane, int i, double &v0, double &sumv, double &suma) {
#else
    static __device__ __attribute__((noinline)) void rk4_stage(B &bb, double qacc, int lane, int i, double &v0, double &sumv, double &suma) {
#endif
        constexpr double h = M::TIMESTEP;
        const bool isdof = lane < NV;
        if (i == 0) {
            if (isdof) v0 = bb.qvel[lane];
            for (int k = lane; k < NQ; k += G) bb.q0[k] = bb.qpos[k];
        }
        const double fv = isdof ? bb.qvel[lane] : 0.0, fa = qacc;
        const double bw = (i == 0 || i == 3) ? 1.0 / 6 : 1.0 / 3, anext = i == 2 ? 1.0 : 0.5;
        if (i == 0)
            sumv = bw * fv, suma = bw * fa;
        else
            sumv += bw * fv, suma += bw * fa;
        const bool last = i == 3;
        const double dv = last ? sumv : anext * fv, da = last ? suma : anext * fa;
        coop_sync();  // every lane has read the stage's qpos / qvel
        if (isdof) bb.dv[lane] = dv, bb.qvel[lane] = v0 + h * da;
        if (i > 0)
            for (int k = lane; k < NQ; k += G) bb.qpos[k] = bb.q0[k];
        coop_sync();
        integrate_pos(bb, bb.dv, h, lane);
    }

    // one mj_step: semi-implicit Euler (one forward pass) or RK4 (four), through a single forward() call site
    static MJX_DEV void step(B &bb, R &r, int lane) {
        constexpr double h = M::TIMESTEP;
        constexpr int NSTAGE = M::INTEGRATOR == 0 ? 1 : 4;
        const bool isdof = lane < NV;
        double v0 = 0, sumv = 0, suma = 0;  // RK4: the first stage's velocity and the weighted sums (rk4_stage)
#pragma unroll 1
        for (int i = 0; i < NSTAGE; i++) {
            forward(bb, r, lane);
            if (M::INTEGRATOR == 0) {
                if (isdof) bb.qvel[lane] += h * r.qacc_int;
                coop_sync();
                integrate_pos(bb, bb.qvel, h, lane);
            } else {
                rk4_stage(bb, r.qacc, lane, i, v0, sumv, suma);
            }
        }
        if constexpr (PGS) r.warm = r.qacc;  // mj_advance: "save qacc for next step warmstart" -- the LAST forward pass's qacc, once per step
    }

    // mj_rnePostConstraint, the part the envs read: cfrc_ext of body `lane + 1` from the contact forces of the LAST forward pass
    static MJX_DEV void contact_force_of_body(const B &bb, int lane, double *out) {
        const int b = lane + 1;
#pragma unroll
        for (int k = 0; k < 6; k++) out[k] = 0;
        const int ncon = bb.ncon;
#pragma unroll 1
        for (int c = 0; c < ncon; c++) {
            const int packed = bb.con_pair[c], b1 = (packed >> 16) & 0xff, b2 = (packed >> 24) & 0xff;
            if (b1 != b && b2 != b) continue;
            const double *Fw = bb.C.sol.con_F[c], *rr = bb.con_r[c];
            double tq[3];
            cross3(tq, rr, Fw);
            const double sg = (b2 == b ? 1.0 : 0.0) - (b1 == b ? 1.0 : 0.0);
#pragma unroll
            for (int k = 0; k < 3; k++) out[k] += sg * tq[k], out[3 + k] += sg * Fw[k];
        }
    }

    // What the per-env glue (mjx_kernels.h) reads from the last forward pass besides qpos / qvel, one row per environment:
    //   [0..1] world position (x, y) of body 1, [2..3] sum_b mass_b xipos_b (x, y)  (humanoid_v5.py:17-21 mass_center numerator),
    //   cfrc_ext[NB][6], cinert[NB][10], cvel[NB][6], qfrc_actuator[NV], ten_length[NTENDON], ten_velocity[NTENDON]
    static constexpr int EX_XY = 0, EX_CFRC = 4, EX_CINERT = EX_CFRC + 6 * NB, EX_CVEL = EX_CINERT + 10 * NB, EX_QFA = EX_CVEL + 6 * NB,
                         EX_TEN = EX_QFA + NV, EX_TOTAL = EX_TEN + 2 * M::NTENDON;
    // WHAT: which parts the env glue of this robot reads -- bit 0: cfrc_ext (Ant's contact cost / observation, the Humanoids), bit 1: cinert,
    // cvel, qfrc_actuator, tendons (the Humanoids' observation and infos).  HalfCheetah needs the tracked point only: writing the other
    // 2.6 KB per env-step anyway was a third of the Ant kernel's HBM writes.
    template <int WHAT = 3>
    static MJX_DEV void write_extras(const B &bb, const R &r, int lane, double *ex) {
        if (lane == 0) {
            ex[EX_XY] = bb.xpos[1][0], ex[EX_XY + 1] = bb.xpos[1][1];
            double nx = 0, ny = 0;
#pragma unroll
            for (int b = 0; b < NB; b++) nx += M::body_mass[b] * bb.xipos[b][0], ny += M::body_mass[b] * bb.xipos[b][1];
            ex[EX_XY + 2] = nx, ex[EX_XY + 3] = ny;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                if (WHAT & 1) ex[EX_CFRC + k] = 0;
                if (WHAT & 2) ex[EX_CVEL + k] = 0;
            }
            if (WHAT & 2) {
#pragma unroll
                for (int k = 0; k < 10; k++) ex[EX_CINERT + k] = 0;
            }
        }
        const int b = lane + 1;
        if (b < NB) {
            if (WHAT & 1) {
                double f[6];
                contact_force_of_body(bb, lane, f);
#pragma unroll
                for (int k = 0; k < 6; k++) ex[EX_CFRC + 6 * b + k] = f[k];
            }
            if (WHAT & 2) {
#pragma unroll
                for (int k = 0; k < 6; k++) ex[EX_CVEL + 6 * b + k] = bb.cvel[b][k];
#pragma unroll
                for (int k = 0; k < 10; k++) ex[EX_CINERT + 10 * b + k] = r.cinert[k];
            }
        }
        if (!(WHAT & 2)) return;
        if (lane < NV) ex[EX_QFA + lane] = r.qfrc_actuator;
        if constexpr (M::NTENDON > 0)
            if (lane < 2 * M::NTENDON) ex[EX_TEN + lane] = bb.ten[lane];
    }
#undef MJX_RED
};

}  // namespace coop
}  // namespace mjx

#if !defined(MJX_HOST_EMU)
#pragma clang fp contract(off)
#endif
