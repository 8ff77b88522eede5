// engine.hip -- libmi355env.so: lockstep vector-environment kernels for MI355X (gfx950) and the C ABI of
// include/mi355env.h.
//
// Execution model: one sub-environment per lane, 256-thread workgroups (4 wavefronts of 64, one per SIMD of a CU);
// at num_envs = 65536 the grid is 256 workgroups = one per CU.  Sub-environment state is struct-of-arrays in HBM
// (component-major float64, so a wavefront's 64 loads of one component are one contiguous 512-byte segment), the
// API-typed outputs (obs rows, reward, flags) are written straight from registers as contiguous per-wavefront
// segments.  TimeLimit, the autoreset state machine, the per-env PCG64 streams and the episode-return bookkeeping
// all live on device; a step() is ONE kernel launch, a rollout(T) is ONE launch for T steps with the state held
// in registers in between.  There is no CPU fallback anywhere in this file.
//
// Reference semantics reproduced (paths relative to the reference tree):
//   vector/sync_vector_env.py:187-337  SyncVectorEnv.reset/step incl. NEXT_STEP / SAME_STEP / DISABLED autoreset
//   wrappers/common.py:116-150         TimeLimit
//   wrappers/vector/common.py:156-235  RecordEpisodeStatistics (r, l)
//   envs/classic_control/*.py          dynamics (envs_classic.h)
//   utils/seeding.py:10-42             per-env Generator(PCG64(SeedSequence(seed + i))) (pcg64_dev.h)
//   spaces/multi_discrete.py:176-178, spaces/box.py:463-465   action_space.sample() (rollout with on-device policy)
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <new>
#include <string>
#include <vector>

// TWO translation units are made of this file (gymnasium_amd/csrc/build.py): engine.o (everything but the classic-control kernels) and, with
// -DMI_CLASSIC_TU (classic.hip), classic.o: the step / reset / rollout / epilogue kernels of the five classic-control kinds and their launchers
// (namespace mi_classic), compiled with LLVM's max-ILP machine scheduler.  Those kernels run ONE wavefront per SIMD at the benchmark's 65 536
// sub-environments, where nothing hides a dependent instruction's latency but the wavefront's own independent instructions: max-ILP scheduling
// measured +6.4 % (CartPole), +5.8 % (Pendulum), +2.7 % (MountainCarContinuous) against the default max-occupancy scheduler, with every result bit
// unchanged (scheduling reorders, it does not re-associate; tests/test_gpu_parity.py compares array_equal) -- and -1 ... -2 % on the ToyText and
// one-lane MuJoCo kernels, which therefore stay in engine.o on the default scheduler (profiles/r04_maxilp_classic.txt).
#include "../../include/mi355env.h"
#include "envs_classic.h"
#include "wrappers_internal.h"
#include "pcg64_dev.h"
#ifndef MI_CLASSIC_TU
#include "mjx_kernels.h"
#include "mjx_coop.h"
#include "mjx_physics.h"
#endif

using namespace mi;

namespace {

constexpr int kBlock = 256;  // 4 wavefronts: one per SIMD of a CU
constexpr int kErrInvalidAction = 1, kErrDisabledStepped = 2;
constexpr uint32_t kFlagShift = 30, kElapsedMask = (1u << kFlagShift) - 1u;

}  // namespace
// Plain types that cross the boundary between the two translation units (launcher arguments, members of mi_vecenv): a named namespace, so that
// mi_classic's functions have external linkage.
namespace mi_internal {
// Transition table of a tabular (ToyText) environment, device pointers (mi_tabular_load).
struct TabTable {
    int nS, nA, K;
    const double *csprob, *prob, *reward, *isd;
    const int32_t *next, *count;
    const uint8_t *term;
    const int32_t *env_table;  // [N] table of each sub-environment when there are several (mi_tabular_table.num_tables > 1), else nullptr
};

// Device view of one vector environment (passed by value as a kernel argument).
struct DevEnv {
    double *state;       // [S][N]   physics state, component-major
    uint32_t *meta;      // [N]      TimeLimit elapsed steps (bits 0..29) | flags (bits 30..31)
    uint64_t *rng;       // [4][N]   PCG64 {state_hi, state_lo, inc_hi, inc_lo}
    double *ep_ret;      // [N]      running episode return
    int32_t *ep_len;     // [N]      running episode length
    uint64_t *blk_count; // [grid][4] per-workgroup totals: env_steps, reset_steps, episodes, length_sum
    double *blk_ret;     // [grid]    per-workgroup sum of finished-episode returns
    int *error;          // sticky device error word
    int N;
    int max_steps;
    int solver_newton;   // MI_CFG_SOLVER_NEWTON (one-lane MuJoCo kernels; the cooperative kernel is chosen at launch)
    int tab_fickle_rows; // MI_ENV_TABULAR with params[2] != 0 (Taxi's fickle passenger): the state has a third row (TabLane::aux)
    EnvParams P;
    TabTable tab;
};
}  // namespace mi_internal
using namespace mi_internal;
namespace {

struct LaneStats {
    uint32_t env_steps, reset_steps, episodes;
    uint64_t length_sum;
    double return_sum;
};

template <class E>
struct Lane {
    double s[E::S];
    uint32_t elapsed, flags;
    double ep_ret;
    int32_t ep_len;
    typename E::Trig trig;  // sin / cos of the state's angles from the last obs() (envs_classic.h), registers only
};

template <class E>
MI_DEV void load_lane(const DevEnv &d, int i, Lane<E> &L) {
#pragma unroll
    for (int k = 0; k < E::S; k++) L.s[k] = d.state[(size_t)k * d.N + i];
    const uint32_t m = d.meta[i];
    L.elapsed = m & kElapsedMask, L.flags = m >> kFlagShift;
    L.ep_ret = d.ep_ret[i], L.ep_len = d.ep_len[i];
    trig_invalidate(L.trig);
}
// step_kernel's form of load_lane + load_rng: the raw words are REQUESTED here and looked at only after the math tables are staged (arrive()),
// held opaque in between -- the compiler otherwise starts on them at once (the first 128-bit multiply of the generator is speculated out of
// the reset branch), i.e. waits for this trip to memory before it requests the tables: two trips in a row where one does.
template <class T>
MI_DEV void hold_opaque(T &x) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "one or two VGPRs");
    asm volatile("" : "+v"(x));
}
template <class E>
struct LaneRequest {
    double s[E::S], ep_ret;
    uint32_t meta;
    int32_t ep_len;
    uint64_t g[4];
    uint64_t ag[2];  // SAMPLE: the lane's state of the action stream (mi_step with actions == NULL)
    typename E::Act a;
    template <bool SAMPLE>
    MI_DEV void request(const DevEnv &d, const void *actions, const uint64_t *act_lane, int i, bool with_generator) {
#pragma unroll
        for (int k = 0; k < E::S; k++) s[k] = d.state[(size_t)k * d.N + i];
        meta = d.meta[i], ep_ret = d.ep_ret[i], ep_len = d.ep_len[i];
        if constexpr (SAMPLE)
            ag[0] = act_lane[i], ag[1] = act_lane[(size_t)d.N + i], a = (typename E::Act)0;
        else
            ag[0] = ag[1] = 0, a = static_cast<const typename E::Act *>(actions)[i];
#pragma unroll
        for (int k = 0; k < 4; k++) g[k] = with_generator ? d.rng[(size_t)k * d.N + i] : 0;
    }
    MI_DEV void arrive(Lane<E> &L, Pcg64 &gen) {
#pragma unroll
        for (int k = 0; k < E::S; k++) hold_opaque(s[k]), L.s[k] = s[k];
        hold_opaque(meta), hold_opaque(ep_ret), hold_opaque(ep_len), hold_opaque(a), hold_opaque(ag[0]), hold_opaque(ag[1]);
#pragma unroll
        for (int k = 0; k < 4; k++) hold_opaque(g[k]);
        L.elapsed = meta & kElapsedMask, L.flags = meta >> kFlagShift;
        L.ep_ret = ep_ret, L.ep_len = ep_len;
        trig_invalidate(L.trig);
        gen.state = make_u128(g[0], g[1]), gen.inc = make_u128(g[2], g[3]);
    }
};
template <class E>
MI_DEV void store_lane(const DevEnv &d, int i, const Lane<E> &L) {
#pragma unroll
    for (int k = 0; k < E::S; k++) d.state[(size_t)k * d.N + i] = L.s[k];
    d.meta[i] = (L.elapsed & kElapsedMask) | (L.flags << kFlagShift);
    d.ep_ret[i] = L.ep_ret, d.ep_len[i] = L.ep_len;
}
MI_DEV Pcg64 load_rng(const DevEnv &d, int i) {
    Pcg64 r;
    r.state = make_u128(d.rng[i], d.rng[(size_t)d.N + i]);
    r.inc = make_u128(d.rng[(size_t)2 * d.N + i], d.rng[(size_t)3 * d.N + i]);
    return r;
}
MI_DEV void store_rng_state(const DevEnv &d, int i, const Pcg64 &r) {
    d.rng[i] = (uint64_t)(r.state >> 64), d.rng[(size_t)d.N + i] = (uint64_t)r.state;
}

// The next E::NDRAWS values of the lane's own stream, drawn AHEAD of the reset that will consume them.
// In a wavefront of 64 CartPoles some lane finishes an episode in ~95 % of the steps, so a reset path executed
// on demand costs every wavefront four 128-bit LCG steps (40 integer multiplies) on almost every
// step for the benefit of ~3 lanes.  A fused rollout instead refills all empty queues of a wavefront together
// every kRefillPeriod steps and a reset merely moves the queued values into the state.  The stream order is
// unchanged (the env's generator is consumed by resets only), and unconsumed draws are handed back at the end of
// the launch (Pcg64::unstep), so the generator state in HBM is always exactly the reference's.
// The generator itself is held in registers for the whole launch: a loop without global loads also keeps the
// compiler from draining the store queue (s_waitcnt vmcnt(0)) on every iteration -- loads and stores share one
// in-order counter on gfx950.
template <class E>
struct ResetQueue {
    double u[E::NDRAWS];
    double rs[E::S];  // the reset state those draws give under the default bounds (autoreset has no options): computed once per refill
    // (an integer, not a bool: a bool that lives across the role branches of rollout_duo_kernel is a lane mask in scalar registers, which the
    //  compiler merges with three scalar instructions at every join of every phase -- ~9 instructions per phase and flag, on every wavefront)
    uint32_t have;
    Pcg64 rng;
    MI_DEV void refill() {
#pragma unroll
        for (int k = 0; k < E::NDRAWS; k++) u[k] = rng.next_double();
        double b0, b1;
        uint32_t f = 0;
        E::default_bounds(b0, b1);
        E::reset_u(u, rs, f, b0, b1);
        have = 1u;
    }
    // what reset_u does to the flag word does not depend on the draws: it sets or clears kStateF32 (envs_classic.h)
    static MI_DEV uint32_t reset_flags(uint32_t flags) {
        const double zero[E::NDRAWS] = {};
        double tmp[E::S], b0, b1;
        E::default_bounds(b0, b1);
        E::reset_u(zero, tmp, flags, b0, b1);
        return flags;
    }
};
constexpr int kRefillPeriod = 8;

template <class E>
MI_DEV void draw_reset_values(const DevEnv &d, int i, double u[E::NDRAWS], const Pcg64 *preloaded = nullptr) {
    Pcg64 rng = preloaded ? *preloaded : load_rng(d, i);
#pragma unroll
    for (int k = 0; k < E::NDRAWS; k++) u[k] = rng.next_double();
    store_rng_state(d, i, rng);
}

// Reset of one lane from its own stream, with the reference's default bounds (autoreset: reset() has no options).
template <class E>
MI_DEV void lane_autoreset(const DevEnv &d, int i, Lane<E> &L, ResetQueue<E> *q, const Pcg64 *preloaded = nullptr) {
    double u[E::NDRAWS];
    if (q) {
        if (!q->have) q->refill();  // rare: two episode ends within one refill period
#pragma unroll
        for (int k = 0; k < E::NDRAWS; k++) u[k] = q->u[k];
        q->have = 0u;
    } else {
        draw_reset_values<E>(d, i, u, preloaded);
    }
    double b0, b1;
    E::default_bounds(b0, b1);
    E::reset_u(u, L.s, L.flags, b0, b1);
    L.elapsed = 0;  // TimeLimit.reset (wrappers/common.py:149)
    L.ep_ret = 0.0, L.ep_len = 0;
}

template <class E>
struct StepOut {
    float obs[E::OBS];
    float final_obs[E::OBS];
    double reward, ep_ret;
    int32_t ep_len;
    bool terminated, truncated, has_final;
};

// One lockstep step of one sub-environment: sync_vector_env.py:277-329 + TimeLimit + RecordEpisodeStatistics.
template <class E, int MODE>
MI_DEV void lane_step(const DevEnv &d, int i, Lane<E> &L, typename E::Act a, StepOut<E> &o, LaneStats &st,
                      ResetQueue<E> *q = nullptr, const Pcg64 *preloaded = nullptr) {
    bool te = false, tr = false;
    double rew = 0.0;
    o.has_final = false;
    if (MODE == MI_AUTORESET_NEXT_STEP && (L.flags & kNeedsReset)) {
        // :279-284 the step after a finished episode resets, ignores the action, returns reward 0 / not done
        lane_autoreset<E>(d, i, L, q, preloaded);
        st.reset_steps++;
    } else if (MODE == MI_AUTORESET_DISABLED && (L.flags & kNeedsReset)) {
        // :295 `assert not self._autoreset_envs[i]`: report through the sticky error word, leave the lane untouched
        *d.error = kErrDisabledStepped;
        E::obs(L.s, L.flags, o.obs, L.trig);
        o.reward = 0.0, o.terminated = false, o.truncated = false, o.ep_ret = 0.0, o.ep_len = 0;
        return;
    } else {
        if (!E::valid(a)) {
            // cartpole.py:165-167 asserts before it touches the state: report through the sticky error word, leave the lane untouched (host
            // callers never get here -- their actions are validated before the launch)
            *d.error = kErrInvalidAction;
            E::obs(L.s, L.flags, o.obs, L.trig);
            o.reward = 0.0, o.terminated = false, o.truncated = false, o.ep_ret = 0.0, o.ep_len = 0;
            return;
        }
        E::step(L.s, L.flags, a, d.P, rew, te, L.trig);
        L.elapsed += 1;  // TimeLimit.step (wrappers/common.py:129-133)
        tr = d.max_steps > 0 && (int)L.elapsed >= d.max_steps;
        L.ep_ret += rew, L.ep_len += 1;
        st.env_steps++;
    }
    const bool done = te || tr;
    o.ep_ret = done ? L.ep_ret : 0.0;
    o.ep_len = done ? L.ep_len : 0;
    if (done) {
        st.episodes++;
        st.return_sum += L.ep_ret;
        st.length_sum += (uint64_t)L.ep_len;
    }
    if (MODE == MI_AUTORESET_SAME_STEP && done) {
        // :302-319 final_obs, then reset within the same step
        E::obs(L.s, L.flags, o.final_obs, L.trig);
        o.has_final = true;
        lane_autoreset<E>(d, i, L, q, preloaded);
    }
    E::obs(L.s, L.flags, o.obs, L.trig);
    o.reward = rew, o.terminated = te, o.truncated = tr;
    if (done && MODE != MI_AUTORESET_SAME_STEP)
        L.flags |= kNeedsReset;  // _autoreset_envs (:329)
    else
        L.flags &= ~kNeedsReset;
}

// NEXT_STEP step of one lane inside a fused rollout, written without divergent control flow: a wavefront that runs
// alone on its SIMD pays an issue slot for every s_and_saveexec / s_cbranch and a refetch for every taken branch,
// and with 64 CartPoles per wavefront some lane is in its autoreset step on ~95 % of the steps anyway.  So every
// lane integrates (a finished episode's state is finite; the result is discarded) and the autoreset lanes SELECT the
// queued reset values instead.  Same results as lane_step<E, NEXT_STEP>.
template <class E, bool CHECK_ACTION>
MI_DEV void lane_step_fused(const DevEnv &d, Lane<E> &L, typename E::Act a, StepOut<E> &o, LaneStats &st, ResetQueue<E> &q) {
    const bool resetting = (L.flags & kNeedsReset) != 0;
    if (__builtin_expect(resetting && !q.have, 0)) q.refill();  // rare: two episode ends within one refill period
    // cartpole.py:165-167 asserts before it touches the state: an action outside the space is reported through the sticky error word and
    // the lane is left untouched (lane_step's rule) -- it integrates a stand-in action like a resetting lane does and SELECTS its old state.
    bool invalid = false;
    double s0[E::S];
    if (CHECK_ACTION) {
#pragma unroll
        for (int k = 0; k < E::S; k++) s0[k] = L.s[k];
        invalid = !resetting && !E::valid(a);
        if (invalid) {
            *d.error = kErrInvalidAction;
            a = (typename E::Act)0;
        }
    }
    // the reset candidate (sync_vector_env.py:279-284): the state was formed from the queued draws when they were drawn
    const double (&rs)[E::S] = q.rs;
    const uint32_t rflags = ResetQueue<E>::reset_flags(L.flags & ~kNeedsReset);
    // the step candidate
    double rew;
    bool te;
    uint32_t sflags = L.flags;
    if constexpr (E::SPLIT_TERMINAL) {
        // Acrobot: the terminal test needs cos(theta1) of the NEW state, which the observation evaluates anyway.  Integrate, select the reset
        // lanes' state, take the observation, then test the terminal condition on it with the observation's trig values: for a stepping lane
        // the selected state IS its new state (same argument, same function, same bits); a resetting lane's flag is discarded below.
        E::integrate(L.s, sflags, a, L.trig);
        if (CHECK_ACTION && invalid) {
#pragma unroll
            for (int k = 0; k < E::S; k++) L.s[k] = s0[k];
            sflags = L.flags;
        }
#pragma unroll
        for (int k = 0; k < E::S; k++) L.s[k] = resetting ? rs[k] : L.s[k];
        E::obs(L.s, sflags, o.obs, L.trig);
        E::terminal_after_obs(L.s, L.trig, rew, te);
        if (CHECK_ACTION && invalid) rew = 0.0, te = false;
    } else {
        E::step(L.s, sflags, a, d.P, rew, te, L.trig);
        if (CHECK_ACTION && invalid) {  // rare, divergent: undo
#pragma unroll
            for (int k = 0; k < E::S; k++) L.s[k] = s0[k];
            sflags = L.flags, rew = 0.0, te = false;
            trig_invalidate(L.trig);
        }
    }
    const uint32_t moved = (CHECK_ACTION && invalid) ? 0u : 1u;
    const uint32_t elapsed = L.elapsed + moved;  // TimeLimit.step (wrappers/common.py:129-133)
    const bool tr = moved && d.max_steps > 0 && (int)elapsed >= d.max_steps;
    const double ep_ret = L.ep_ret + rew;
    const int32_t ep_len = L.ep_len + (int32_t)moved;
    const bool done = !resetting && (te || tr);
    if constexpr (!E::SPLIT_TERMINAL) {
#pragma unroll
        for (int k = 0; k < E::S; k++) L.s[k] = resetting ? rs[k] : L.s[k];
    }
    L.flags = resetting ? rflags : (done ? (sflags | kNeedsReset) : sflags);
    L.elapsed = resetting ? 0u : elapsed;
    L.ep_ret = resetting ? 0.0 : ep_ret;
    L.ep_len = resetting ? 0 : ep_len;
    q.have = resetting ? 0u : q.have;
    st.reset_steps += resetting ? 1u : 0u;
    st.env_steps += resetting ? 0u : moved;
    st.episodes += done ? 1u : 0u;
    st.return_sum += done ? ep_ret : 0.0;
    st.length_sum += done ? (uint64_t)ep_len : 0ull;
    if constexpr (!E::SPLIT_TERMINAL) E::obs(L.s, L.flags, o.obs, L.trig);
    o.reward = resetting ? 0.0 : rew;
    o.terminated = !resetting && te, o.truncated = !resetting && tr;
    o.ep_ret = done ? ep_ret : 0.0, o.ep_len = done ? ep_len : 0;
    o.has_final = false;
}

template <int W>
MI_DEV void store_row(float *dst, const float *src) {
    if (W == 4) {
        *reinterpret_cast<float4 *>(dst) = make_float4(src[0], src[1], src[2], src[3]);
    } else if (W == 2) {
        *reinterpret_cast<float2 *>(dst) = make_float2(src[0], src[1]);
    } else if (W == 6) {
        float2 *p = reinterpret_cast<float2 *>(dst);
        p[0] = make_float2(src[0], src[1]), p[1] = make_float2(src[2], src[3]), p[2] = make_float2(src[4], src[5]);
    } else {
#pragma unroll
        for (int k = 0; k < W; k++) dst[k] = src[k];
    }
}

template <int W>
MI_DEV void load_row(const float *src, float *dst) {
#pragma unroll
    for (int k = 0; k < W; k++) dst[k] = src[k];
}

MI_DEV double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
MI_DEV uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor((int)v, off, 64);
    return v;
}
MI_DEV uint64_t wave_sum(uint64_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += (uint64_t)__shfl_xor((long long)v, off, 64);
    return v;
}

// Per-workgroup totals: wavefront butterflies, 4 partials through LDS, one plain read-modify-write of the
// workgroup's own slot (no atomics: 1024 same-address atomics would cost more than the whole step).
// The workgroup's previous totals as step_kernel loads them AHEAD (with the lane's state): the read-modify-write at the end of the kernel is
// then a plain store, not one more dependent trip to memory in a kernel whose whole run time is a handful of such trips.
struct BlockTotals {
    uint64_t count;  // threads 0..3: blk_count[block][thread]
    double ret;      // thread 64: blk_ret[block]
    bool loaded;
};
MI_DEV BlockTotals block_totals_load(const DevEnv &d) {
    BlockTotals t = {0ull, 0.0, true};
    if (threadIdx.x < 4) t.count = d.blk_count[(size_t)blockIdx.x * 4 + threadIdx.x];
    if (threadIdx.x == 64) {
        uint32_t zero = 0;  // (an offset the compiler cannot see through: a VECTOR load that stays in flight -- the uniform address would make it an s_load, which the wavefront waits for on the spot)
        hold_opaque(zero);
        t.ret = d.blk_ret[(size_t)blockIdx.x + zero];
    }
    return t;
}
MI_DEV void block_accumulate(const DevEnv &d, const LaneStats &st, const BlockTotals &before = BlockTotals{0ull, 0.0, false}) {
    __shared__ uint64_t sh_c[4][kBlock / 64];
    __shared__ double sh_r[kBlock / 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t c0 = wave_sum(st.env_steps), c1 = wave_sum(st.reset_steps), c2 = wave_sum(st.episodes);
    // c2 is wavefront-uniform: the two wide reductions are skipped by wavefronts in which no episode finished
    const uint64_t c3 = c2 ? wave_sum(st.length_sum) : 0;
    const double r = c2 ? wave_sum(st.return_sum) : 0.0;
    if (lane == 0) sh_c[0][wave] = c0, sh_c[1][wave] = c1, sh_c[2][wave] = c2, sh_c[3][wave] = c3, sh_r[wave] = r;
    __syncthreads();
    if (threadIdx.x < 4) {
        uint64_t t = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; w++) t += sh_c[threadIdx.x][w];
        if (t) {
            uint64_t *slot = &d.blk_count[(size_t)blockIdx.x * 4 + threadIdx.x];
            *slot = (before.loaded ? before.count : *slot) + t;
        }
    } else if (threadIdx.x == 64) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; w++) t += sh_r[w];
        if (t != 0.0) d.blk_ret[blockIdx.x] = (before.loaded ? before.ret : d.blk_ret[blockIdx.x]) + t;
    }
}

// ---------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------------------------
// Step epilogue: gymnasium.wrappers.vector.{NormalizeObservation, NormalizeReward, ClipReward} as the output stage of step_kernel
// (mi_set_step_epilogue; the same arithmetic as the stand-alone passes of wrappers.hip, which cite the reference line by line).
//   phase 1, inside step_kernel: ClipReward placed before the normalisation, the discounted-return update, and this workgroup's column sums
//     of (obs - running mean), (return - running mean) and their squares: wavefront butterflies, four partials through LDS, one row of
//     doubles per workgroup in global memory.  No cross-workgroup traffic inside the kernel: a "last workgroup folds the rows" scheme
//     was measured first and cost 24 us per step -- a device-scope release on gfx950 writes back the whole XCD-local L2, which the step
//     has just filled with dirty state and observation lines.
//   phase 2, epilogue_finish: both normalisations need the UPDATED statistics of the whole batch (stateful_observation.py:146-152 updates
//     before it normalises), i.e. a grid-wide dependency, so they are one small second launch: every workgroup folds the rows (the same
//     sums in the same order -> the same bits everywhere), applies RunningMeanStd.update to a private copy and rewrites its obs / reward in
//     place, before the single device-to-host copy of the NumPy path; workgroup 0 also stores the new statistics -- into the handle's
//     SECOND buffer set, because the other workgroups are still reading the old one; the host swaps the two pointer sets after the launch.
//     Beyond kEpiFoldMax step workgroups the fold is a one-workgroup launch of its own (epilogue_combine) and phase 2 reads its result.
// Together: 2 launches per wrapped step (3 beyond 262144 sub-environments) instead of 1 + 4 + 5 + 1 with the stand-alone passes.
// ---------------------------------------------------------------------------------------------------------
constexpr int kEpiObsMax = 6, kEpiCols = 2 * kEpiObsMax + 3, kEpiFoldMax = 1024;
}  // namespace
namespace mi_internal {
struct EpiDev {
    int obs_on, obs_update, ret_on, ret_update, same_step, clip_pre, clip_post, fold_in_finish;  // clip_*: bit 0 = has min, bit 1 = has max
    int step_blocks;
    float gamma;
    double obs_eps, ret_eps, pre_lo, pre_hi, post_lo, post_hi;
    double *obs_mean, *obs_var, *obs_count, *ret_mean, *ret_var, *ret_count;              // the statistics before this step (read only)
    double *obs_mean2, *obs_var2, *obs_count2, *ret_mean2, *ret_var2, *ret_count2;        // where the updated ones go
    float *acc;
    uint8_t *prev_done;
    double *partial;   // [step_blocks][kEpiCols]
};
}  // namespace mi_internal
namespace {
MI_DEV double clip_to(double v, int flags, double lo, double hi) {  // np.clip(reward, min_reward, max_reward)
    if (flags & 1) v = v < lo ? lo : v;
    if (flags & 2) v = v > hi ? hi : v;
    return v;
}
__host__ __device__ inline bool epi_reduces(const EpiDev &e) { return (e.obs_on && e.obs_update) || (e.ret_on && e.ret_update); }

template <int NC>
MI_DEV void block_sum_columns(double (&v)[NC], double (*sh)[kEpiCols]) {  // result on threads < NC of the workgroup: v[0] = their column's total
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < NC; k++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off, 64);
    }
    __syncthreads();  // sh may still be read from an earlier call
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NC; k++) sh[wave][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < NC) {
        double t = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; w++) t += sh[w][threadIdx.x];
        v[0] = t;
    }
}

// every thread of the workgroup calls this (valid = it owns a sub-environment); reward is rewritten with what the finish pass / the caller reads
template <class E>
MI_DEV void epilogue_phase1(const EpiDev &e, int i, bool valid, const float *obs, double &reward, bool terminated) {
    constexpr int OBS = E::OBS, NC = 2 * OBS + 3;
    static_assert(OBS <= kEpiObsMax, "epilogue rows are sized for the classic-control observations");
    double v[NC];
#pragma unroll
    for (int k = 0; k < NC; k++) v[k] = 0;
    if (valid) {
        double r = clip_to(reward, e.clip_pre, e.pre_lo, e.pre_hi);
        if (e.ret_on) {  // stateful_reward.py:150-176 (wrappers.hip accumulate_return / update_stats)
            const bool pd = e.prev_done[i] != 0, active = e.same_step || !pd;
            float a = e.acc[i];
            if (active) {
                const float g = a * e.gamma;
                a = (float)((double)g * (terminated ? 0.0 : 1.0) + r);
                e.acc[i] = a;
            }
            if (e.ret_update && active) {
                const double d = (double)a - e.ret_mean[0];
                v[2 * OBS] = d, v[2 * OBS + 1] = d * d, v[2 * OBS + 2] = 1.0;
            }
        } else {
            r = clip_to(r, e.clip_post, e.post_lo, e.post_hi);  // no normalisation in between: both clips happen here
        }
        reward = r;
        if (e.obs_on && e.obs_update) {
#pragma unroll
            for (int c = 0; c < OBS; c++) {
                const double d = (double)obs[c] - e.obs_mean[c];
                v[c] = d, v[OBS + c] = d * d;
            }
        }
    }
    if (!epi_reduces(e)) return;  // grid-uniform
    __shared__ double sh[kBlock / 64][kEpiCols];
    block_sum_columns<NC>(v, sh);
    if (threadIdx.x < NC) e.partial[(size_t)blockIdx.x * kEpiCols + threadIdx.x] = v[0];
}

// The statistics after this step's update, for the whole workgroup: st[0..OBS) mean, st[OBS..2 OBS) var, st[2 OBS] return var, st[2 OBS + 1] return
// mean, st[2 OBS + 2] obs count, st[2 OBS + 3] return count.  fold = sum the step workgroups' rows first (else: the statistics as they are).
template <int OBS>
MI_DEV void epilogue_statistics(const EpiDev &e, int N, bool fold, double (*sh)[kEpiCols], double *st) {
    constexpr int NC = 2 * OBS + 3;
    if (threadIdx.x < OBS && e.obs_on) st[threadIdx.x] = e.obs_mean[threadIdx.x], st[OBS + threadIdx.x] = e.obs_var[threadIdx.x];
    if (threadIdx.x == 64) {
        st[2 * OBS] = e.ret_on ? e.ret_var[0] : 1.0, st[2 * OBS + 1] = e.ret_on ? e.ret_mean[0] : 0.0;
        st[2 * OBS + 2] = e.obs_on ? *e.obs_count : 0.0, st[2 * OBS + 3] = e.ret_on ? *e.ret_count : 0.0;
    }
    if (fold) {  // workgroup-uniform
        double v[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) v[k] = 0;
        for (int b = threadIdx.x; b < e.step_blocks; b += kBlock) {
#pragma unroll
            for (int k = 0; k < NC; k++) v[k] += e.partial[(size_t)b * kEpiCols + k];
        }
        block_sum_columns<NC>(v, sh);
        __syncthreads();
        if (threadIdx.x < NC) sh[0][threadIdx.x] = v[0];
        __syncthreads();
        if (e.obs_on && e.obs_update && (int)threadIdx.x < OBS) {  // float32 observations, float32 statistics (RunningMeanStd(dtype=float32))
            const int c = threadIdx.x;
            mi_wrap::update_column<float, float>(st[c], st[OBS + c], *e.obs_count, sh[0][c], sh[0][OBS + c], (double)N);
        }
        if (threadIdx.x == 64) {
            if (e.obs_on && e.obs_update) st[2 * OBS + 2] += (double)N;
            const double rows = sh[0][2 * OBS + 2];
            if (e.ret_on && e.ret_update && rows > 0) {  // float32 returns, float64 statistics; `if np.any(active)`
                mi_wrap::update_column<double, float>(st[2 * OBS + 1], st[2 * OBS], *e.ret_count, sh[0][2 * OBS], sh[0][2 * OBS + 1], rows);
                st[2 * OBS + 3] += rows;
            }
        }
    }
    __syncthreads();
}
template <int OBS>
MI_DEV void epilogue_store_statistics(const EpiDev &e, const double *st) {  // one workgroup: the updated statistics into the second buffer set
    if (e.obs_on && (int)threadIdx.x < OBS) e.obs_mean2[threadIdx.x] = st[threadIdx.x], e.obs_var2[threadIdx.x] = st[OBS + threadIdx.x];
    if (threadIdx.x == 64) {
        if (e.obs_on) *e.obs_count2 = st[2 * OBS + 2];
        if (e.ret_on) e.ret_var2[0] = st[2 * OBS], e.ret_mean2[0] = st[2 * OBS + 1], *e.ret_count2 = st[2 * OBS + 3];
    }
}
template <int OBS>
__global__ __launch_bounds__(kBlock) void epilogue_combine(EpiDev e, int N) {
    __shared__ double sh[kBlock / 64][kEpiCols];
    __shared__ double st[2 * OBS + 4];
    epilogue_statistics<OBS>(e, N, true, sh, st);
    epilogue_store_statistics<OBS>(e, st);
}

template <int OBS>
__global__ __launch_bounds__(kBlock) void epilogue_finish(EpiDev e, int N, float *obs, double *reward, const uint8_t *terminated, const uint8_t *truncated) {
    __shared__ double sh[kBlock / 64][kEpiCols];
    __shared__ double st[2 * OBS + 4];
    if (e.fold_in_finish) {
        epilogue_statistics<OBS>(e, N, epi_reduces(e), sh, st);
        if (blockIdx.x == 0 && epi_reduces(e)) epilogue_store_statistics<OBS>(e, st);
    } else {
        // epilogue_combine ran iff some statistic is being updated: then the second buffer set holds the statistics after this step.  With
        // frozen statistics (update_running_mean = False on both wrappers) nothing was folded or swapped: the primary set is the current one.
        const bool upd = epi_reduces(e);
        const double *om = upd ? e.obs_mean2 : e.obs_mean, *ov = upd ? e.obs_var2 : e.obs_var, *rv = upd ? e.ret_var2 : e.ret_var;
        if (threadIdx.x < OBS && e.obs_on) st[threadIdx.x] = om[threadIdx.x], st[OBS + threadIdx.x] = ov[threadIdx.x];
        if (threadIdx.x == 64 && e.ret_on) st[2 * OBS] = rv[0];
        __syncthreads();
    }
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    if (e.obs_on) {  // (obs - mean) / np.sqrt(var + epsilon), float32 throughout (wrappers.hip normalize_obs<float, float, float>)
        float o[OBS];
        load_row<OBS>(obs + (size_t)i * OBS, o);
#pragma unroll
        for (int c = 0; c < OBS; c++) {
            const float num = o[c] - (float)st[c];
            const float den = (float)sqrt((double)(float)((float)st[OBS + c] + (float)e.obs_eps));
            o[c] = num / den;
        }
        store_row<OBS>(obs + (size_t)i * OBS, o);
    }
    if (e.ret_on) {  // wrappers.hip finish_reward
        const uint8_t done = (terminated[i] || truncated[i]) ? 1 : 0;
        e.prev_done[i] = done;
        if (e.same_step && done) e.acc[i] = 0.0f;
        const double r = reward[i] / sqrt(st[2 * OBS] + e.ret_eps);
        reward[i] = clip_to(r, e.clip_post, e.post_lo, e.post_hi);
    }
}

}  // namespace
namespace mi_internal {
struct StepPtrs {
    const void *actions;
    float *obs;
    double *reward;
    uint8_t *terminated, *truncated;
    float *final_obs;
    double *ep_ret;
    int32_t *ep_len;
    // the on-device policy (mi_step with actions == NULL): per-lane states of the action stream [2][N] (lane i: the state whose output is draw
    // pos + i * act_dim), where the drawn actions go (or nullptr), and the jump to the lane's slot in the next batch (N * act_dim draws on)
    uint64_t *act_lane;
    void *actions_out;
    PcgJump act_jump;
};
}  // namespace mi_internal
namespace {

// XSL-RR output of a PCG64 state (the state AFTER its step: what the action-stream lanes hold)
MI_DEV uint64_t pcg_output(u128 state) {
    const uint64_t hi = (uint64_t)(state >> 64), lo = (uint64_t)state;
    const uint64_t x = hi ^ lo;
    const unsigned rot = (unsigned)(hi >> 58);
    return (x >> rot) | (x << ((0u - rot) & 63u));
}

// action_space.sample() of one lane from the state of the action stream whose output is the lane's draw
template <class E>
MI_DEV typename E::Act action_of_state(u128 astate) {
    if constexpr (E::SAMPLE_FROM_BITS) {
        return E::sample_state((uint64_t)(astate >> 64), (uint64_t)astate);
    } else {
        return E::sample((double)(pcg_output(astate) >> 11) * (1.0 / 9007199254740992.0));
    }
}

// SAMPLE: mi_step with actions == NULL -- `step(action_space.sample())` in one launch: the lane draws its own action from its state of the action
// stream (spaces/multi_discrete.py:176-178, spaces/box.py:463-465: draw number pos + i of the batched space's generator), which rides along with the
// lane's other loads, and leaves the state of its draw in the NEXT batch behind.  Nothing on the host changes from step to step: capturable.
template <class E, int MODE, bool EPI = false, bool SAMPLE = false>
__global__ __launch_bounds__(kBlock) void step_kernel(DevEnv d, StepPtrs io, EpiDev epi) {
    // One step is ~200 instructions per lane between trips to memory that take a microsecond each: everything the step may read -- the lane's
    // state and action, its generator (consumed only by a reset, which some lane of a 64-CartPole wavefront needs on ~95 % of the steps) and the
    // workgroup's running totals -- is requested BEFORE the math tables are staged into LDS, so that all of it is ONE trip, not four in a row.
    const int i = blockIdx.x * kBlock + threadIdx.x;
    LaneStats st = {0u, 0u, 0u, 0ull, 0.0};
    StepOut<E> o;
    LaneRequest<E> rq;
    if (i < d.N) rq.template request<SAMPLE>(d, io.actions, io.act_lane, i, MODE != MI_AUTORESET_DISABLED);
    BlockTotals before = block_totals_load(d);
    tables_init<E>();
    hold_opaque(before.count), hold_opaque(before.ret);
    if (i < d.N) {
        Lane<E> L;
        Pcg64 gen;
        rq.arrive(L, gen);
        typename E::Act a = rq.a;
        if constexpr (SAMPLE) {
            const u128 astate = make_u128(rq.ag[0], rq.ag[1]);
            a = action_of_state<E>(astate);
            const u128 next = io.act_jump.mult * astate + io.act_jump.plus;
            io.act_lane[i] = (uint64_t)(next >> 64), io.act_lane[(size_t)d.N + i] = (uint64_t)next;
            if (io.actions_out) static_cast<typename E::Act *>(io.actions_out)[i] = a;
        }
        lane_step<E, MODE>(d, i, L, a, o, st, nullptr, MODE != MI_AUTORESET_DISABLED ? &gen : nullptr);
        store_lane<E>(d, i, L);
        if (io.obs) store_row<E::OBS>(io.obs + (size_t)i * E::OBS, o.obs);
        if (!EPI && io.reward) io.reward[i] = o.reward;
        if (io.terminated) io.terminated[i] = o.terminated;
        if (io.truncated) io.truncated[i] = o.truncated;
        if (MODE == MI_AUTORESET_SAME_STEP && io.final_obs && o.has_final)
            store_row<E::OBS>(io.final_obs + (size_t)i * E::OBS, o.final_obs);
        if (io.ep_ret) io.ep_ret[i] = o.ep_ret;
        if (io.ep_len) io.ep_len[i] = o.ep_len;
    }
    if (EPI) {  // the wrappers' statistics see what the step wrote; episode statistics (above) keep the raw reward, like the reference's order
        double r = i < d.N ? o.reward : 0.0;
        epilogue_phase1<E>(epi, i, i < d.N, o.obs, r, i < d.N && o.terminated);
        if (i < d.N && io.reward) io.reward[i] = r;
    }
    block_accumulate(d, st, before);
}

template <class E>
__global__ __launch_bounds__(kBlock) void reset_kernel(DevEnv d, const uint8_t *mask, int has_bounds, double b0, double b1,
                                                       float *obs) {
    tables_init<E>();
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= d.N) return;
    if (mask && !mask[i]) return;
    Lane<E> L;
    load_lane<E>(d, i, L);
    double u[E::NDRAWS];
    draw_reset_values<E>(d, i, u);
    if (!has_bounds) E::default_bounds(b0, b1);
    L.flags &= ~kNeedsReset;
    E::reset_u(u, L.s, L.flags, b0, b1);
    L.elapsed = 0, L.ep_ret = 0.0, L.ep_len = 0;
    store_lane<E>(d, i, L);
    if (obs) {
        float o[E::OBS];
        E::obs(L.s, L.flags, o, L.trig);
        store_row<E::OBS>(obs + (size_t)i * E::OBS, o);
    }
}

}  // namespace
namespace mi_internal {
// Action stream of the batched action space: ONE PCG64 generator drawn in sub-environment index order, so lane i
// consumes draw number t*N + i of the stream at step t.  jump[j] advances by 2^j draws; jump_n advances by N.
struct ActionStream {
    uint64_t state_hi, state_lo, inc_hi, inc_lo;
    const PcgJump *pow2;  // [64] device
    PcgJump jump_n;
};

struct RolloutPtrs {
    const void *actions_in;
    void *actions_out;
    void *obs;
    double *reward;
    uint8_t *terminated, *truncated;
};
// MI_CFG_SHARED_RNG (include/mi355env.h): CartPoleVectorEnv's ONE generator on the device.  It is kept as a fixed BASE state (what mi_seed gave) plus
// the number of draws taken since: draw number n of the stream is the output after skipping n steps ahead of the base (Brown's O(log n) jump through
// the pow2 table), so every lane finds its own draws without a serial pass and nothing but two counters ever changes.
struct SharedRng {
    uint64_t *words;       // [8] device: base {state_hi, state_lo, inc_hi, inc_lo}, [4] consumed, [5] pos_base = first draw of the call in flight, [6] k = sub-envs it re-draws,
                           // [7] != 0: a batch with an action outside the space was refused (shared_validate_kernel) -- every later step is a no-op until the host has raised it
    const PcgJump *pow2;   // [64] device: jump by 2^j steps of the base generator's increment
    uint32_t *blk_done;    // [grid] sub-environments of each step workgroup that finished an episode in the previous step
    uint32_t *blk_prefix;  // [grid] exclusive scan of blk_done (shared_scan_kernel)
    double low, high;      // self.low / self.high (cartpole.py:489-491): the bounds of the last reset() serve the autoresets
};
}  // namespace mi_internal
// The classic-control kernels behind plain launch functions (defined in the MI_CLASSIC_TU unit, called from the other): what mi_step / mi_reset /
// mi_rollout do for the kinds CARTPOLE ... MOUNTAIN_CAR_CONTINUOUS once the arguments are validated and the buffers chosen.
namespace mi_classic {
int step(mi_vecenv *v, const mi_internal::StepPtrs &p, int act_kind);                                           // launch_step<E> of the env's kind
int reset(mi_vecenv *v, const uint8_t *device_mask, int has_bounds, double b0, double b1, float *device_obs);  // reset_kernel<E>
int rollout(mi_vecenv *v, const mi_internal::RolloutPtrs &p, const mi_internal::ActionStream &as, int T, bool sample, int actions_in_kind);
// MI_CFG_SHARED_RNG (CartPole only): reset = [bookkeeping, reset kernel]; step = [scan of the finished sub-environments, step kernel];
// rollout = T x [policy sample,] step; recount = blk_done from the flag words (after mi_set_state)
int shared_reset(mi_vecenv *v, float *device_obs);
int shared_step(mi_vecenv *v, const mi_internal::StepPtrs &p);
int shared_rollout(mi_vecenv *v, const mi_internal::RolloutPtrs &p, const mi_internal::ActionStream &as, int T, bool sample);
int shared_recount(mi_vecenv *v);
}  // namespace mi_classic
namespace {

// FULL: every trajectory output is materialised -- the per-store null checks (five taken branches per step for a
// wavefront that runs alone on its SIMD) disappear from the loop.
template <class E, int MODE, bool SAMPLE, bool FULL>
__global__ __launch_bounds__(kBlock) void rollout_kernel(DevEnv d, RolloutPtrs io, ActionStream as, int T) {
    tables_init<E>();
    const int i = blockIdx.x * kBlock + threadIdx.x;
    LaneStats st = {0u, 0u, 0u, 0ull, 0.0};
    if (i < d.N) {
        Lane<E> L;
        load_lane<E>(d, i, L);
        u128 astate = 0;
        const u128 ainc = make_u128(as.inc_hi, as.inc_lo);
        if (SAMPLE) {
            // skip ahead by (i + 1) draws: one affine map per set bit of (i + 1)
            astate = make_u128(as.state_hi, as.state_lo);
            uint32_t delta = (uint32_t)i + 1u;
            for (int j = 0; delta; j++, delta >>= 1)
                if (delta & 1u) astate = as.pow2[j].mult * astate + as.pow2[j].plus;
        }
        (void)ainc;
        const size_t N = (size_t)d.N;
        ResetQueue<E> q;
        q.have = 0u;
        q.rng = load_rng(d, i);
        // (per-kind unroll factor: 2 for Acrobot -- its long loop body schedules better as two steps, +4.6 %; 1 = none for the others, where 2 and 4
        //  measured +-0.1 %; Acrobot x4: -6 %.  profiles/r04_maxilp_classic.txt)
#pragma unroll E::ROLLOUT_UNROLL
        for (int t = 0; t < T; t++) {
            if ((t & (kRefillPeriod - 1)) == 0 && !q.have) q.refill();
            typename E::Act a;
            if (SAMPLE) {
                a = action_of_state<E>(astate);
                astate = as.jump_n.mult * astate + as.jump_n.plus;
                if (FULL || io.actions_out) static_cast<typename E::Act *>(io.actions_out)[t * N + i] = a;
            } else {
                a = static_cast<const typename E::Act *>(io.actions_in)[t * N + i];
            }
            StepOut<E> o;
            if (MODE == MI_AUTORESET_NEXT_STEP)
                lane_step_fused<E, !SAMPLE>(d, L, a, o, st, q);
            else
                lane_step<E, MODE>(d, i, L, a, o, st, &q);
            if (FULL || io.obs) store_row<E::OBS>(static_cast<float *>(io.obs) + (t * N + i) * E::OBS, o.obs);
            if (FULL || io.reward) io.reward[t * N + i] = o.reward;
            if (FULL || io.terminated) io.terminated[t * N + i] = o.terminated;
            if (FULL || io.truncated) io.truncated[t * N + i] = o.truncated;
        }
        store_lane<E>(d, i, L);
        if (q.have) {  // hand the unconsumed draws back to the env's generator
#pragma unroll
            for (int k = 0; k < E::NDRAWS; k++) q.rng.unstep();
        }
        store_rng_state(d, i, q.rng);
    }
    block_accumulate(d, st);
}

// ---------------------------------------------------------------------------------------------------------
// The collector's rollout (NEXT_STEP, on-device policy, every output materialised) as TWO wavefronts per 64 sub-environments.
//
// At the benchmark's 65 536 sub-environments rollout_kernel puts ONE wavefront on every SIMD, and a wavefront that runs alone has nothing but
// its own independent instructions to cover a dependent instruction's latency: 0.63 of the issue slots are used (bench.py issue_bound), while
// the same kernel at 4 wavefronts per SIMD (262 144 lanes) reaches 0.6 of the HBM roofline instead of 0.43.  The sub-environments are the only
// parallelism the problem has, but a step is not one indivisible chain: the policy's action stream does not depend on the environment at all,
// and everything that happens to a step's results (episode return / length, the metric totals, five coalesced stores) is off the path that
// leads to the next state.  So a workgroup is 256 sub-environments and 512 lanes: wavefronts 0..3 ("env") integrate, decide termination /
// truncation / autoreset and produce the observation; wavefronts 4..7 ("aux"; wavefront w + 4 serves wavefront w and shares its SIMD: measured,
// the pairing (2j, 2j + 1) puts two env wavefronts on one SIMD and is 25 % slower) draw the actions ahead, keep the episode statistics and write
// the whole trajectory.  They meet through LDS rings, a chunk of DuoTraits::CHUNK steps at a time:
//   phase p:  aux draws the actions of chunk p -> A[p & 1] (and stores them);  env steps chunk p - 1 from A[(p - 1) & 1] -> O[(p - 1) & 1];
//             aux consumes O[p & 1] (chunk p - 2): bookkeeping and stores;  one workgroup barrier.
// The arithmetic of every value is the one of rollout_kernel / lane_step_fused (same operations on the same operands: bit-identical
// trajectories, tests/test_gpu_rollout_roles.py, tests/test_gpu_parity.py); only WHICH lane issues an instruction changed.  MI355ENV_ROLLOUT_DUO=0
// restores the one-role kernel.  The measurements behind every choice here: profiles/r04_two_role_rollout.txt, profiles/r04_ubench_valu_waves.txt.
constexpr int kDuoBlock = 2 * kBlock;
template <class E>
struct DuoTraits {
#ifdef MI_DUO_CHUNK_OVERRIDE  // (A/B builds: scripts/build_variant.py ... -DMI_DUO_CHUNK_OVERRIDE=8)
    static constexpr int CHUNK = MI_DUO_CHUNK_OVERRIDE;
#else
    static constexpr int CHUNK = E::DUO_CHUNK;  // steps per phase (envs_classic.h: measured per environment; 2 costs 10 .. 14 % in barriers)
#endif
};

// env role: lane_step_fused<E, false> without the episode statistics; `bits`: 1 terminated, 2 truncated, 4 this was the autoreset step
template <class E>
MI_DEV void duo_env_step(const DevEnv &d, Lane<E> &L, typename E::Act a, ResetQueue<E> &q, float obs[E::OBS], double &reward, uint32_t &bits) {
    const bool resetting = (L.flags & kNeedsReset) != 0;
    if (__builtin_expect(resetting && !q.have, 0)) q.refill();  // rare: two episode ends within one refill period
    const double (&rs)[E::S] = q.rs;
    const uint32_t rflags = ResetQueue<E>::reset_flags(L.flags & ~kNeedsReset);
    double rew;
    bool te;
    uint32_t sflags = L.flags;
    if constexpr (E::SPLIT_TERMINAL) {
        E::integrate(L.s, sflags, a, L.trig);
#pragma unroll
        for (int k = 0; k < E::S; k++) L.s[k] = resetting ? rs[k] : L.s[k];
        E::obs(L.s, sflags, obs, L.trig);
        E::terminal_after_obs(L.s, L.trig, rew, te);
    } else if constexpr (E::AUX_REWARD) {  // the aux role evaluates the reward (from E::aux_pre of the state before this step): the dynamics alone
        E::advance(L.s, a, d.P, L.trig);
        rew = 0.0, te = false;
    } else {
        E::step(L.s, sflags, a, d.P, rew, te, L.trig);
    }
    const uint32_t elapsed = L.elapsed + 1u;  // TimeLimit.step (wrappers/common.py:129-133)
    const bool tr = d.max_steps > 0 && (int)elapsed >= d.max_steps;
    const bool done = !resetting && (te || tr);
    if constexpr (!E::SPLIT_TERMINAL) {
#pragma unroll
        for (int k = 0; k < E::S; k++) L.s[k] = resetting ? rs[k] : L.s[k];
    }
    L.flags = resetting ? rflags : (done ? (sflags | kNeedsReset) : sflags);
    L.elapsed = resetting ? 0u : elapsed;
    q.have = resetting ? 0u : q.have;
    if constexpr (!E::SPLIT_TERMINAL) E::obs(L.s, L.flags, obs, L.trig);
    reward = resetting ? 0.0 : rew;
    bits = (resetting ? 4u : 0u) | ((!resetting && te) ? 1u : 0u) | ((!resetting && tr) ? 2u : 0u);
}

// ... and the env role of an environment whose aux role derives the observation row and the flags from the state (E::AUX_DERIVES_FLAGS): the same
// step and autoreset select, then only the state words go over (E::aux_pack) -- no observation conversion, no flag word
template <class E>
MI_DEV void duo_env_step_state(const DevEnv &d, Lane<E> &L, typename E::Act a, ResetQueue<E> &q, double *w64, float *w32) {
    static_assert(!E::SPLIT_TERMINAL, "the split terminal test belongs to the observation, which this role does not take");
    const bool resetting = (L.flags & kNeedsReset) != 0;
    if (__builtin_expect(resetting && !q.have, 0)) q.refill();  // rare: two episode ends within one refill period
    const double (&rs)[E::S] = q.rs;
    const uint32_t rflags = ResetQueue<E>::reset_flags(L.flags & ~kNeedsReset);
    double rew;
    bool te;
    uint32_t sflags = L.flags;
    E::step(L.s, sflags, a, d.P, rew, te, L.trig);
    const uint32_t elapsed = L.elapsed + 1u;  // TimeLimit.step (wrappers/common.py:129-133)
    const bool tr = d.max_steps > 0 && (int)elapsed >= d.max_steps;
    const bool done = !resetting && (te || tr);
#pragma unroll
    for (int k = 0; k < E::S; k++) L.s[k] = resetting ? rs[k] : L.s[k];
    L.flags = resetting ? rflags : (done ? (sflags | kNeedsReset) : sflags);
    L.elapsed = resetting ? 0u : elapsed;
    q.have = resetting ? 0u : q.have;
    E::aux_pack(L.s, w64, w32);
}

// (Measured and not kept: the aux role split once more into "policy" and "book" wavefronts for three per SIMD: CartPole 78.9 us against 76.4 us --
//  more wavefronts do not help any more; s_setprio(3) for the env wavefronts: no change.  docs/classic_kernels.md)
template <class E>
__global__ __launch_bounds__(kDuoBlock) void rollout_duo_kernel(DevEnv d, RolloutPtrs io, ActionStream as, int T) {
    typedef typename E::Act Act;
    constexpr int ROLES = 2;
    constexpr int C = DuoTraits<E>::CHUNK;
    typedef Act ActLds;  // (a byte per Discrete action would save 28 KB of LDS at chunk 8 and costs 1 %: the widening on the env role)
    constexpr bool ACT_AHEAD = E::DUO_ACT_AHEAD;  // the env role requests step k + 1's action while step k runs (measured per environment, envs_classic.h)
    __shared__ ActLds sh_act[2][C + (ACT_AHEAD ? 1 : 0)][kBlock];  // (+ one row nobody writes: the env role's request for "the next step's action" after a chunk's last step)
    // what goes from the env role to the aux role: the observation row and a flag word -- or, for the environments whose aux role derives both from
    // the state (E::AUX_DERIVES_FLAGS, envs_classic.h), the state words
    constexpr bool DERIVE = E::AUX_DERIVES_FLAGS;
    static_assert(!DERIVE || E::REWARD_FROM_TERMINATED, "an aux role that derives the flags also derives the reward from them");
    constexpr int NF64 = E::AUX_F64 > 0 ? E::AUX_F64 : 1, NF32 = E::AUX_F32 > 0 ? E::AUX_F32 : 1;
    __shared__ double sh_w64[DERIVE ? 2 : 1][DERIVE ? C : 1][DERIVE ? kBlock : 1][NF64];
    __shared__ float sh_w32[(DERIVE && E::AUX_F32 > 0) ? 2 : 1][(DERIVE && E::AUX_F32 > 0) ? C : 1][(DERIVE && E::AUX_F32 > 0) ? kBlock : 1][NF32];
    __shared__ float sh_obs[DERIVE ? 1 : 2][DERIVE ? 1 : C][DERIVE ? 1 : kBlock][E::OBS];
    // (Measured and not kept: Pendulum with its reward -- three exact pow and an fmod, none of which feeds the next state -- evaluated by the aux role from
    //  the pre-step state passed through LDS: bit-identical, 168.7 us against 168.7 us for the one-role kernel.  Its instruction count is the limit.)
    constexpr bool AUXREW = E::AUX_REWARD;  // the aux role evaluates the reward from words about the state before the step (Pendulum)
    constexpr bool REW_OF_ACT = E::AUX_REWARD_OF_ACTION;  // ... or from the action it drew and the terminated flag (MountainCarContinuous)
    constexpr bool PASS_REWARD = !E::REWARD_FROM_TERMINATED && !AUXREW && !REW_OF_ACT;  // (not transferred when the aux role can compute it)
    __shared__ double sh_pre[AUXREW ? 2 : 1][AUXREW ? C : 1][AUXREW ? kBlock : 1][E::AUX_PRE];
    __shared__ double sh_rew[PASS_REWARD ? 2 : 1][PASS_REWARD ? C : 1][kBlock];
    // (the flag word's width is tuning, measured per environment at T = 128: CartPole +2.3 % with a dword, MountainCarContinuous +2.9 % with a byte)
    typedef typename std::conditional<E::REWARD_FROM_TERMINATED && E::OBS == 4, uint32_t, uint8_t>::type MI_DUO_FLAG_T;
    __shared__ MI_DUO_FLAG_T sh_bits[DERIVE ? 1 : 2][DERIVE ? 1 : C][DERIVE ? 1 : kBlock];
    __shared__ uint64_t sh_c[4][kBlock / 64];
    __shared__ double sh_r[kBlock / 64];
    tables_init<E>();
    const int role = threadIdx.x / kBlock;  // wavefronts j (env), j + 4 (aux / policy) and j + 8 (book) share a SIMD and 64 sub-environments
    const bool is_env = role == 0, is_policy = role == 1, is_book = role == ROLES - 1;
    const int slot = threadIdx.x & (kBlock - 1);
    const int i = blockIdx.x * kBlock + slot;
    const bool active = i < d.N;
    const size_t N = (size_t)d.N;
    LaneStats st = {0u, 0u, 0u, 0ull, 0.0};
    // env role
    Lane<E> L;
    ResetQueue<E> q;
    // aux role
    u128 astate = 0;
    double ep_ret = 0.0;
    int32_t ep_len = 0, ep_len_start = 0;  // (ep_len_start: 0 for a sub-environment whose first step is its autoreset step -- that episode was counted when it finished)
    uint32_t n_term = 0;
    uint32_t start_flags = 0;
    uint32_t aux_elapsed = 0, aux_needs_reset = 0;  // DERIVE: the aux role's own copy of the TimeLimit counter and of the pending-autoreset flag
    if (active) {
        if (is_env) {
            load_lane<E>(d, i, L);
            q.have = 0u;
            q.rng = load_rng(d, i);
        }
        if (is_book) {
            ep_ret = d.ep_ret[i], ep_len = d.ep_len[i];
            const uint32_t meta0 = d.meta[i];
            start_flags = meta0 >> kFlagShift;
            aux_elapsed = meta0 & kElapsedMask, aux_needs_reset = start_flags & kNeedsReset;
            ep_len_start = (start_flags & kNeedsReset) ? 0 : ep_len;
        }
        if (is_policy) {
            astate = make_u128(as.state_hi, as.state_lo);  // skip ahead by (i + 1) draws: one affine map per set bit of (i + 1)
            uint32_t delta = (uint32_t)i + 1u;
            for (int j = 0; delta; j++, delta >>= 1)
                if (delta & 1u) astate = as.pow2[j].mult * astate + as.pow2[j].plus;
        }
    }
    const int chunks = T / C;  // (the launcher sends a T that C does not divide to the one-role kernel)
#ifdef MI_DUO_TIMING
    unsigned long long t_work = 0, t_wait = 0, t_mark = __builtin_readcyclecounter();
#endif
    // One phase of one role.  (Round 6, last cut: a loop per ROLE instead of one loop with both roles' code in it.  The values
    // a role keeps in scalar registers across the phases -- the env role's float64 constants, the aux role's pointers and multipliers -- were live through the
    // other role's code as well: Pendulum's bookkeeping loop reloaded 11 spilled scalars per step.  Both loops meet at the same number of barriers.)
    auto phase = [&](auto env_tag, const int p) __attribute__((always_inline)) {
        constexpr bool ENV = decltype(env_tag)::value;
        if (active) {
            if constexpr (ENV) {
                const int c = p - 1;
                if (c >= 0 && c < chunks) {
                    const int buf = c & 1;
                    // (rolled: with a partner wavefront on the SIMD the LDS read of the action at the top of a step is covered, and four copies
                    //  of the step body -- libm slow paths included -- are 30 KB of instruction cache: measured +5 % against the unrolled form)
                    // (the queue's periodic refill, rollout_kernel's `t % kRefillPeriod == 0`: with chunks that divide the period the test leaves the step loop)
                    constexpr bool REFILL_PER_CHUNK = kRefillPeriod % C == 0;
                    if (REFILL_PER_CHUNK && ((c * C) & (kRefillPeriod - 1)) == 0 && !q.have) q.refill();
                    // ACT_AHEAD: the action of step k + 1 is requested while step k runs -- the rolled loop otherwise starts every step with an LDS round trip
                    ActLds a_next = sh_act[buf][0][slot];
#pragma unroll 1
                    for (int k = 0; k < C; k++) {
                        const int t = c * C + k;
                        if (!REFILL_PER_CHUNK && (t & (kRefillPeriod - 1)) == 0 && !q.have) q.refill();
                        const Act a_k = ACT_AHEAD ? (Act)a_next : (Act)sh_act[buf][k][slot];
                        if (ACT_AHEAD) a_next = sh_act[buf][k + 1][slot];
                        if constexpr (DERIVE) {
                            double w64[NF64];
                            float w32[NF32];
                            duo_env_step_state<E>(d, L, a_k, q, w64, w32);
#pragma unroll
                            for (int j = 0; j < E::AUX_F64; j++) sh_w64[buf][k][slot][j] = w64[j];
#pragma unroll
                            for (int j = 0; j < E::AUX_F32; j++) sh_w32[buf][k][slot][j] = w32[j];
                        } else {
                            float o[E::OBS];
                            double rew;
                            uint32_t bits;
                            uint32_t pre_pending = 0u;  // E::AUX_PRE_SQUARES: which of pre[]'s leading entries are still arguments, not squares (bit j)
                            if constexpr (AUXREW) {  // (of a sub-environment in its autoreset step too: the aux role discards that reward)
                                double pre[E::AUX_PRE];
                                pre_pending = E::aux_pre(L.s, pre);
#pragma unroll
                                for (int j = 0; j < E::AUX_PRE; j++) sh_pre[buf][k][slot][j] = pre[j];
                            }
                            duo_env_step<E>(d, L, a_k, q, o, rew, bits);
                            if constexpr (AUXREW && E::AUX_PRE_SQUARES > 0) bits |= pre_pending << 3;
#pragma unroll
                            for (int j = 0; j < E::OBS; j++) sh_obs[buf][k][slot][j] = o[j];
                            if constexpr (PASS_REWARD) sh_rew[buf][k][slot] = rew;
                            sh_bits[buf][k][slot] = (MI_DUO_FLAG_T)bits;
                        }
                    }
                }
            } else {
                const int c = p - 2;
                if (is_book && c >= 0) {  // what became of chunk c: episode statistics (RecordEpisodeStatistics order: the raw reward), totals, the trajectory rows
                    const int buf = c & 1;
                    if constexpr (AUXREW && E::AUX_PRE_SQUARES > 0) {
                        // The squares the env role could not take as plain products (flag bits 3 ..; 1 argument in 32): the exact pow routine, one pass per
                        // pending argument of this lane's chunk, all lanes of the wavefront side by side -- the wavefront runs as many passes as its
                        // unluckiest lane has pending arguments (2.7 on average for 2 x 8), not one per argument.  In place: a lane reads and writes
                        // only its own ring entries.
                        static_assert(E::AUX_PRE_SQUARES * C <= 32 && E::AUX_PRE_SQUARES <= 4 && sizeof(MI_DUO_FLAG_T) == 1, "pending bits: one word per chunk, bits 3 .. 6 of the flag byte");
                        uint32_t pend = 0u;
#pragma unroll
                        for (int k = 0; k < C; k++)
                            pend |= (((uint32_t)sh_bits[buf][k][slot] >> 3) & ((1u << E::AUX_PRE_SQUARES) - 1u)) << (E::AUX_PRE_SQUARES * k);
#pragma nounroll
                        while (pend) {
                            const int idx = __builtin_ctz(pend);
                            double *x = &sh_pre[buf][idx / E::AUX_PRE_SQUARES][slot][idx % E::AUX_PRE_SQUARES];
                            *x = E::aux_square(*x);
                            pend &= pend - 1u;
                        }
                    }
                    if constexpr (AUXREW) {
                        if constexpr (E::AUX_ACT_SQUARE) {
                            // ... and the float32 square of the (clipped) action the reward reads: the same scheme, in place in the action ring -- this chunk's
                            // actions were consumed by the env role a phase ago and the policy below refills this half only after the loop that follows
                            static_assert(C <= 32 && sizeof(ActLds) == sizeof(float), "one pending bit per step; the square takes the action's place");
                            uint32_t apend = 0u;
#pragma unroll
                            for (int k = 0; k < C; k++) {
                                const float u = E::act_square_arg((Act)sh_act[buf][k][slot]);
                                float h;
                                const bool plain = E::Math::sqf_is_plain(u, h);
                                sh_act[buf][k][slot] = (ActLds)(plain ? h : u);
                                apend |= plain ? 0u : 1u << k;
                            }
#pragma nounroll
                            while (apend) {
                                ActLds *x = &sh_act[buf][__builtin_ctz(apend)][slot];
                                *x = (ActLds)E::Math::sqf((float)*x);
                                apend &= apend - 1u;
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < C; k++) {
                        const size_t t = (size_t)c * C + k;
                        float o[E::OBS];
                        bool resetting, te, tr;
                        if constexpr (DERIVE) {
                            double w64[NF64];
                            float w32[NF32];
#pragma unroll
                            for (int j = 0; j < E::AUX_F64; j++) w64[j] = sh_w64[buf][k][slot][j];
#pragma unroll
                            for (int j = 0; j < E::AUX_F32; j++) w32[j] = sh_w32[buf][k][slot][j];
                            bool t0;
                            E::aux_unpack(w64, w32, d.P, o, t0);
                            // the vectoriser's state machine once more, on this role's own counters (sync_vector_env.py:277-329, wrappers/common.py:129-133):
                            // the same integers the env role holds, so the same flags
                            resetting = aux_needs_reset != 0;
                            const uint32_t el = aux_elapsed + 1u;
                            te = !resetting && t0;
                            tr = !resetting && d.max_steps > 0 && (int)el >= d.max_steps;
                            aux_elapsed = resetting ? 0u : el;
                            aux_needs_reset = (te || tr) ? 1u : 0u;
                        } else {
#pragma unroll
                            for (int j = 0; j < E::OBS; j++) o[j] = sh_obs[buf][k][slot][j];
                            const uint32_t bits = sh_bits[buf][k][slot];
                            resetting = (bits & 4u) != 0, te = (bits & 1u) != 0, tr = (bits & 2u) != 0;
                        }
                        double rew;
                        const bool done = te || tr;
                        if constexpr (E::REWARD_FROM_TERMINATED) {
                            // the reward is a function of the flags, so every total follows from three counters and the episode length (finish_totals,
                            // below): no float64 accumulation per step
                            rew = resetting ? 0.0 : E::reward_from_terminated(te, d.P);
                            // (an aux role that keeps its own TimeLimit counter needs no episode length per step: see the totals below.  Measured for
                            //  MountainCar: +3.8 %; for CartPole, with the counter handed over through LDS: -1 % -- profiles/r06_duo_three_changes_ab.txt)
                            if constexpr (!DERIVE) ep_len = resetting ? 0 : ep_len + 1;
                            st.reset_steps += resetting ? 1u : 0u;
                            st.episodes += done ? 1u : 0u;
                            n_term += te ? 1u : 0u;
                        } else {
                            if constexpr (AUXREW) {
                                double pre[E::AUX_PRE];
#pragma unroll
                                for (int j = 0; j < E::AUX_PRE; j++) pre[j] = sh_pre[buf][k][slot][j];
                                // (the action of this very step: chunk c was drawn two phases ago into the ring half the policy refills only AFTER this loop)
                                if constexpr (E::AUX_ACT_SQUARE)
                                    rew = resetting ? 0.0 : E::aux_reward_squared(pre, (float)sh_act[buf][k][slot]);  // (the ring holds the squares now, see above)
                                else
                                    rew = resetting ? 0.0 : E::aux_reward(pre, (Act)sh_act[buf][k][slot]);
                            } else if constexpr (REW_OF_ACT) {
                                rew = resetting ? 0.0 : E::reward_of_action(te, (Act)sh_act[buf][k][slot]);
                            } else {
                                rew = sh_rew[buf][k][slot];
                            }
                            const double ret = ep_ret + rew;
                            const int32_t len = ep_len + 1;
                            ep_ret = resetting ? 0.0 : ret;
                            ep_len = resetting ? 0 : len;
                            st.reset_steps += resetting ? 1u : 0u;
                            st.env_steps += resetting ? 0u : 1u;
                            st.episodes += done ? 1u : 0u;
                            st.return_sum += done ? ret : 0.0;
                            st.length_sum += done ? (uint64_t)len : 0ull;
                        }
                        store_row<E::OBS>(static_cast<float *>(io.obs) + (t * N + i) * E::OBS, o);
                        io.reward[t * N + i] = rew;
                        io.terminated[t * N + i] = te;
                        io.truncated[t * N + i] = tr;
                    }
                }
                if (is_policy && p < chunks) {  // the policy: action_space.sample() for chunk p (spaces/multi_discrete.py:176-178, spaces/box.py:463-465)
                    const int buf = p & 1;
#pragma unroll
                    for (int k = 0; k < C; k++) {
                        const size_t t = (size_t)p * C + k;
                        const Act a = action_of_state<E>(astate);
                        astate = as.jump_n.mult * astate + as.jump_n.plus;
                        sh_act[buf][k][slot] = (ActLds)a;
                        static_cast<Act *>(io.actions_out)[t * N + i] = a;
                    }
                }
            }
        }
#ifdef MI_DUO_TIMING
        {
            const unsigned long long now = __builtin_readcyclecounter();
            t_work += now - t_mark, t_mark = now;
        }
#endif
        __syncthreads();
#ifdef MI_DUO_TIMING
        {
            const unsigned long long now = __builtin_readcyclecounter();
            t_wait += now - t_mark, t_mark = now;
        }
#endif
    };
    if (__builtin_amdgcn_readfirstlane((int)is_env)) {  // (a role is a whole number of wavefronts: the branch is uniform)
#pragma unroll 1
        for (int p = 0; p < chunks + 2; p++) phase(std::true_type(), p);
    } else {
#pragma unroll 1
        for (int p = 0; p < chunks + 2; p++) phase(std::false_type(), p);
    }
#ifdef MI_DUO_TIMING  // scripts/r04/duo_timing.py: where each role's time goes
    if (blockIdx.x == 7 && (threadIdx.x & 63) == 0)
        printf("duo timing: wave %d (role %d) work %llu wait-at-barrier %llu cycles over %d phases\n", (int)(threadIdx.x >> 6), role, t_work, t_wait, chunks + 2);
#endif
    if (active) {
        if (is_env) {
            // (ep_ret / ep_len belong to the aux role: store_lane without them)
#pragma unroll
            for (int k = 0; k < E::S; k++) d.state[(size_t)k * d.N + i] = L.s[k];
            d.meta[i] = (L.elapsed & kElapsedMask) | (L.flags << kFlagShift);
            if (q.have) {  // hand the unconsumed draws back to the env's generator
#pragma unroll
                for (int k = 0; k < E::NDRAWS; k++) q.rng.unstep();
            }
            store_rng_state(d, i, q.rng);
        } else if (is_book) {
            if constexpr (E::REWARD_FROM_TERMINATED) {
                // The totals of the launch from the counters.  Every env-step lengthens exactly one episode by one, so the lengths of the episodes
                // that FINISHED add up to the env-steps taken plus what the running episode had at the start minus what it has now; and with a
                // reward that is r_T on the terminating step and r_N on every other one (E::reward_from_terminated: +-1 / 0 -- sums of such values are
                // exact in float64 in any order) the finished episodes' returns are r_N (lengths - terminations) + r_T terminations, the running
                // episode's return r_N x its length.  Same numbers as the step-by-step accumulation of rollout_kernel, bit for bit.
                const double r_n = E::reward_from_terminated(false, d.P), r_t = E::reward_from_terminated(true, d.P);
                // how the LAST step ended: its flags are still in the ring (nothing is carried through the phases for this)
                bool last_done, last_te;
                if constexpr (DERIVE) {  // (this role's own flag; whether the finishing step terminated: the last state is still in the ring)
                    double w64[NF64];
                    float w32[NF32], o[E::OBS];
#pragma unroll
                    for (int j = 0; j < E::AUX_F64; j++) w64[j] = sh_w64[((chunks > 0 ? chunks : 1) - 1) & 1][C - 1][slot][j];
#pragma unroll
                    for (int j = 0; j < E::AUX_F32; j++) w32[j] = sh_w32[((chunks > 0 ? chunks : 1) - 1) & 1][C - 1][slot][j];
                    bool t0;
                    E::aux_unpack(w64, w32, d.P, o, t0);
                    last_done = aux_needs_reset != 0, last_te = last_done && t0;
                } else {
                    const uint32_t last_bits = chunks > 0 ? (uint32_t)sh_bits[(chunks - 1) & 1][C - 1][slot] : ((start_flags & kNeedsReset) ? 2u : 0u);
                    last_done = (last_bits & 3u) != 0, last_te = (last_bits & 1u) != 0;
                }
                // DERIVE: the running episode's length is the TimeLimit counter once an autoreset step has zeroed both (they advance together from
                // there), and what it was plus the launch's steps otherwise
                if (DERIVE && chunks > 0) ep_len = st.reset_steps ? (int32_t)aux_elapsed : ep_len + chunks * C;
                st.env_steps = (uint32_t)(chunks * C) - st.reset_steps;
                // (an episode that finished in the very last step still sits in ep_len, waiting for its autoreset step: it IS among the finished ones)
                st.length_sum = (uint64_t)((int64_t)st.env_steps + (int64_t)ep_len_start - (last_done ? (int64_t)0 : (int64_t)ep_len));
                st.return_sum = r_n * (double)((int64_t)st.length_sum - (int64_t)n_term) + r_t * (double)n_term;
                st.return_sum = st.episodes ? st.return_sum : 0.0;  // (no finished episode: +0.0 like the accumulation, whatever the signs of r_n, r_t)
                ep_ret = (last_done && last_te) ? r_n * (double)(ep_len - 1) + r_t : r_n * (double)ep_len;
                if (chunks == 0) ep_ret = d.ep_ret[i];
            }
            d.ep_ret[i] = ep_ret, d.ep_len[i] = ep_len;
        }
    }
    // the workgroup's totals (block_accumulate for several roles: only the book-keeping wavefronts carry any)
    const int wave = slot >> 6, lane = threadIdx.x & 63;
    if (is_book) {
        const uint32_t c0 = wave_sum(st.env_steps), c1 = wave_sum(st.reset_steps), c2 = wave_sum(st.episodes);
        const uint64_t c3 = c2 ? wave_sum(st.length_sum) : 0;
        const double r = c2 ? wave_sum(st.return_sum) : 0.0;
        if (lane == 0) sh_c[0][wave] = c0, sh_c[1][wave] = c1, sh_c[2][wave] = c2, sh_c[3][wave] = c3, sh_r[wave] = r;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        uint64_t t = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; w++) t += sh_c[threadIdx.x][w];
        if (t) d.blk_count[(size_t)blockIdx.x * 4 + threadIdx.x] += t;
    } else if (threadIdx.x == 64) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; w++) t += sh_r[w];
        if (t != 0.0) d.blk_ret[blockIdx.x] += t;
    }
}

// ---------------------------------------------------------------------------------------------------------
// MI_CFG_SHARED_RNG: the reference's NumPy vector environment CartPoleVectorEnv (cartpole.py:353-505) -- one generator for all sub-environments,
// drawn component-major over the sub-environments that reset (see include/mi355env.h).  Same dynamics (E::step), other bookkeeping.
// ---------------------------------------------------------------------------------------------------------
MI_DEV u128 shared_advance(const SharedRng &sr, u128 state, uint64_t n) {  // `state` n steps further along the base generator's sequence
    for (int j = 0; n; j++, n >>= 1)
        if (n & 1ull) state = sr.pow2[j].mult * state + sr.pow2[j].plus;
    return state;
}
// The draws of one workgroup start at a position every lane shares (the call's first draw + the component's stride + the workgroup's prefix) and differ by
// less than the workgroup size: threads 0..3 skip ahead to the four shared positions (tens of 128-bit multiply-adds, once per workgroup, through LDS) and a
// lane adds its own offset < 256 (at most eight).  Call from every thread, before a __syncthreads(); `stride` = draws between two components.
MI_DEV void shared_block_bases(const SharedRng &sr, uint64_t first, uint64_t stride, uint64_t (*sh)[2]) {
    if (threadIdx.x < 4) {
        const u128 s = shared_advance(sr, make_u128(sr.words[0], sr.words[1]), first + (uint64_t)threadIdx.x * stride);
        sh[threadIdx.x][0] = (uint64_t)(s >> 64), sh[threadIdx.x][1] = (uint64_t)s;
    }
}
MI_DEV double shared_draw_from(const SharedRng &sr, const uint64_t (*sh)[2], int c, uint32_t offset) {  // draw `offset` after the workgroup's base of component c
    Pcg64 g;
    g.state = shared_advance(sr, make_u128(sh[c][0], sh[c][1]), offset), g.inc = make_u128(sr.words[2], sr.words[3]);
    return g.next_double();
}

// One workgroup, before every reset (fixed_draws = 4 N) / step (fixed_draws = 0: 4 k, k = the sub-environments that finished in the previous step):
// exclusive scan of the per-workgroup counts, and the stream bookkeeping -- the call in flight draws from pos_base, the next one after it.
// CartPoleVectorEnv.step asserts `self.action_space.contains(action)` for the WHOLE batch before it touches a sub-environment or the generator
// (cartpole.py:424-426).  A device caller's batch is checked by this pre-pass: one bad action marks the batch refused, and the scan / step kernels
// of this and of every later call return at once (no draw consumed, no sub-environment stepped) until the host has raised the error (raise_device_error).
template <class E>
__global__ __launch_bounds__(kBlock) void shared_validate_kernel(DevEnv d, const typename E::Act *actions, SharedRng sr) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < d.N && !E::valid(actions[i])) {
        *d.error = kErrInvalidAction;
        sr.words[7] = 1;
    }
}

__global__ __launch_bounds__(kBlock) void shared_scan_kernel(SharedRng sr, int grid, uint64_t fixed_draws) {
    __shared__ uint32_t wave_total[kBlock / 64];
    if (!fixed_draws && sr.words[7]) return;  // a refused batch (uniform)
    const int per = (grid + kBlock - 1) / kBlock, lo = threadIdx.x * per, hi = min(grid, lo + per);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t sum = 0;
    if (!fixed_draws)
        for (int b = lo; b < hi; b++) sum += sr.blk_done[b];
    // inclusive scan of the threads' chunk totals: Hillis-Steele inside each wavefront (cross-lane reads, no LDS round trips), the four wavefront totals through LDS
    uint32_t incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) wave_total[wave] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; w++) {
        before += w < wave ? wave_total[w] : 0u;
        total += wave_total[w];
    }
    if (threadIdx.x == 0) {
        const uint64_t consumed = sr.words[4];
        sr.words[5] = consumed, sr.words[6] = total;
        sr.words[4] = consumed + (fixed_draws ? fixed_draws : 4ull * total);
    }
    if (!fixed_draws) {
        uint32_t run = before + incl - sum;  // exclusive prefix of this thread's chunk
        for (int b = lo; b < hi; b++) {
            const uint32_t c = sr.blk_done[b];
            sr.blk_prefix[b] = run, run += c;
        }
    }
}

MI_DEV void shared_store_done_count(const SharedRng &sr, bool done) {  // this workgroup's entry of blk_done for the next step's scan
    __shared__ uint32_t sh[kBlock / 64];
    const uint64_t bal = __ballot(done);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = (uint32_t)__popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; w++) t += sh[w];
        sr.blk_done[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(kBlock) void shared_count_kernel(DevEnv d, SharedRng sr) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    shared_store_done_count(sr, i < d.N && ((d.meta[i] >> kFlagShift) & kNeedsReset));
}

// reset(): state[c][i] = uniform(low, high) from draw c * N + i (cartpole.py:493-500: size=(4, N))
template <class E>
__global__ __launch_bounds__(kBlock) void shared_reset_kernel(DevEnv d, SharedRng sr, float *obs) {
    __shared__ uint64_t sh_base[4][2];
    static_assert(E::NDRAWS == 4, "four state components: threads 0..3 prepare their bases");
    tables_init<E>();
    const int i = blockIdx.x * kBlock + threadIdx.x;
    shared_block_bases(sr, sr.words[5] + (uint64_t)blockIdx.x * kBlock, (uint64_t)d.N, sh_base);
    __syncthreads();
    if (i < d.N) {
        Lane<E> L;
        load_lane<E>(d, i, L);
        double u[E::NDRAWS];
#pragma unroll
        for (int c = 0; c < E::NDRAWS; c++) u[c] = shared_draw_from(sr, sh_base, c, threadIdx.x);
        L.flags = 0;
        E::reset_u(u, L.s, L.flags, sr.low, sr.high);
        L.elapsed = 0, L.ep_ret = 0.0, L.ep_len = 0;
        store_lane<E>(d, i, L);
        if (obs) {
            float o[E::OBS];
            E::obs(L.s, L.flags, o, L.trig);
            store_row<E::OBS>(obs + (size_t)i * E::OBS, o);
        }
    }
    shared_store_done_count(sr, false);
}

// step() (cartpole.py:421-479): a sub-environment that finished in the PREVIOUS step takes the j-th column of uniform(low, high, size=(4, k)) --
// j = its rank among the k such sub-environments, in index order: workgroup prefix (scan kernel) + wavefront ballots -- with steps 0, reward 0, not
// done; every other one integrates.  `-np.array(terminated, dtype=np.float32)` makes the Sutton-Barto reward of a surviving pole -0.0 (:466).
template <class E>
__global__ __launch_bounds__(kBlock) void shared_step_kernel(DevEnv d, StepPtrs io, SharedRng sr) {
    __shared__ uint32_t sh_wave[kBlock / 64];
    __shared__ uint64_t sh_base[4][2];
    static_assert(E::NDRAWS == 4, "four state components: threads 0..3 prepare their bases");
    if (sr.words[7]) return;  // a refused batch (shared_validate_kernel): nothing is touched, blk_done keeps the previous step's counts
    tables_init<E>();
    const int i = blockIdx.x * kBlock + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (sr.blk_done[blockIdx.x]) shared_block_bases(sr, sr.words[5] + (uint64_t)sr.blk_prefix[blockIdx.x], sr.words[6], sh_base);  // (workgroup-uniform test)
    LaneStats st = {0u, 0u, 0u, 0ull, 0.0};
    Lane<E> L;
    const bool active = i < d.N;
    if (active) load_lane<E>(d, i, L);
    const bool resetting = active && (L.flags & kNeedsReset);
    const uint64_t bal = __ballot(resetting);
    if (lane == 0) sh_wave[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t rank = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; w++) rank += sh_wave[w];
    bool done = false;
    if (active) {
        double rew = 0.0, ep_ret = 0.0;
        int32_t ep_len = 0;
        bool te = false, tr = false, untouched = false;
        if (resetting) {
            double u[E::NDRAWS];
#pragma unroll
            for (int c = 0; c < E::NDRAWS; c++) u[c] = shared_draw_from(sr, sh_base, c, rank);  // draw c * k + (workgroup prefix + rank) of this call
            L.flags = 0;
            E::reset_u(u, L.s, L.flags, sr.low, sr.high);
            L.elapsed = 0, L.ep_ret = 0.0, L.ep_len = 0;
            st.reset_steps++;
        } else {
            const typename E::Act a = static_cast<const typename E::Act *>(io.actions)[i];
            if (!E::valid(a)) {  // cartpole.py:424-426 asserts before it touches anything: the sticky error word, this sub-environment untouched
                *d.error = kErrInvalidAction;
                untouched = true;
            } else {
                E::step(L.s, L.flags, a, d.P, rew, te, L.trig);
                if (d.P.p[0] != 0.0 && !te) rew = -0.0;
                L.elapsed += 1;
                tr = d.max_steps > 0 && (int)L.elapsed >= d.max_steps;  // self.steps >= self.max_episode_steps (:461)
                L.ep_ret += rew, L.ep_len += 1;
                st.env_steps++;
                done = te || tr;
                if (done) {
                    ep_ret = L.ep_ret, ep_len = L.ep_len;
                    st.episodes++, st.return_sum += L.ep_ret, st.length_sum += (uint64_t)L.ep_len;
                    L.flags |= kNeedsReset;  // self.prev_done (:479)
                }
            }
        }
        if (!untouched) store_lane<E>(d, i, L);
        float o[E::OBS];
        E::obs(L.s, L.flags, o, L.trig);
        if (io.obs) store_row<E::OBS>(io.obs + (size_t)i * E::OBS, o);
        if (io.reward) io.reward[i] = rew;
        if (io.terminated) io.terminated[i] = te;
        if (io.truncated) io.truncated[i] = tr;
        if (io.ep_ret) io.ep_ret[i] = ep_ret;
        if (io.ep_len) io.ep_len[i] = ep_len;
    }
    shared_store_done_count(sr, done);
    block_accumulate(d, st);
}

// the policy of a shared-generator rollout's step t: draw t * N + i of the batched action space's stream (what rollout_kernel computes inline)
template <class E>
__global__ __launch_bounds__(kBlock) void shared_sample_kernel(DevEnv d, ActionStream as, int t, typename E::Act *out) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= d.N) return;
    u128 s = make_u128(as.state_hi, as.state_lo);
    uint64_t n = (uint64_t)t * (uint64_t)d.N + (uint64_t)i + 1ull;
    for (int j = 0; n; j++, n >>= 1)
        if (n & 1ull) s = as.pow2[j].mult * s + as.pow2[j].plus;
    out[i] = action_of_state<E>(s);
}

#ifndef MI_CLASSIC_TU
// ---------------------------------------------------------------------------------------------------------
// MuJoCo-family kernels (mjx_kernels.h): same vectoriser state machine, float64 observations, float32 action rows
// ---------------------------------------------------------------------------------------------------------
struct MjStepPtrs {
    const void *actions;  // [N][NU] float32 rows, or float64 rows taken un-rounded (act_f64; mujoco_env.py:148 data.ctrl[:] = ctrl)
    double *obs, *reward;
    uint8_t *terminated, *truncated;
    double *final_obs, *ep_ret;
    int32_t *ep_len;
    double *info, *final_info;
    int obs_dim;
    const double *extras;  // [N][EX_TOTAL] rows written by mj_physics_kernel, or nullptr (one-lane simulator inside the step kernel)
    int act_f64;
};

template <class E>
struct MjLane {
    double s[E::S];
    uint32_t elapsed, flags;
    double ep_ret;
    int32_t ep_len;
};
template <class E>
MI_DEV void mj_load(const DevEnv &d, int i, MjLane<E> &L) {
    for (int k = 0; k < E::S; k++) L.s[k] = d.state[(size_t)k * d.N + i];
    const uint32_t m = d.meta[i];
    L.elapsed = m & kElapsedMask, L.flags = m >> kFlagShift;
    L.ep_ret = d.ep_ret[i], L.ep_len = d.ep_len[i];
}
template <class E>
MI_DEV void mj_store(const DevEnv &d, int i, const MjLane<E> &L) {
    for (int k = 0; k < E::S; k++) d.state[(size_t)k * d.N + i] = L.s[k];
    d.meta[i] = (L.elapsed & kElapsedMask) | (L.flags << kFlagShift);
    d.ep_ret[i] = L.ep_ret, d.ep_len[i] = L.ep_len;
}
template <class E>
MI_DEV void mj_autoreset(const DevEnv &d, int i, MjLane<E> &L, double *obs) {
    Pcg64 rng = load_rng(d, i);
    E::reset(rng, L.s, d.P, obs);
    store_rng_state(d, i, rng);
    L.elapsed = 0, L.ep_ret = 0.0, L.ep_len = 0;
}

// one lockstep step of one MuJoCo sub-environment; obs / info rows are written straight to their destination
// `extras` != nullptr: the physics of this step was already advanced by mj_physics_kernel (L.s holds the new qpos / qvel
// and the old tracked point); nullptr: the one-lane simulator runs here.
template <class E, int MODE, bool COOP = false>
MI_DEV void mj_lane_step(const DevEnv &d, int i, MjLane<E> &L, const mjx::ActRow action, double *obs, double *final_obs, double *info,
                         double &reward, bool &te, bool &tr, double &out_ret, int32_t &out_len, LaneStats &st,
                         const double *extras = nullptr, double *final_info = nullptr) {
    te = tr = false, reward = 0.0;
    if (MODE == MI_AUTORESET_NEXT_STEP && (L.flags & kNeedsReset)) {
        mj_autoreset<E>(d, i, L, obs);
        if (info) E::reset_info(L.s, info);
        st.reset_steps++;
    } else if (MODE == MI_AUTORESET_DISABLED && (L.flags & kNeedsReset)) {
        *d.error = kErrDisabledStepped;
        out_ret = 0.0, out_len = 0;
        return;
    } else {
        if (COOP) {
            typedef mjx::coop::Sim<typename E::Model, E::COOP_G> S;
            typename E::StepExtras x;
            E::after_from_extras(L.s, extras, x.after);
            x.cfrc = reinterpret_cast<const double (*)[6]>(extras + S::EX_CFRC);
            x.cinert = reinterpret_cast<const double (*)[10]>(extras + S::EX_CINERT);
            x.cvel = reinterpret_cast<const double (*)[6]>(extras + S::EX_CVEL);
            x.qfrc_actuator = extras + S::EX_QFA;
            x.qfrc_constraint = nullptr;
            x.ten = extras + S::EX_TEN;
            const double before[2] = {L.s[E::NQ + 2 * E::NV], L.s[E::NQ + 2 * E::NV + 1]};
            E::finish(L.s, before, x, action, d.P, obs, reward, te, info);
        } else {
            E::step(L.s, action, d.P, obs, reward, te, info, d.solver_newton != 0);
        }
        L.elapsed += 1;
        tr = d.max_steps > 0 && (int)L.elapsed >= d.max_steps;
        L.ep_ret += reward, L.ep_len += 1;
        st.env_steps++;
    }
    const bool done = te || tr;
    out_ret = done ? L.ep_ret : 0.0, out_len = done ? L.ep_len : 0;
    if (done) st.episodes++, st.return_sum += L.ep_ret, st.length_sum += (uint64_t)L.ep_len;
    if (MODE == MI_AUTORESET_SAME_STEP && done) {
        if (final_obs)
            for (int k = 0; k < E::obs_dim(d.P); k++) final_obs[k] = obs[k];
        if (info && final_info)  // sync_vector_env.py:309-317: the finishing step's info goes to "final_info" ...
            for (int k = 0; k < E::INFO; k++) final_info[k] = info[k];
        mj_autoreset<E>(d, i, L, obs);
        if (info) E::reset_info(L.s, info);  // ... and the top-level entries of this sub-env are its reset info (:319)
    }
    if (done && MODE != MI_AUTORESET_SAME_STEP)
        L.flags |= kNeedsReset;
    else
        L.flags &= ~kNeedsReset;
}

template <class E, int MODE, bool COOP>
__global__ __launch_bounds__(kBlock) void mj_step_kernel(DevEnv d, MjStepPtrs io) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    LaneStats st = {0u, 0u, 0u, 0ull, 0.0};
    if (i < d.N) {
        MjLane<E> L;
        mj_load<E>(d, i, L);
        double reward, out_ret;
        int32_t out_len;
        bool te, tr;
        const mjx::ActRow a = {static_cast<const char *>(io.actions) + (size_t)i * E::NU * (io.act_f64 ? 8 : 4), io.act_f64 != 0};
        mj_lane_step<E, MODE, COOP>(d, i, L, a, io.obs + (size_t)i * io.obs_dim,
                              io.final_obs ? io.final_obs + (size_t)i * io.obs_dim : nullptr,
                              io.info ? io.info + (size_t)i * E::INFO : nullptr, reward, te, tr, out_ret, out_len, st,
                              COOP ? io.extras + (size_t)i * mjx::coop::Sim<typename E::Model, E::COOP_G>::EX_TOTAL : nullptr,
                              io.final_info ? io.final_info + (size_t)i * E::INFO : nullptr);
        mj_store<E>(d, i, L);
        if (io.reward) io.reward[i] = reward;
        if (io.terminated) io.terminated[i] = te;
        if (io.truncated) io.truncated[i] = tr;
        if (io.ep_ret) io.ep_ret[i] = out_ret;
        if (io.ep_len) io.ep_len[i] = out_len;
    }
    block_accumulate(d, st);
}

template <class E>
__global__ __launch_bounds__(kBlock) void mj_reset_kernel(DevEnv d, const uint8_t *mask, double *obs, int obs_dim) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= d.N || (mask && !mask[i])) return;
    MjLane<E> L;
    mj_load<E>(d, i, L);
    L.flags &= ~kNeedsReset;
    mj_autoreset<E>(d, i, L, obs ? obs + (size_t)i * obs_dim : nullptr);
    mj_store<E>(d, i, L);
}

// fused rollout: T steps per launch, Box action space sampled on device from the batched space's single PCG64 stream
// (draw number (t*N + i)*NU + u belongs to lane i, component u, step t)
template <class E, int MODE, bool SAMPLE>
__global__ __launch_bounds__(kBlock) void mj_rollout_kernel(DevEnv d, RolloutPtrs io, ActionStream as, int T, int obs_dim, int in_f64) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    LaneStats st = {0u, 0u, 0u, 0ull, 0.0};
    if (i < d.N) {
        MjLane<E> L;
        mj_load<E>(d, i, L);
        u128 astate = 0;
        const u128 ainc = make_u128(as.inc_hi, as.inc_lo);
        if (SAMPLE) {
            astate = make_u128(as.state_hi, as.state_lo);
            uint64_t delta = (uint64_t)i * E::NU + 1u;
            for (int j = 0; delta; j++, delta >>= 1)
                if (delta & 1u) astate = as.pow2[j].mult * astate + as.pow2[j].plus;
        }
        const size_t N = (size_t)d.N;
        double scratch_obs[E::MAX_OBS];
        for (int t = 0; t < T; t++) {
            float a[E::NU];
            mjx::ActRow row = {a, false};
            if (SAMPLE) {
                for (int u = 0; u < E::NU; u++) {
                    const uint64_t hi = (uint64_t)(astate >> 64), lo = (uint64_t)astate, x = hi ^ lo;
                    const unsigned rot = (unsigned)(hi >> 58);
                    const uint64_t out = (x >> rot) | (x << ((0u - rot) & 63u));
                    const double lo_b = (double)(float)E::Model::actuator_ctrlrange[u][0], hi_b = (double)(float)E::Model::actuator_ctrlrange[u][1];
                    a[u] = (float)(lo_b + (hi_b - lo_b) * ((double)(out >> 11) * (1.0 / 9007199254740992.0)));
                    if (u + 1 < E::NU) astate = astate * pcg_mult() + ainc;
                }
                astate = as.jump_n.mult * astate + as.jump_n.plus;
                if (io.actions_out)
                    for (int u = 0; u < E::NU; u++) static_cast<float *>(io.actions_out)[(t * N + i) * E::NU + u] = a[u];
            } else {  // the caller's rows, read in place
                row.p = static_cast<const char *>(io.actions_in) + (t * N + i) * E::NU * (in_f64 ? 8 : 4), row.f64 = in_f64 != 0;
            }
            double reward, out_ret;
            int32_t out_len;
            bool te, tr;
            double *obs = io.obs ? static_cast<double *>(io.obs) + (t * N + i) * obs_dim : scratch_obs;
            mj_lane_step<E, MODE>(d, i, L, row, obs, nullptr, nullptr, reward, te, tr, out_ret, out_len, st);
            if (io.reward) io.reward[t * N + i] = reward;
            if (io.terminated) io.terminated[t * N + i] = te;
            if (io.truncated) io.truncated[t * N + i] = tr;
        }
        mj_store<E>(d, i, L);
    }
    block_accumulate(d, st);
}

// ---- cooperative physics (mjx_coop.h): G lanes per sub-environment, 64 / G sub-environments per wavefront --------------
// mj_physics_kernel lives in mjx_physics.h and is instantiated in physics16.hip / physics32.hip (their own compiler settings); here it
// is only launched: mi_phys::launch16 / launch32.

// Box.sample() of the batched action space for step t of a rollout: draw number (t N + i) NU + u of the stream
template <class E>
__global__ __launch_bounds__(kBlock) void mj_sample_kernel(DevEnv d, ActionStream as, int t, float *out) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= d.N) return;
    const u128 ainc = make_u128(as.inc_hi, as.inc_lo);
    u128 astate = make_u128(as.state_hi, as.state_lo);
    uint64_t delta = ((uint64_t)t * (uint64_t)d.N + (uint64_t)i) * E::NU + 1u;
    for (int j = 0; delta; j++, delta >>= 1)
        if (delta & 1u) astate = as.pow2[j].mult * astate + as.pow2[j].plus;
    for (int u = 0; u < E::NU; u++) {
        const uint64_t hi = (uint64_t)(astate >> 64), lo = (uint64_t)astate, x = hi ^ lo;
        const unsigned rot = (unsigned)(hi >> 58);
        const uint64_t o = (x >> rot) | (x << ((0u - rot) & 63u));
        const double lo_b = (double)(float)E::Model::actuator_ctrlrange[u][0], hi_b = (double)(float)E::Model::actuator_ctrlrange[u][1];
        out[(size_t)i * E::NU + u] = (float)(lo_b + (hi_b - lo_b) * ((double)(o >> 11) * (1.0 / 9007199254740992.0)));
        astate = astate * pcg_mult() + ainc;
    }
}

// ---------------------------------------------------------------------------------------------------------
// ToyText: finite MDPs.  state row = {state index, probability of the last transition}; integer lookups only.
//   step:  i = argmax(cumsum(p) > rng.random()) (toy_text/utils.py:4-8), then P[s][a][i]  (frozen_lake.py:324-335)
//   reset: s = argmax(cumsum(initial_state_distrib) > rng.random())                        (frozen_lake.py:337-348)
// ---------------------------------------------------------------------------------------------------------
MI_DEV int tab_categorical(const double *csprob, int n, Pcg64 &rng) {
    const double u = rng.next_double();
    // first index with csprob[k] > u (np.argmax(np.cumsum(p) > u); 0 when there is none).  The first entries by scan -- transition rows
    // have <= 3 outcomes, FrozenLake / CliffWalking start in state 0 -- the rest of a long row (Taxi's 500-state initial distribution) by
    // bisection, valid because cumulative sums are non-decreasing: 8 probes instead of a scan that costs a wavefront its slowest lane.
    const int head = n < 4 ? n : 4;
    for (int k = 0; k < head; k++)
        if (csprob[k] > u) return k;
    int lo = head, hi = n;  // invariant: csprob[k] <= u for k < lo, csprob[k] > u for k >= hi
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (csprob[mid] > u)
            hi = mid;
        else
            lo = mid + 1;
    }
    return lo < n ? lo : 0;
}
struct TabLane {
    double s, prob;
    double aux;  // Taxi with fickle_passenger only: fickle_step | the generator's buffered 32-bit half << 1 (third state row)
    uint32_t elapsed, flags;
    double ep_ret;
    int32_t ep_len;
};
// Taxi's fickle passenger (taxi.py:436-451, :462-464): params[2] != 0 switches the rule on, params[3] = fickle_probability
MI_DEV bool tab_fickle(const DevEnv &d) { return d.tab_fickle_rows != 0; }
// The tabular kernels serve three kinds of environment (a plain transition table, Taxi with the fickle passenger, Blackjack) behind run-time tests; the
// fused rollout is instantiated per kind (TK: kTabPlain / kTabFickle / kTabBlackjack; kTabAny = decide at run time, the per-step kernels), so that a
// FrozenLake rollout carries neither Blackjack's dealer loop nor Taxi's re-draw (round 6: the kernel issues one instruction per ~5 cycles, profiles/r06_acrobot_toytext_root_cause.txt).
enum { kTabAny = -1, kTabPlain = 0, kTabFickle = 1, kTabBlackjack = 2 };
template <int TK>
MI_DEV bool tab_is_fickle(const DevEnv &d) { return TK == kTabAny ? d.tab_fickle_rows != 0 : TK == kTabFickle; }
template <int TK = kTabAny>
MI_DEV void tab_load(const DevEnv &d, int i, TabLane &L) {
    L.s = d.state[i], L.prob = d.state[(size_t)d.N + i];
    L.aux = tab_is_fickle<TK>(d) ? d.state[(size_t)2 * d.N + i] : 0.0;
    const uint32_t m = d.meta[i];
    L.elapsed = m & kElapsedMask, L.flags = m >> kFlagShift;
    L.ep_ret = d.ep_ret[i], L.ep_len = d.ep_len[i];
}
template <int TK = kTabAny>
MI_DEV void tab_store(const DevEnv &d, int i, const TabLane &L) {
    d.state[i] = L.s, d.state[(size_t)d.N + i] = L.prob;
    if (tab_is_fickle<TK>(d)) d.state[(size_t)2 * d.N + i] = L.aux;
    d.meta[i] = (L.elapsed & kElapsedMask) | (L.flags << kFlagShift);
    d.ep_ret[i] = L.ep_ret, d.ep_len[i] = L.ep_len;
}
// ---- Blackjack-v1 (gymnasium/envs/toy_text/blackjack.py:17-232) rides on the tabular kernels (d.tab.nS < 0) --------------
// state word  = player raw sum | has-ace << 6 | two-cards << 7 | dealer card 0 << 8 | dealer card 1 << 12
// second word = the generator's buffered 32-bit half (1 << 32 | value; 0 = empty): draw_card is np_random.choice(deck) =
// Lemire's bounded uint32 on next_uint32, and PCG64's next_uint32 hands out the low, then the high half of one 64-bit output.
MI_DEV bool is_blackjack(const DevEnv &d) { return d.tab.nS < 0; }
template <int TK>
MI_DEV bool tab_is_blackjack(const DevEnv &d) { return TK == kTabAny ? d.tab.nS < 0 : TK == kTabBlackjack; }
MI_DEV uint32_t bj_next32(Pcg64 &rng, double &aux_slot) {
    const uint64_t aux = (uint64_t)aux_slot;
    if (aux >> 32) {
        aux_slot = 0.0;
        return (uint32_t)aux;
    }
    const uint64_t x = rng.next64();
    aux_slot = (double)((1ull << 32) | (x >> 32));
    return (uint32_t)x;
}
MI_DEV uint32_t bj_bounded(Pcg64 &rng, double &aux, uint32_t n) {
    uint64_t m = (uint64_t)bj_next32(rng, aux) * n;
    uint32_t left = (uint32_t)m;
    if (left < n) {
        const uint32_t thr = (uint32_t)((0x100000000ull - n) % n);
        while (left < thr) m = (uint64_t)bj_next32(rng, aux) * n, left = (uint32_t)m;
    }
    return (uint32_t)(m >> 32);
}
MI_DEV int bj_card(Pcg64 &rng, double &aux) {  // deck = [1..10, 10, 10, 10]
    const uint32_t k = bj_bounded(rng, aux, 13);
    return k < 9 ? (int)k + 1 : 10;
}
MI_DEV int bj_sum_hand(int raw, int ace) { return (ace && raw + 10 <= 21) ? raw + 10 : raw; }
MI_DEV void bj_obs(double s, int64_t *o) {  // _get_obs: (player sum with a usable ace as 11, dealer's first card, usable ace)
    const int64_t p = (int64_t)s;
    const int psum = (int)(p & 63), pace = (int)((p >> 6) & 1);
    o[0] = bj_sum_hand(psum, pace), o[1] = (p >> 8) & 15, o[2] = pace && psum + 10 <= 21;
}
MI_DEV void bj_reset(Pcg64 &rng, double &s, double &aux) {  // blackjack.py:181-202 incl. the render-only suit / face draws
    const int d0 = bj_card(rng, aux), d1 = bj_card(rng, aux), p0 = bj_card(rng, aux), p1 = bj_card(rng, aux);
    (void)bj_bounded(rng, aux, 4);
    if (d0 == 10) (void)bj_bounded(rng, aux, 3);
    s = (double)((p0 + p1) | ((int)(p0 == 1 || p1 == 1) << 6) | (1 << 7) | (d0 << 8) | (d1 << 12));
}
MI_DEV void bj_step(Pcg64 &rng, double &s, double &aux, int64_t action, bool natural, bool sab, double &reward, bool &te) {  // :143-174
    const int64_t p = (int64_t)s;
    int psum = (int)(p & 63), pace = (int)((p >> 6) & 1), ptwo = (int)((p >> 7) & 1);
    const int d0 = (int)((p >> 8) & 15), d1 = (int)((p >> 12) & 15);
    if (action) {  // hit
        const int c = bj_card(rng, aux);
        psum += c, pace |= c == 1, ptwo = 0;
        te = bj_sum_hand(psum, pace) > 21;
        reward = te ? -1.0 : 0.0;
    } else {  // stick: the dealer draws to 17, then the hands are scored
        int dsum = d0 + d1, dace = d0 == 1 || d1 == 1, dtwo = 1;
        while (bj_sum_hand(dsum, dace) < 17) {
            const int c = bj_card(rng, aux);
            dsum += c, dace |= c == 1, dtwo = 0;
        }
        const int ph = bj_sum_hand(psum, pace), dh = bj_sum_hand(dsum, dace);
        const int ps = ph > 21 ? 0 : ph, ds = dh > 21 ? 0 : dh;
        te = true;
        reward = (double)(ps > ds) - (double)(ps < ds);
        const bool pnat = ptwo && pace && psum == 11, dnat = dtwo && dace && dsum == 11;  // sorted(hand) == [1, 10]
        if (sab && pnat && !dnat)
            reward = 1.0;
        else if (!sab && natural && pnat && reward == 1.0)
            reward = 1.5;
    }
    s = (double)(psum | (pace << 6) | (ptwo << 7) | (d0 << 8) | (d1 << 12));
}

// `held`: the lane's generator kept in registers by a fused rollout (loaded before its loop, stored after it); nullptr = the per-step
// kernels, which load and store the stream around every use.
template <int TK = kTabAny>
MI_DEV void tab_autoreset(const DevEnv &d, int i, TabLane &L, Pcg64 *held = nullptr) {
    Pcg64 local;
    if (!held) local = load_rng(d, i);
    Pcg64 &rng = held ? *held : local;
    if (tab_is_blackjack<TK>(d)) {
        bj_reset(rng, L.s, L.prob);
    } else {
        const size_t tb = d.tab.env_table ? (size_t)d.tab.env_table[i] : 0;  // the sub-environment's own table (its own random map, ...)
        L.s = (double)tab_categorical(d.tab.isd + tb * d.tab.nS, d.tab.nS, rng), L.prob = 1.0;
        if (tab_is_fickle<TK>(d)) {  // taxi.py:462-464: fickle_step = fickle_passenger and np_random.random() < fickle_probability -- one more draw
            const double flag = rng.next_double() < d.P.p[3] ? 1.0 : 0.0;
            L.aux = floor(L.aux * 0.5) * 2.0 + flag;
        }
    }
    if (!held) store_rng_state(d, i, rng);
    L.elapsed = 0, L.ep_ret = 0.0, L.ep_len = 0;
}
// observation row of a tabular lane: the state index, or Blackjack's three integers
template <int TK = kTabAny>
MI_DEV void tab_write_obs(const DevEnv &d, double s, int64_t *base, size_t row) {
    if (tab_is_blackjack<TK>(d))
        bj_obs(s, base + 3 * row);
    else
        base[row] = (int64_t)s;
}
template <int MODE, int TK = kTabAny>
MI_DEV void tab_lane_step(const DevEnv &d, int i, TabLane &L, int64_t a, int64_t &obs, int64_t &final_obs, bool &has_final, double &reward,
                          bool &te, bool &tr, double &out_ret, int32_t &out_len, LaneStats &st, Pcg64 *held = nullptr, double *final_prob = nullptr) {
    te = tr = false, reward = 0.0, has_final = false;
    if (MODE == MI_AUTORESET_NEXT_STEP && (L.flags & kNeedsReset)) {
        tab_autoreset<TK>(d, i, L, held);
        st.reset_steps++;
    } else if (MODE == MI_AUTORESET_DISABLED && (L.flags & kNeedsReset)) {
        *d.error = kErrDisabledStepped;
        obs = (int64_t)L.s, out_ret = 0.0, out_len = 0;
        return;
    } else {
        if (a < 0 || a >= d.tab.nA) {  // like the DISABLED case above: report, leave the lane untouched
            *d.error = kErrInvalidAction;
            obs = (int64_t)L.s, out_ret = 0.0, out_len = 0;
            return;
        }
        Pcg64 local;
        if (!held) local = load_rng(d, i);
        Pcg64 &rng = held ? *held : local;
        if (tab_is_blackjack<TK>(d)) {
            bj_step(rng, L.s, L.prob, a, d.P.p[0] != 0.0, d.P.p[1] != 0.0, reward, te);
            if (!held) store_rng_state(d, i, rng);
        } else {
            const size_t tb = d.tab.env_table ? (size_t)d.tab.env_table[i] : 0;
            const size_t cell = (tb * d.tab.nS + (size_t)L.s) * d.tab.nA + (size_t)a, row = cell * d.tab.K;
            const int k = tab_categorical(d.tab.csprob + row, d.tab.count[cell], rng);
            int next = d.tab.next[row + k];
            if (tab_is_fickle<TK>(d) && ((int64_t)L.aux & 1)) {
                // taxi.py:436-451: the passenger was in the taxi before this step (shadow pass_loc == 4) and the step moved the taxi: once per episode
                // the destination is re-drawn among the other three -- Generator.choice = a Lemire-bounded 32-bit draw on the buffered halves (bj_bounded)
                const int old = (int)L.s;
                const int srow = old / 100, scol = old / 20 % 5, spass = old / 4 % 5, sdest = old % 4;
                const int nrow = next / 100, ncol = next / 20 % 5, npass = next / 4 % 5;
                if (spass == 4 && (nrow != srow || ncol != scol)) {
                    double half = floor(L.aux * 0.5);
                    const int pick = (int)bj_bounded(rng, half, 3);
                    const int dest = pick + (pick >= sdest ? 1 : 0);  // possible_destinations = [i for i in range(4) if i != shadow_dest_idx]
                    L.aux = half * 2.0;                                // fickle_step = False
                    next = ((nrow * 5 + ncol) * 5 + npass) * 4 + dest;
                }
            }
            if (!held) store_rng_state(d, i, rng);
            L.s = (double)next, L.prob = d.tab.prob[row + k];
            reward = d.tab.reward[row + k], te = d.tab.term[row + k] != 0;
        }
        L.elapsed += 1;
        tr = d.max_steps > 0 && (int)L.elapsed >= d.max_steps;
        L.ep_ret += reward, L.ep_len += 1;
        st.env_steps++;
    }
    const bool done = te || tr;
    out_ret = done ? L.ep_ret : 0.0, out_len = done ? L.ep_len : 0;
    if (done) st.episodes++, st.return_sum += L.ep_ret, st.length_sum += (uint64_t)L.ep_len;
    if (MODE == MI_AUTORESET_SAME_STEP && done) {
        final_obs = (int64_t)L.s, has_final = true;
        if (final_prob) *final_prob = L.prob;  // info["prob"] of the finishing transition ("final_info"); the reset then reports prob = 1
        tab_autoreset<TK>(d, i, L, held);
    }
    obs = (int64_t)L.s;
    if (done && MODE != MI_AUTORESET_SAME_STEP)
        L.flags |= kNeedsReset;
    else
        L.flags &= ~kNeedsReset;
}

struct TabStepPtrs {
    const int64_t *actions;
    int64_t *obs;
    double *reward;
    uint8_t *terminated, *truncated;
    int64_t *final_obs;
    double *ep_ret;
    int32_t *ep_len;
    double *info, *final_info;
    uint64_t *act_lane;  // SAMPLE (mi_step with actions == NULL): as StepPtrs
    int64_t *actions_out;
    PcgJump act_jump;
};
template <int MODE, bool SAMPLE = false>
__global__ __launch_bounds__(kBlock) void tab_step_kernel(DevEnv d, TabStepPtrs io) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    LaneStats st = {0u, 0u, 0u, 0ull, 0.0};
    if (i < d.N) {
        TabLane L;
        tab_load(d, i, L);
        int64_t obs, fin = 0;
        double reward, out_ret;
        int32_t out_len;
        bool te, tr, has_final;
        double fin_prob = 0.0;
        int64_t a;
        if constexpr (SAMPLE) {  // (random(N) * nvec).astype(int64): draw pos + i of the batched MultiDiscrete's generator (spaces/multi_discrete.py:176-178)
            const u128 astate = make_u128(io.act_lane[i], io.act_lane[(size_t)d.N + i]);
            a = (int64_t)((double)(pcg_output(astate) >> 11) * (1.0 / 9007199254740992.0) * (double)d.tab.nA);
            const u128 next = io.act_jump.mult * astate + io.act_jump.plus;
            io.act_lane[i] = (uint64_t)(next >> 64), io.act_lane[(size_t)d.N + i] = (uint64_t)next;
            if (io.actions_out) io.actions_out[i] = a;
        } else {
            a = io.actions[i];
        }
        tab_lane_step<MODE>(d, i, L, a, obs, fin, has_final, reward, te, tr, out_ret, out_len, st, nullptr, &fin_prob);
        tab_store(d, i, L);
        if (io.obs) tab_write_obs(d, (double)obs, io.obs, (size_t)i);
        if (io.reward) io.reward[i] = reward;
        if (io.terminated) io.terminated[i] = te;
        if (io.truncated) io.truncated[i] = tr;
        if (io.final_obs && has_final) tab_write_obs(d, (double)fin, io.final_obs, (size_t)i);
        if (io.ep_ret) io.ep_ret[i] = out_ret;
        if (io.ep_len) io.ep_len[i] = out_len;
        if (io.info && !is_blackjack(d)) io.info[i] = L.prob;
        if (io.final_info && has_final && !is_blackjack(d)) io.final_info[i] = fin_prob;
    }
    block_accumulate(d, st);
}
__global__ __launch_bounds__(kBlock) void tab_reset_kernel(DevEnv d, const uint8_t *mask, int64_t *obs) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= d.N || (mask && !mask[i])) return;
    TabLane L;
    tab_load(d, i, L);
    L.flags &= ~kNeedsReset;
    tab_autoreset(d, i, L);
    tab_store(d, i, L);
    if (obs) tab_write_obs(d, L.s, obs, (size_t)i);
}
template <int MODE, bool SAMPLE, bool LDS, int TK>
__global__ __launch_bounds__(kBlock) void tab_rollout_kernel(DevEnv d, RolloutPtrs io, ActionStream as, int T, int lds_bytes) {
    // The transition table is read once per env-step through three levels of dependent loads (count / cumulative probabilities -> branch
    // -> successor, reward, flag); out of L2 that latency is the whole step (Taxi: 100 KB of tables).  When the launcher found that the
    // table fits (LDS) the workgroup first copies it into LDS -- layout: the f64 arrays, then the i32 arrays, then the flags --
    // and every lookup of the T steps is a 64-cycle LDS read instead.
    // LDS is a TEMPLATE parameter (round 6): chosen at run time, the table pointers were generic, the lookups FLAT loads -- 9 per env-step
    // (profiles/r06_frozenlake_rollout.txt: SQ_INSTS_VMEM_RD 1.19e6 per launch against 5e4 LDS instructions) -- and a flat load shares the
    // in-order vmcnt with the trajectory stores: every lookup waited for the stores before it to reach memory.
    extern __shared__ __align__(16) double tab_lds[];
    if constexpr (LDS) {
        const size_t cells = (size_t)d.tab.nS * d.tab.nA, rows = cells * d.tab.K;
        double *cs = tab_lds, *pr = cs + rows, *rw = pr + rows, *isd = rw + rows;
        int32_t *nx = reinterpret_cast<int32_t *>(isd + d.tab.nS), *cnt = nx + rows;
        uint8_t *tm = reinterpret_cast<uint8_t *>(cnt + cells);
        for (size_t k = threadIdx.x; k < rows; k += kBlock)
            cs[k] = d.tab.csprob[k], pr[k] = d.tab.prob[k], rw[k] = d.tab.reward[k], nx[k] = d.tab.next[k], tm[k] = d.tab.term[k];
        for (size_t k = threadIdx.x; k < cells; k += kBlock) cnt[k] = d.tab.count[k];
        for (int k = threadIdx.x; k < d.tab.nS; k += kBlock) isd[k] = d.tab.isd[k];
        __syncthreads();
        d.tab.csprob = cs, d.tab.prob = pr, d.tab.reward = rw, d.tab.isd = isd, d.tab.next = nx, d.tab.count = cnt, d.tab.term = tm;
    }
    const int i = blockIdx.x * kBlock + threadIdx.x;
    LaneStats st = {0u, 0u, 0u, 0ull, 0.0};
    if (i < d.N) {
        TabLane L;
        tab_load<TK>(d, i, L);
        Pcg64 rng = load_rng(d, i);  // the lane's own stream stays in registers for the whole rollout
        u128 astate = 0;
        if (SAMPLE) {
            astate = make_u128(as.state_hi, as.state_lo);
            uint32_t delta = (uint32_t)i + 1u;
            for (int j = 0; delta; j++, delta >>= 1)
                if (delta & 1u) astate = as.pow2[j].mult * astate + as.pow2[j].plus;
        }
        const size_t N = (size_t)d.N;
        // (the policy's draw does not depend on the environment: step t + 1's action is drawn while step t's chain of table lookups is in flight)
        auto draw = [&]() {
            const uint64_t hi = (uint64_t)(astate >> 64), lo = (uint64_t)astate, x = hi ^ lo;
            const unsigned rot = (unsigned)(hi >> 58);
            const uint64_t out = (x >> rot) | (x << ((0u - rot) & 63u));
            astate = as.jump_n.mult * astate + as.jump_n.plus;
            return (int64_t)((double)(out >> 11) * (1.0 / 9007199254740992.0) * (double)d.tab.nA);  // (random(N) * nvec).astype(int64)
        };
        int64_t a_next = 0;
        if (SAMPLE && T > 0) a_next = draw();
        for (int t = 0; t < T; t++) {
            int64_t a;
            if (SAMPLE) {
                a = a_next;
                if (io.actions_out) static_cast<int64_t *>(io.actions_out)[t * N + i] = a;
                a_next = draw();  // (one draw past the last step: the stream's position is kept by the host, not by this state)
            } else {
                a = static_cast<const int64_t *>(io.actions_in)[t * N + i];
            }
            int64_t obs, fin = 0;
            double reward, out_ret;
            int32_t out_len;
            bool te, tr, has_final;
            tab_lane_step<MODE, TK>(d, i, L, a, obs, fin, has_final, reward, te, tr, out_ret, out_len, st, &rng);
            if (io.obs) tab_write_obs<TK>(d, (double)obs, static_cast<int64_t *>(io.obs), t * N + i);
            if (io.reward) io.reward[t * N + i] = reward;
            if (io.terminated) io.terminated[t * N + i] = te;
            if (io.truncated) io.truncated[t * N + i] = tr;
        }
        tab_store<TK>(d, i, L);
        store_rng_state(d, i, rng);
    }
    block_accumulate(d, st);
}

// ---------------------------------------------------------------------------------------------------------
// The collector's rollout of a PLAIN transition table (FrozenLake, CliffWalking, Taxi without the fickle passenger): NEXT_STEP, on-device
// policy, one table for all sub-environments, at most three outcomes per (state, action).  Round 6, after the counters
// (profiles/r06_acrobot_toytext_root_cause.txt: 381 instructions per env-step from a lone wavefront, 39 branches in the loop body, three levels of
// dependent LDS reads): the same values as tab_rollout_kernel<NEXT_STEP, true, true, kTabPlain> from a loop body WITHOUT data-dependent control flow.
//   * A step and an autoreset step both take exactly ONE draw from the sub-environment's generator (utils.py:4-8 categorical_sample in
//     frozen_lake.py:330 and :345), so the draw is unconditional and the two cases are selects on its result, not two copies of the generator.
//   * The table is repacked into LDS as one record per (state, action) -- the cumulative probabilities with -1 behind the row's last outcome
//     ("first k with csprob[k] > u, 0 if none" then needs no count), the rewards, and successor | terminated << 31 -- so ONE level of reads
//     fetches every candidate and the outcome is selected in registers.
//   * The initial-state draw (first state whose cumulative probability exceeds u) starts from a 256-entry guide table -- guide[g] = first
//     state with isd_cs > g / 256, a lower bound of the answer for every u in [g / 256, (g + 1) / 256) because the sums are
//     non-decreasing -- and scans forward: one read for FrozenLake / CliffWalking, ~two for Taxi's 300 start states; under one
//     wavefront-uniform branch that only the steps with an autoreset in the wavefront take.
//   * info["prob"] of the last transition is looked up once, after the loop.
// KL: outcomes per record (1 or 3; a table with two outcomes uses 3).  FULL: all five trajectory arrays are present.
// ONE_START: every episode starts in `start_state` (FrozenLake, CliffWalking: the initial distribution is one state, mi_tabular_load found it) -- the
//   draw is taken and not looked at.  APOW2: the number of actions is a power of two -- (random() * nA).astype(int64) is the top bits of the output.
// ---------------------------------------------------------------------------------------------------------
template <int KL>
struct TabLeanCell;
template <>
struct TabLeanCell<1> {
    static constexpr int BYTES = 16;  // {reward, successor | terminated << 31, -}
};
template <>
struct TabLeanCell<3> {
    static constexpr int BYTES = 64;  // {cs0, cs1 | cs2, reward0 | reward1, reward2 | nt0, nt1, nt2, -}
};
constexpr int kTabGuide = 256;
static inline size_t tab_lean_lds_bytes(int nS, int nA, int KL) {
    return (size_t)nS * nA * (KL == 1 ? TabLeanCell<1>::BYTES : TabLeanCell<3>::BYTES) + (size_t)nS * sizeof(double) + kTabGuide * sizeof(uint32_t);
}
template <int KL, bool FULL, bool ONE_START, bool APOW2>
__global__ __launch_bounds__(kBlock) void tab_rollout_lean_kernel(DevEnv d, RolloutPtrs io, ActionStream as, int T, int start_state, int action_shift) {
    extern __shared__ __align__(16) double tab_lds[];
    constexpr int CELL = TabLeanCell<KL>::BYTES;
    const int nS = d.tab.nS, nA = d.tab.nA, K = d.tab.K;
    const int cells = nS * nA;
    char *const cell_base = reinterpret_cast<char *>(tab_lds);
    double *const isd = reinterpret_cast<double *>(cell_base + (size_t)cells * CELL);
    uint32_t *const guide = reinterpret_cast<uint32_t *>(isd + nS);
    for (int c = threadIdx.x; c < cells; c += kBlock) {
        const int n = d.tab.count[c];
        char *rec = cell_base + (size_t)c * CELL;
        if constexpr (KL == 1) {
            *reinterpret_cast<double *>(rec) = d.tab.reward[(size_t)c * K];
            *reinterpret_cast<uint32_t *>(rec + 8) = (uint32_t)d.tab.next[(size_t)c * K] | (d.tab.term[(size_t)c * K] ? 0x80000000u : 0u);
        } else {
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const bool have = j < K && j < n;
                const size_t r = (size_t)c * K + (j < K ? j : 0);
                reinterpret_cast<double *>(rec)[j] = have ? d.tab.csprob[r] : -1.0;
                reinterpret_cast<double *>(rec)[3 + j] = d.tab.reward[r];
                reinterpret_cast<uint32_t *>(rec + 48)[j] = (uint32_t)d.tab.next[r] | (d.tab.term[r] ? 0x80000000u : 0u);
            }
        }
    }
    for (int k = threadIdx.x; !ONE_START && k < nS; k += kBlock) isd[k] = d.tab.isd[k];
    for (int g = threadIdx.x; !ONE_START && g < kTabGuide; g += kBlock) {  // first state with isd_cs > g / 256 (nS when there is none): bisection in the global array
        const double lim = (double)g * (1.0 / kTabGuide);
        int lo = 0, hi = nS;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (d.tab.isd[mid] > lim)
                hi = mid;
            else
                lo = mid + 1;
        }
        guide[g] = (uint32_t)lo;
    }
    __syncthreads();
    const int i = blockIdx.x * kBlock + threadIdx.x;
    LaneStats st = {0u, 0u, 0u, 0ull, 0.0};
    if (i < d.N) {
        const size_t N = (size_t)d.N;
        int s = (int)d.state[i];
        double prob = d.state[N + i];
        const uint32_t m = d.meta[i];
        uint32_t elapsed = m & kElapsedMask;
        const uint32_t flags0 = m >> kFlagShift;
        uint32_t need_reset = flags0 & kNeedsReset;
        double ep_ret = d.ep_ret[i];
        int32_t ep_len = d.ep_len[i];
        Pcg64 rng = load_rng(d, i);
        u128 astate = make_u128(as.state_hi, as.state_lo);
        {
            uint32_t delta = (uint32_t)i + 1u;
            for (int j = 0; delta; j++, delta >>= 1)
                if (delta & 1u) astate = as.pow2[j].mult * astate + as.pow2[j].plus;
        }
        const double dnA = (double)nA;
        auto draw = [&]() {  // (random(N) * nvec).astype(int64), spaces/multi_discrete.py:176-178
            const uint64_t hi = (uint64_t)(astate >> 64), lo = (uint64_t)astate, x = hi ^ lo;
            const unsigned rot = (unsigned)(hi >> 58);
            const uint64_t out = (x >> rot) | (x << ((0u - rot) & 63u));
            astate = pcg_muladd(astate, as.jump_n.mult, as.jump_n.plus);
            // (nA a power of two: random() * nA is exact, its integer part the output's top bits)
            if constexpr (APOW2) return (int32_t)(out >> action_shift);
            return (int32_t)((double)(out >> 11) * (1.0 / 9007199254740992.0) * dnA);
        };
        int32_t a_next = draw();
        int64_t *out_act = static_cast<int64_t *>(io.actions_out) + i, *out_obs = static_cast<int64_t *>(io.obs) + i;
        double *out_rew = io.reward + i;
        uint8_t *out_te = io.terminated + i, *out_tr = io.truncated + i;
        uint32_t last_rec = 0xffffffffu;  // record offset | outcome of the last step (0xffffffff: it was an autoreset step, or there was none)
#pragma unroll 2
        for (int t = 0; t < T; t++) {
            const int32_t a = a_next;
            a_next = draw();  // (one draw past the last step: the stream's position is kept by the host, not by this state)
            const bool resetting = need_reset != 0;
            // (ONE_START with one outcome per record: nothing reads u -- the generator still advances, its output function is dead code)
            const double u = rng.next_double();
            const uint32_t rec = (uint32_t)(s * nA + a) * (uint32_t)CELL;
            const char *p = cell_base + rec;
            double rew;
            uint32_t nt;
            uint32_t kk = 0;
            if constexpr (KL == 1) {
                const uint4 q = *reinterpret_cast<const uint4 *>(p);
                rew = __hiloint2double((int)q.y, (int)q.x);
                nt = q.z;
            } else {
                double2 q0 = *reinterpret_cast<const double2 *>(p), q1 = *reinterpret_cast<const double2 *>(p + 16),
                              q2 = *reinterpret_cast<const double2 *>(p + 32);
                uint4 q3 = *reinterpret_cast<const uint4 *>(p + 48);
                hold_opaque(q1.y), hold_opaque(q2.x), hold_opaque(q2.y), hold_opaque(q3.x), hold_opaque(q3.y), hold_opaque(q3.z);  // (all four reads are issued together: left alone the compiler sinks two of them behind a branch on c0)
                const bool c0 = q0.x > u, c1 = q0.y > u, c2 = q1.x > u;
                const bool take0 = c0 | !(c1 | c2);  // the first outcome, also when no cumulative sum exceeds u (np.argmax of all-False)
                rew = take0 ? q1.y : (c1 ? q2.x : q2.y);
                nt = take0 ? q3.x : (c1 ? q3.y : q3.z);
                kk = take0 ? 0u : (c1 ? 1u : 2u);
            }
            int rs = start_state;
            if (!ONE_START && resetting) {  // frozen_lake.py:345: categorical_sample(initial_state_distrib); only the steps with an autoreset in the wavefront come here
                int idx = (int)guide[(int)(u * (double)kTabGuide)];
                while (idx < nS && !(isd[idx] > u)) idx++;
                rs = idx < nS ? idx : 0;
            }
            const int s_new = resetting ? rs : (int)(nt & 0x7fffffffu);
            const bool te = !resetting && (nt >> 31) != 0;
            const double reward = resetting ? 0.0 : rew;
            const uint32_t el = elapsed + 1u;
            const bool tr = !resetting && d.max_steps > 0 && (int)el >= d.max_steps;
            const bool done = te || tr;
            const double ret = ep_ret + reward;
            const int32_t len = ep_len + 1;
            st.reset_steps += resetting ? 1u : 0u;
            st.episodes += done ? 1u : 0u;
            st.return_sum += done ? ret : 0.0;
            st.length_sum += done ? (uint64_t)len : 0ull;
            ep_ret = resetting ? 0.0 : ret;
            ep_len = resetting ? 0 : len;
            elapsed = resetting ? 0u : el;
            need_reset = done ? 1u : 0u;
            s = s_new;
            if (t == T - 1) last_rec = resetting ? 0xffffffffu : (rec / (uint32_t)CELL) * 4u + kk;
            if (FULL || io.actions_out) *out_act = (int64_t)a;
            if (FULL || io.obs) *out_obs = (int64_t)s_new;
            if (FULL || io.reward) *out_rew = reward;
            if (FULL || io.terminated) *out_te = te;
            if (FULL || io.truncated) *out_tr = tr;
            out_act += N, out_obs += N, out_rew += N, out_te += N, out_tr += N;
        }
        st.env_steps = (uint32_t)T - st.reset_steps;
        if (T > 0) prob = last_rec == 0xffffffffu ? 1.0 : d.tab.prob[(size_t)(last_rec >> 2) * K + (last_rec & 3u)];
        d.state[i] = (double)s, d.state[N + i] = prob;
        d.meta[i] = (elapsed & kElapsedMask) | (((flags0 & ~kNeedsReset) | need_reset) << kFlagShift);
        d.ep_ret[i] = ep_ret, d.ep_len[i] = ep_len;
        store_rng_state(d, i, rng);
    }
    block_accumulate(d, st);
}

// ---------------------------------------------------------------------------------------------------------
// Blackjack-v1's rollout (NEXT_STEP, on-device policy) without data-dependent control flow -- the values of tab_rollout_kernel<NEXT_STEP, true, false,
// kTabBlackjack>, which spends 784 instructions per env-step in three divergent branches (hit / stick with the dealer's loop / autoreset with its
// five or six draws), each with its own copies of the generator step and of Lemire's rejection loop.
//   * A step consumes a VARIABLE number of 32-bit values (np_random.choice = Lemire's bounded draw on PCG64's buffered halves): every lane produces the
//     next three 64-bit outputs unconditionally -- with the buffered half that is six or seven values, enough for an autoreset (5 - 6), a hit (1) and
//     a dealer who draws up to six cards -- evaluates all six as cards, and only COUNTS how many its case consumed: the generator's new state is
//     one of the four states it already has, the new buffered half one of the three high halves.
//   * hit, stick (the dealer's loop unrolled six times under a running `need` flag) and autoreset are evaluated side by side and selected.
//   * What does not fit -- a rejected draw (9 / 2^32 per card), a dealer who needs a seventh card -- is flagged per lane, and the flagged lanes redo
//     the step from the saved generator state through bj_step / bj_reset, behind one wavefront-uniform branch (`force_slow`: tests send every k-th
//     lane-step there).
// ---------------------------------------------------------------------------------------------------------
MI_DEV int bj_hand(int raw, int ace) { return (ace != 0 && raw + 10 <= 21) ? raw + 10 : raw; }
template <bool FULL>
__global__ __launch_bounds__(kBlock) void bj_rollout_lean_kernel(DevEnv d, RolloutPtrs io, ActionStream as, int T, int force_slow) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    LaneStats st = {0u, 0u, 0u, 0ull, 0.0};
    if (i < d.N) {
        const size_t N = (size_t)d.N;
        const int64_t packed0 = (int64_t)d.state[i];
        int psum = (int)(packed0 & 63), pace = (int)((packed0 >> 6) & 1), ptwo = (int)((packed0 >> 7) & 1);
        int d0 = (int)((packed0 >> 8) & 15), d1 = (int)((packed0 >> 12) & 15);
        const uint64_t aux0 = (uint64_t)d.state[N + i];
        int hh = (aux0 >> 32) ? 1 : 0;  // a buffered 32-bit half is waiting
        uint32_t half = (uint32_t)aux0;
        const uint32_t m0 = d.meta[i];
        uint32_t elapsed = m0 & kElapsedMask;
        const uint32_t flags0 = m0 >> kFlagShift;
        uint32_t need_reset = flags0 & kNeedsReset;
        double ep_ret = d.ep_ret[i];
        int32_t ep_len = d.ep_len[i];
        Pcg64 rng = load_rng(d, i);
        const bool natural = d.P.p[0] != 0.0, sab = d.P.p[1] != 0.0;
        u128 astate = make_u128(as.state_hi, as.state_lo);
        {
            uint32_t delta = (uint32_t)i + 1u;
            for (int j = 0; delta; j++, delta >>= 1)
                if (delta & 1u) astate = as.pow2[j].mult * astate + as.pow2[j].plus;
        }
        auto draw = [&]() {  // Discrete(2): (random() * 2).astype(int64) is the output's top bit
            const int32_t a = (int32_t)(pcg_output(astate) >> 63);
            astate = pcg_muladd(astate, as.jump_n.mult, as.jump_n.plus);
            return a;
        };
        int32_t a_next = draw();
        int64_t *out_act = static_cast<int64_t *>(io.actions_out) + i, *out_obs = static_cast<int64_t *>(io.obs) + (size_t)3 * i;
        double *out_rew = io.reward + i;
        uint8_t *out_te = io.terminated + i, *out_tr = io.truncated + i;
        for (int t = 0; t < T; t++) {
            const int32_t a = a_next;
            a_next = draw();
            const bool resetting = need_reset != 0;
            // the next three outputs and the six values a step can consume
            const u128 s0 = rng.state;
            const u128 s1 = pcg_muladd(s0, pcg_mult(), rng.inc), s2 = pcg_muladd(s1, pcg_mult(), rng.inc), s3 = pcg_muladd(s2, pcg_mult(), rng.inc);
            const uint64_t o1 = pcg_output(s1), o2 = pcg_output(s2), o3 = pcg_output(s3);
            const uint32_t l1 = (uint32_t)o1, h1 = (uint32_t)(o1 >> 32), l2 = (uint32_t)o2, h2 = (uint32_t)(o2 >> 32), l3 = (uint32_t)o3, h3 = (uint32_t)(o3 >> 32);
            uint32_t x[6];
            x[0] = hh ? half : l1, x[1] = hh ? l1 : h1, x[2] = hh ? h1 : l2, x[3] = hh ? l2 : h2, x[4] = hh ? h2 : l3, x[5] = hh ? l3 : h3;
            int c[6];
            bool rej[6];
#pragma unroll
            for (int j = 0; j < 6; j++) {  // bj_card: deck[Lemire(13)], rejected when the product's low word is below 2^32 mod 13 = 9
                const uint32_t k = __umulhi(x[j], 13u);
                c[j] = k < 9u ? (int)k + 1 : 10;
                rej[j] = x[j] * 13u < 9u;
            }
            // autoreset (blackjack.py:181-202): dealer's hand, player's hand, the suit draw (n = 4: never rejected), the face draw when the dealer shows a ten (n = 3: rejects 0)
            const int r_cnt = 5 + (c[0] == 10 ? 1 : 0);
            const bool r_over = rej[0] | rej[1] | rej[2] | rej[3] | (c[0] == 10 && x[5] == 0u);
            const int r_psum = c[2] + c[3], r_pace = (c[2] == 1 || c[3] == 1) ? 1 : 0;
            // hit (:144-151)
            const int h_psum = psum + c[0], h_pace = pace | (c[0] == 1 ? 1 : 0);
            const bool h_te = bj_hand(h_psum, h_pace) > 21;
            // stick (:152-169): the dealer draws to 17
            int dsum = d0 + d1, dace = (d0 == 1 || d1 == 1) ? 1 : 0, dtwo = 1, s_cnt = 0;
            bool s_over = false, alive = true;
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const bool need = alive && bj_hand(dsum, dace) < 17;
                s_cnt += need ? 1 : 0;
                dsum += need ? c[j] : 0;
                dace |= (need && c[j] == 1) ? 1 : 0;
                dtwo = need ? 0 : dtwo;
                s_over |= need && rej[j];
                alive = need;
            }
            s_over |= alive && bj_hand(dsum, dace) < 17;  // a seventh card
            const int ph = bj_hand(psum, pace), dh = bj_hand(dsum, dace);
            const int ps = ph > 21 ? 0 : ph, ds = dh > 21 ? 0 : dh;
            double s_rew = (double)(ps > ds) - (double)(ps < ds);
            const bool pnat = ptwo && pace && psum == 11, dnat = dtwo && dace && dsum == 11;  // sorted(hand) == [1, 10]
            s_rew = (sab && pnat && !dnat) ? 1.0 : ((!sab && natural && pnat && s_rew == 1.0) ? 1.5 : s_rew);
            // the lane's case
            const bool hit = a != 0;
            int cnt = resetting ? r_cnt : (hit ? 1 : s_cnt);
            bool over = resetting ? r_over : (hit ? rej[0] : s_over);
            if (force_slow) over |= (uint32_t)(i + t) % (uint32_t)force_slow == 0u;
            const int o_psum = psum, o_pace = pace, o_ptwo = ptwo, o_d0 = d0, o_d1 = d1;
            double rew = resetting ? 0.0 : (hit ? (h_te ? -1.0 : 0.0) : s_rew);
            bool te0 = !resetting && (hit ? h_te : true);
            psum = resetting ? r_psum : (hit ? h_psum : psum);
            pace = resetting ? r_pace : (hit ? h_pace : pace);
            ptwo = resetting ? 1 : (hit ? 0 : ptwo);
            d0 = resetting ? c[0] : d0, d1 = resetting ? c[1] : d1;
            // what the case consumed: `cnt` values = the buffered half first, then halves of new outputs
            const int m = cnt - hh, calls = (m + 1) >> 1;
            const int o_hh = hh;
            const uint32_t o_half = half;
            half = calls == 0 ? half : (calls == 1 ? h1 : (calls == 2 ? h2 : h3));
            hh = m & 1;
            rng.state = calls == 0 ? s0 : (calls == 1 ? s1 : (calls == 2 ? s2 : s3));
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(over) != 0ull, 0)) {
                if (over) {  // the general routines from the state before the step
                    Pcg64 r = {s0, rng.inc};
                    double aux = o_hh ? (double)((1ull << 32) | (uint64_t)o_half) : 0.0;
                    double sd = (double)(o_psum | (o_pace << 6) | (o_ptwo << 7) | (o_d0 << 8) | (o_d1 << 12));
                    if (resetting) {
                        bj_reset(r, sd, aux);
                        rew = 0.0, te0 = false;
                    } else {
                        bj_step(r, sd, aux, (int64_t)a, natural, sab, rew, te0);
                    }
                    const int64_t q = (int64_t)sd;
                    psum = (int)(q & 63), pace = (int)((q >> 6) & 1), ptwo = (int)((q >> 7) & 1), d0 = (int)((q >> 8) & 15), d1 = (int)((q >> 12) & 15);
                    const uint64_t ax = (uint64_t)aux;
                    hh = (ax >> 32) ? 1 : 0, half = (uint32_t)ax;
                    rng.state = r.state;
                }
            }
            const bool te = te0;
            const uint32_t el = elapsed + 1u;
            const bool tr = !resetting && d.max_steps > 0 && (int)el >= d.max_steps;
            const bool done = te || tr;
            const double ret = ep_ret + rew;
            const int32_t len = ep_len + 1;
            st.reset_steps += resetting ? 1u : 0u;
            st.episodes += done ? 1u : 0u;
            st.return_sum += done ? ret : 0.0;
            st.length_sum += done ? (uint64_t)len : 0ull;
            ep_ret = resetting ? 0.0 : ret;
            ep_len = resetting ? 0 : len;
            elapsed = resetting ? 0u : el;
            need_reset = done ? 1u : 0u;
            if (FULL || io.actions_out) *out_act = (int64_t)a;
            if (FULL || io.obs) {  // _get_obs: (player sum with a usable ace as 11, dealer's first card, usable ace)
                const int usable = (pace != 0 && psum + 10 <= 21) ? 1 : 0;
                out_obs[0] = (int64_t)(usable ? psum + 10 : psum), out_obs[1] = (int64_t)d0, out_obs[2] = (int64_t)usable;
            }
            if (FULL || io.reward) *out_rew = rew;
            if (FULL || io.terminated) *out_te = te;
            if (FULL || io.truncated) *out_tr = tr;
            out_act += N, out_obs += 3 * N, out_rew += N, out_te += N, out_tr += N;
        }
        st.env_steps = (uint32_t)T - st.reset_steps;
        d.state[i] = (double)(psum | (pace << 6) | (ptwo << 7) | (d0 << 8) | (d1 << 12));
        d.state[N + i] = hh ? (double)((1ull << 32) | (uint64_t)half) : 0.0;
        d.meta[i] = (elapsed & kElapsedMask) | (((flags0 & ~kNeedsReset) | need_reset) << kFlagShift);
        d.ep_ret[i] = ep_ret, d.ep_len[i] = ep_len;
        store_rng_state(d, i, rng);
    }
    block_accumulate(d, st);
}

__global__ void seed_words_kernel(DevEnv d, const uint64_t *words, const uint8_t *mask) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.N || (mask && !mask[i])) return;
#pragma unroll
    for (int k = 0; k < 4; k++) d.rng[(size_t)k * d.N + i] = words[(size_t)4 * i + k];
    if (d.tab.nS < 0) d.state[(size_t)d.N + i] = 0.0;  // Blackjack: a fresh Generator has no buffered 32-bit half
    if (d.tab_fickle_rows) d.state[(size_t)2 * d.N + i] = 0.0;  // fickle Taxi: likewise
}

__global__ void seed_sequence_kernel(DevEnv d, uint64_t first_seed, const uint8_t *mask) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.N || (mask && !mask[i])) return;
    uint64_t w[4];
    seed_sequence_words(first_seed + (uint64_t)i, w);
    Pcg64 r;
    r.srandom(w);
    d.rng[i] = (uint64_t)(r.state >> 64), d.rng[(size_t)d.N + i] = (uint64_t)r.state;
    d.rng[(size_t)2 * d.N + i] = (uint64_t)(r.inc >> 64), d.rng[(size_t)3 * d.N + i] = (uint64_t)r.inc;
    if (d.tab.nS < 0) d.state[(size_t)d.N + i] = 0.0;  // Blackjack: a fresh Generator has no buffered 32-bit half
    if (d.tab_fickle_rows) d.state[(size_t)2 * d.N + i] = 0.0;  // fickle Taxi: likewise
}

// ---------------------------------------------------------------------------------------------------------
// The action stream as per-lane states (mi_step with actions == NULL, mi_action_sample): `action_space.sample()` of the BATCHED space is one
// generator drawn in sub-environment order (spaces/multi_discrete.py:176-178 `random(nvec.shape) * nvec`, spaces/box.py:463-465 `uniform(low, high,
// size)` over (N, act_dim) row-major), so lane i owns draws pos + i * D + u of every batch.  act_lane[2][N] holds, per lane, the state whose output IS
// its next draw; a batch later the lane is N * D draws further on (jump).  Kept on the device, the stream needs nothing from the host between
// steps: a captured HIP graph replays sampled steps, and a loop of step(sample()) enqueues no host-to-device traffic.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void act_init_kernel(ActionStream as, uint64_t *act_lane, int N, int D) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    u128 s = make_u128(as.state_hi, as.state_lo);
    uint64_t delta = (uint64_t)i * (uint64_t)D + 1ull;  // skip ahead: one affine map per set bit
    for (int j = 0; delta; j++, delta >>= 1)
        if (delta & 1ull) s = as.pow2[j].mult * s + as.pow2[j].plus;
    act_lane[i] = (uint64_t)(s >> 64), act_lane[(size_t)N + i] = (uint64_t)s;
}
// what a draw becomes: Discrete (n > 0): (u * n).astype(int64); Box: uniform(low[u], high[u]).astype(float32) with the space's float32 bounds
struct ActSpec {
    int D;           // act_dim
    double n;        // number of actions of a Discrete sub-space, 0 for Box
    double lo[24], range[24];
};
// T batches: out[t][i][u]; one launch (a host class hands them out one batch per sample() call)
__global__ __launch_bounds__(kBlock) void act_sample_kernel(ActSpec sp, ActionStream as, uint64_t *act_lane, void *out, int T, int N) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const u128 inc = make_u128(as.inc_hi, as.inc_lo);
    u128 s = make_u128(act_lane[i], act_lane[(size_t)N + i]);
    for (int t = 0; t < T; t++) {
        const size_t row = ((size_t)t * N + i) * sp.D;
        for (int u = 0; u < sp.D; u++) {
            const double x = (double)(pcg_output(s) >> 11) * (1.0 / 9007199254740992.0);
            if (sp.n > 0)
                static_cast<int64_t *>(out)[row + u] = (int64_t)(x * sp.n);
            else
                static_cast<float *>(out)[row + u] = (float)(sp.lo[u] + sp.range[u] * x);
            if (u + 1 < sp.D) s = s * pcg_mult() + inc;
        }
        s = as.jump_n.mult * s + as.jump_n.plus;
    }
    act_lane[i] = (uint64_t)(s >> 64), act_lane[(size_t)N + i] = (uint64_t)s;
}

#endif  // !MI_CLASSIC_TU

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------

}  // namespace
// shared with the other translation units of the library (hidden visibility): sets mi_last_error(), returns `code`
namespace mi_internal {
#ifndef MI_CLASSIC_TU
thread_local char g_err[512] = "";
int set_error(int code, const char *msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}
#else
int set_error(int code, const char *msg);
#endif
}  // namespace mi_internal
namespace {

int fail(int code, const char *fmt, const char *detail = "") {
    char msg[512];
    snprintf(msg, sizeof msg, fmt, detail);
    return mi_internal::set_error(code, msg);
}

#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            char msg_[512];                                                                               \
            snprintf(msg_, sizeof msg_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return mi_internal::set_error((int)MI_ERR_HIP, msg_);                                          \
        }                                                                                                 \
    } while (0)

constexpr int kClassicKinds = 5;
const mi_layout kLayouts[kClassicKinds] = {
    {CartPole::OBS, MI_F32, 1, MI_I64, CartPole::S, 0, {0, 0}},
    {Pendulum::OBS, MI_F32, 1, MI_F32, Pendulum::S, 0, {0, 0}},
    {Acrobot::OBS, MI_F32, 1, MI_I64, Acrobot::S, 0, {0, 0}},
    {MountainCar::OBS, MI_F32, 1, MI_I64, MountainCar::S, 0, {0, 0}},
    {MountainCarContinuous::OBS, MI_F32, 1, MI_F32, MountainCarContinuous::S, 0, {0, 0}},
};
const int kNumActions[MI_ENV_KIND_COUNT] = {2, 0, 3, 3, 0, 0, 0, 0, 0};

}  // namespace

struct mi_vecenv {
    mi_config cfg;
    mi_layout lay;
    int device;
    hipStream_t stream, own_stream;
    DevEnv d;
    int grid;
    bool seeded, was_reset, act_seeded;
    bool has_epi;           // mi_set_step_epilogue: the wrappers run as the output stage of step_kernel
    EpiDev epi;
    mi_step_epilogue epi_host;  // what the caller attached: the statistics handles swap their two buffer sets every step, so epi is rebuilt
    double *d_epi_partial;      // [grid][kEpiCols] column sums of the step workgroups
    Pcg64 act_rng;          // host copy of the action-space generator
    PcgJump *d_pow2;        // [64] device jump table for act_rng.inc
    PcgJump jump_n;
    // the action stream as per-lane states on the device (act_sample_kernel): act_lane_valid = they correspond to the stream's position;
    // act_on_device = they ARE the position (the host copy act_rng is stale until act_sync_host reads lane 0 back)
    uint64_t *d_act_lane;   // [2][N]
    bool act_lane_valid, act_on_device;
    void *d_act_stage;      // mi_action_sample(MI_HOST): where the kernel writes before the copy to the caller's array
    size_t act_stage_bytes;
    // Staging for the MI_HOST entry points (SURVEY 8(b) "Ownership"): every step output lives in ONE device block and ONE pinned host
    // block of the same layout -- [error word | obs | reward | terminated | truncated | info | episode_return | episode_length | final_obs |
    // final_info], 256-byte aligned sections -- so a host step is one H2D (actions), one launch and one D2H of the prefix in use.
    // d_obs ... d_final_info / h_* are views into the two blocks.
    char *d_out, *h_out, *h_out_dev;  // h_out_dev: the device address of the pinned block (zero-copy step of the classic kinds)
    int *h_err;                       // sticky device error word, page-locked host memory
    size_t out_bytes, off_end[9];  // end offset of each section in the order above (after the header)
    void *d_actions, *h_actions;
    void *d_obs, *d_final;
    double *d_reward, *d_epret;
    uint8_t *d_term, *d_trunc, *d_mask;
    int32_t *d_eplen;
    uint64_t *d_words;
    size_t act_bytes, act_bytes_max, obs_bytes;
    double *d_info, *d_final_info;
    size_t info_bytes;
    mi_step_io h_io;        // the pinned host arrays (mi_host_buffers)
    mi_step_io pending;     // user pointers of a step that was enqueued by mi_step_async and not waited for yet
    size_t pending_bytes;
    bool has_pending;
    void *tab_bufs[8];
    bool tab_loaded;
    int tab_start_state;  // the one state every episode starts in (its cumulative initial probability is >= 1), -1 when the start is drawn among several
    // MuJoCo family: cooperative physics kernel (default) or the one-lane simulator (MI355ENV_MJ_SERIAL=1, cross-check)
    bool mj_coop;
    int extras_dim;
    double *d_extras;       // [N][EX_TOTAL]
    float *d_act_scratch;   // [N][NU] actions of the current rollout step when the caller does not keep them
    void *d_obs_scratch;    // [N][obs_dim] observations of the current rollout step when the caller does not keep them
    // MI_CFG_SHARED_RNG (CartPoleVectorEnv's semantics): the one generator and the bookkeeping of its draw positions, all device memory
    bool shared_rng;
    SharedRng shared;
    PcgJump *d_shared_pow2;
    Pcg64 shared_base;      // host copy of the base generator (mi_get_rng: base advanced by the device's `consumed` counter)
};

namespace {

template <class M>
struct AcrobotMath {
    typedef M type;
};
template <>
struct AcrobotMath<ExactMath> {
    typedef ExactMathBuiltinFma type;
};
template <class M, class F>
int dispatch_kind_math(int kind, F &&f) {
    switch (kind) {
    case MI_ENV_CARTPOLE: return f(CartPoleT<M>());
    case MI_ENV_PENDULUM: return f(PendulumT<M>());
    case MI_ENV_ACROBOT: return f(AcrobotT<typename AcrobotMath<M>::type>());  // (exact math: without the inline-asm Horner steps, envs_classic.h)
    case MI_ENV_MOUNTAIN_CAR: return f(MountainCarT<M>());
    case MI_ENV_MOUNTAIN_CAR_CONTINUOUS: return f(MountainCarContinuousT<M>());
    }
    return fail(MI_ERR_INVALID_ARGUMENT, "unknown env kind");
}
// fast_math: MI_CFG_FAST_MATH (envs_classic.h FastMath); the default is the reference's libm bit for bit
template <class F>
int dispatch_kind(int kind, bool fast_math, F &&f) {
    return fast_math ? dispatch_kind_math<FastMath>(kind, f) : dispatch_kind_math<ExactMath>(kind, f);
}
// ... and the kind of the action rows (mi_step_io.actions_dtype): the two Box kinds have float64 instantiations (envs_classic.h ActF64 / ActF64Weak)
template <class M, class F>
int dispatch_kind_act_math(int kind, int act_kind, F &&f) {
    if (act_kind != MI_F32) {
        if (kind == MI_ENV_PENDULUM) return f(PendulumT<M, ActF64>());  // (a Python float and an np.float64 behave alike there)
        if (kind == MI_ENV_MOUNTAIN_CAR_CONTINUOUS) return act_kind == MI_F64_WEAK ? f(MountainCarContinuousT<M, ActF64Weak>()) : f(MountainCarContinuousT<M, ActF64>());
    }
    return dispatch_kind_math<M>(kind, f);
}
template <class F>
int dispatch_kind_act(int kind, bool fast_math, int act_kind, F &&f) {
    return fast_math ? dispatch_kind_act_math<FastMath>(kind, act_kind, f) : dispatch_kind_act_math<ExactMath>(kind, act_kind, f);
}

#ifndef MI_CLASSIC_TU
typedef mjx::MjEnv<mjx::HalfCheetahModel, mjx::kHalfCheetah> HalfCheetahEnv;
typedef mjx::MjEnv<mjx::AntModel, mjx::kAnt> AntEnv;
typedef mjx::MjEnv<mjx::HumanoidModel, mjx::kHumanoid> HumanoidEnv;
typedef mjx::MjEnv<mjx::HopperModel, mjx::kHopper> HopperEnv;
typedef mjx::MjEnv<mjx::Walker2dModel, mjx::kWalker2d> Walker2dEnv;
typedef mjx::MjEnv<mjx::InvertedPendulumModel, mjx::kInvertedPendulum> InvertedPendulumEnv;
typedef mjx::MjEnv<mjx::InvertedDoublePendulumModel, mjx::kInvertedDoublePendulum> InvertedDoublePendulumEnv;
typedef mjx::MjEnv<mjx::ReacherModel, mjx::kReacher> ReacherEnv;
typedef mjx::MjEnv<mjx::HumanoidStandupModel, mjx::kHumanoidStandup> HumanoidStandupEnv;
typedef mjx::MjEnv<mjx::SwimmerModel, mjx::kSwimmer> SwimmerEnv;
typedef mjx::MjEnv<mjx::PusherModel, mjx::kPusher> PusherEnv;
#endif
bool is_mj(int kind) { return (kind >= kClassicKinds && kind <= MI_ENV_HUMANOID) || (kind >= MI_ENV_HOPPER && kind <= MI_ENV_INVERTED_DOUBLE_PENDULUM) || kind == MI_ENV_REACHER || kind == MI_ENV_HUMANOID_STANDUP || kind == MI_ENV_SWIMMER || kind == MI_ENV_PUSHER; }
bool is_tab(int kind) { return kind == MI_ENV_TABULAR || kind == MI_ENV_BLACKJACK; }  // Blackjack rides on the tabular kernels
#ifndef MI_CLASSIC_TU
template <class F>
int dispatch_mj(int kind, F &&f) {
    switch (kind) {
    case MI_ENV_HALF_CHEETAH: return f(HalfCheetahEnv());
    case MI_ENV_ANT: return f(AntEnv());
    case MI_ENV_HUMANOID: return f(HumanoidEnv());
    case MI_ENV_HOPPER: return f(HopperEnv());
    case MI_ENV_WALKER2D: return f(Walker2dEnv());
    case MI_ENV_INVERTED_PENDULUM: return f(InvertedPendulumEnv());
    case MI_ENV_INVERTED_DOUBLE_PENDULUM: return f(InvertedDoublePendulumEnv());
    case MI_ENV_REACHER: return f(ReacherEnv());
    case MI_ENV_HUMANOID_STANDUP: return f(HumanoidStandupEnv());
    case MI_ENV_SWIMMER: return f(SwimmerEnv());
    case MI_ENV_PUSHER: return f(PusherEnv());
    }
    return fail(MI_ERR_UNSUPPORTED, "this MuJoCo kind is not built into the HIP engine yet");
}

// The sticky device error word lives in page-locked host memory (the kernels write it through its device address on the rare error): after
// a stream synchronisation it is simply read, no copy.
int raise_device_error(mi_vecenv *v) {  // precondition: the stream is idle (a kernel in flight could set the word again right after the clear)
    const int err = *v->h_err;
    if (!err) return MI_OK;
    *v->h_err = 0;
    if (v->shared_rng && v->shared.words) {  // the shared-generator mode refuses every step after a bad batch until it has been raised: lift that
        (void)hipMemsetAsync(v->shared.words + 7, 0, sizeof(uint64_t), v->stream);
        (void)hipStreamSynchronize(v->stream);
    }
    if (err == kErrInvalidAction) return fail(MI_ERR_INVALID_ARGUMENT, "action outside the action space");
    return fail(MI_ERR_STATE, "DISABLED autoreset: a finished sub-environment was stepped without reset");
}
int check_device_error(mi_vecenv *v) {
    HIP_TRY(hipStreamSynchronize(v->stream));
    return raise_device_error(v);
}

#endif  // !MI_CLASSIC_TU

// the statistics handles' current buffer sets into the device view (they swap after every step that updated them)
void epilogue_bind(mi_vecenv *v) {
    EpiDev &d = v->epi;
    if (const mi_running_stats *o = v->epi_host.obs_rms)
        d.obs_mean = o->mean, d.obs_var = o->var, d.obs_count = o->count, d.obs_mean2 = o->mean2, d.obs_var2 = o->var2, d.obs_count2 = o->count2;
    if (const mi_running_stats *r = v->epi_host.return_rms)
        d.ret_mean = r->mean, d.ret_var = r->var, d.ret_count = r->count, d.ret_mean2 = r->mean2, d.ret_var2 = r->var2, d.ret_count2 = r->count2;
}
void epilogue_swap(mi_vecenv *v) {
    auto swap = [](mi_running_stats *s) {
        double *t;
        t = s->mean, s->mean = s->mean2, s->mean2 = t;
        t = s->var, s->var = s->var2, s->var2 = t;
        t = s->count, s->count = s->count2, s->count2 = t;
    };
    if (v->epi.obs_on && v->epi.obs_update) swap(v->epi_host.obs_rms);
    if (v->epi.ret_on && v->epi.ret_update) swap(v->epi_host.return_rms);
}

template <class E, bool EPI, bool SAMPLE = false>
void launch_step_mode(mi_vecenv *v, const StepPtrs &p) {
    const dim3 g(v->grid), b(kBlock);
    switch (v->cfg.autoreset_mode) {
    case MI_AUTORESET_NEXT_STEP: hipLaunchKernelGGL((step_kernel<E, MI_AUTORESET_NEXT_STEP, EPI, SAMPLE>), g, b, 0, v->stream, v->d, p, v->epi); break;
    case MI_AUTORESET_SAME_STEP: hipLaunchKernelGGL((step_kernel<E, MI_AUTORESET_SAME_STEP, EPI, SAMPLE>), g, b, 0, v->stream, v->d, p, v->epi); break;
    default: hipLaunchKernelGGL((step_kernel<E, MI_AUTORESET_DISABLED, EPI, SAMPLE>), g, b, 0, v->stream, v->d, p, v->epi); break;
    }
}
template <class E>
int launch_step(mi_vecenv *v, const StepPtrs &p) {
    if (!v->has_epi) {
        // the on-device policy draws the space's own dtype (float32 rows / int64): the float64-row instantiations never sample (the caller, step_enqueue,
        // sends a wrapped step that samples through the stand-alone sampler + this kernel with p.actions set instead)
        if constexpr (E::ACT_KIND == MI_F32 || E::ACT_KIND == MI_I64) {
            if (p.act_lane)
                launch_step_mode<E, false, true>(v, p);
            else
                launch_step_mode<E, false>(v, p);
        } else {
            launch_step_mode<E, false>(v, p);
        }
    } else {
        if (!p.obs || !p.reward || !p.terminated || !p.truncated) return fail(MI_ERR_INVALID_ARGUMENT, "a step with an epilogue needs obs, reward, terminated and truncated");
        epilogue_bind(v);
        launch_step_mode<E, true>(v, p);
        const EpiDev &e = v->epi;
        if (e.obs_on || e.ret_on) {
            if (!e.fold_in_finish && epi_reduces(e)) hipLaunchKernelGGL((epilogue_combine<E::OBS>), dim3(1), dim3(kBlock), 0, v->stream, e, v->d.N);
            hipLaunchKernelGGL((epilogue_finish<E::OBS>), dim3(v->grid), dim3(kBlock), 0, v->stream, e, v->d.N, p.obs, p.reward, p.terminated, p.truncated);
            epilogue_swap(v);
        }
    }
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

template <class E, int MODE, bool SAMPLE, bool FULL>
void launch_rollout_variant(mi_vecenv *v, const RolloutPtrs &p, const ActionStream &as, int T) {
    hipLaunchKernelGGL((rollout_kernel<E, MODE, SAMPLE, FULL>), dim3(v->grid), dim3(kBlock), 0, v->stream, v->d, p, as, T);
}

template <class E>
int launch_rollout(mi_vecenv *v, const RolloutPtrs &p, const ActionStream &as, int T, bool sample) {
    const bool next_step = v->cfg.autoreset_mode == MI_AUTORESET_NEXT_STEP;
    const bool full = p.obs && p.reward && p.terminated && p.truncated && (!sample || p.actions_out);
    if constexpr (E::ACT_KIND == MI_F64 || E::ACT_KIND == MI_F64_WEAK) {  // float64 action rows are always the caller's (the sampler draws float32)
        if (sample) return fail(MI_ERR_INVALID_ARGUMENT, "the on-device policy samples float32 actions");
        if (next_step)
            launch_rollout_variant<E, MI_AUTORESET_NEXT_STEP, false, false>(v, p, as, T);
        else
            launch_rollout_variant<E, MI_AUTORESET_SAME_STEP, false, false>(v, p, as, T);
    } else if (next_step && sample && full) {  // the collector's configuration (bench.py)
        bool duo = false;
        if constexpr (E::DUO_ROLLOUT) {
            const char *env = getenv("MI355ENV_ROLLOUT_DUO");  // "0": the one-role kernel (A/B: scripts/r04/duo_ab_bench.sh)
            duo = !(env && env[0] == '0') && T % DuoTraits<E>::CHUNK == 0;
            if (duo) hipLaunchKernelGGL((rollout_duo_kernel<E>), dim3(v->grid), dim3(kDuoBlock), 0, v->stream, v->d, p, as, T);
        }
        if (!duo) launch_rollout_variant<E, MI_AUTORESET_NEXT_STEP, true, true>(v, p, as, T);
    }
    else if (next_step && sample)
        launch_rollout_variant<E, MI_AUTORESET_NEXT_STEP, true, false>(v, p, as, T);
    else if (next_step)
        launch_rollout_variant<E, MI_AUTORESET_NEXT_STEP, false, false>(v, p, as, T);
    else if (sample)
        launch_rollout_variant<E, MI_AUTORESET_SAME_STEP, true, false>(v, p, as, T);
    else
        launch_rollout_variant<E, MI_AUTORESET_SAME_STEP, false, false>(v, p, as, T);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

}  // namespace

#ifdef MI_CLASSIC_TU
// ---- the launchers the other translation unit calls (this unit is compiled with the max-ILP scheduler, see the head of the file) ----------------
namespace mi_classic {
int step(mi_vecenv *v, const StepPtrs &p, int act_kind) {
    return dispatch_kind_act(v->cfg.kind, (v->cfg.reserved[0] & MI_CFG_FAST_MATH) != 0, act_kind, [&](auto env) -> int { return launch_step<decltype(env)>(v, p); });
}
int reset(mi_vecenv *v, const uint8_t *dm, int has_bounds, double b0, double b1, float *dobs) {
    return dispatch_kind(v->cfg.kind, (v->cfg.reserved[0] & MI_CFG_FAST_MATH) != 0, [&](auto env) -> int {
        using E = decltype(env);
        hipLaunchKernelGGL((reset_kernel<E>), dim3(v->grid), dim3(kBlock), 0, v->stream, v->d, dm, has_bounds, b0, b1, dobs);
        HIP_TRY(hipGetLastError());
        return (int)MI_OK;
    });
}
int rollout(mi_vecenv *v, const RolloutPtrs &p, const ActionStream &as, int T, bool sample, int in_kind) {
    return dispatch_kind_act(v->cfg.kind, (v->cfg.reserved[0] & MI_CFG_FAST_MATH) != 0, in_kind, [&](auto env) -> int { return launch_rollout<decltype(env)>(v, p, as, T, sample); });
}
// ---- MI_CFG_SHARED_RNG: MI_ENV_CARTPOLE only (mi_create refuses the bit for every other kind) ---------------------------------------------
template <class F>
static int dispatch_shared(mi_vecenv *v, F &&f) {
    if (v->cfg.reserved[0] & MI_CFG_FAST_MATH) return f(CartPoleT<FastMath>());
    return f(CartPoleT<ExactMath>());
}
int shared_reset(mi_vecenv *v, float *dobs) {
    hipLaunchKernelGGL(shared_scan_kernel, dim3(1), dim3(kBlock), 0, v->stream, v->shared, v->grid, (uint64_t)4 * (uint64_t)v->cfg.num_envs);
    return dispatch_shared(v, [&](auto env) -> int {
        hipLaunchKernelGGL((shared_reset_kernel<decltype(env)>), dim3(v->grid), dim3(kBlock), 0, v->stream, v->d, v->shared, dobs);
        HIP_TRY(hipGetLastError());
        return (int)MI_OK;
    });
}
int shared_step(mi_vecenv *v, const StepPtrs &p) {
    dispatch_shared(v, [&](auto env) -> int {
        using E = decltype(env);
        hipLaunchKernelGGL((shared_validate_kernel<E>), dim3(v->grid), dim3(kBlock), 0, v->stream, v->d, static_cast<const typename E::Act *>(p.actions), v->shared);
        return (int)MI_OK;
    });
    hipLaunchKernelGGL(shared_scan_kernel, dim3(1), dim3(kBlock), 0, v->stream, v->shared, v->grid, (uint64_t)0);
    return dispatch_shared(v, [&](auto env) -> int {
        hipLaunchKernelGGL((shared_step_kernel<decltype(env)>), dim3(v->grid), dim3(kBlock), 0, v->stream, v->d, p, v->shared);
        HIP_TRY(hipGetLastError());
        return (int)MI_OK;
    });
}
int shared_rollout(mi_vecenv *v, const RolloutPtrs &p, const ActionStream &as, int T, bool sample) {
    const size_t N = (size_t)v->cfg.num_envs;
    for (int t = 0; t < T; t++) {
        StepPtrs sp;
        memset(&sp, 0, sizeof sp);
        if (sample) {
            int64_t *dst = p.actions_out ? static_cast<int64_t *>(p.actions_out) + (size_t)t * N : static_cast<int64_t *>(v->d_actions);
            const int rc = dispatch_shared(v, [&](auto env) -> int {
                hipLaunchKernelGGL((shared_sample_kernel<decltype(env)>), dim3(v->grid), dim3(kBlock), 0, v->stream, v->d, as, t, dst);
                return (int)MI_OK;
            });
            if (rc) return rc;
            sp.actions = dst;
        } else {
            sp.actions = static_cast<const int64_t *>(p.actions_in) + (size_t)t * N;
        }
        sp.obs = p.obs ? static_cast<float *>(p.obs) + (size_t)t * N * CartPole::OBS : nullptr;
        sp.reward = p.reward ? p.reward + (size_t)t * N : nullptr;
        sp.terminated = p.terminated ? p.terminated + (size_t)t * N : nullptr;
        sp.truncated = p.truncated ? p.truncated + (size_t)t * N : nullptr;
        if (const int rc = shared_step(v, sp)) return rc;
    }
    return MI_OK;
}
int shared_recount(mi_vecenv *v) {
    hipLaunchKernelGGL(shared_count_kernel, dim3(v->grid), dim3(kBlock), 0, v->stream, v->d, v->shared);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}
}  // namespace mi_classic
#else
namespace {

// One vector step of a MuJoCo env: [cooperative physics ->] per-lane glue (autoreset state machine, reward, observation, stats)
template <class E>
int launch_mj_step(mi_vecenv *v, MjStepPtrs mp) {
    const dim3 g(v->grid), b(kBlock);
    const int mode = v->cfg.autoreset_mode;
    mp.extras = nullptr;
    if constexpr (!E::HAS_COOP) {
        switch (mode) {
        case MI_AUTORESET_NEXT_STEP: hipLaunchKernelGGL((mj_step_kernel<E, MI_AUTORESET_NEXT_STEP, false>), g, b, 0, v->stream, v->d, mp); break;
        case MI_AUTORESET_SAME_STEP: hipLaunchKernelGGL((mj_step_kernel<E, MI_AUTORESET_SAME_STEP, false>), g, b, 0, v->stream, v->d, mp); break;
        default: hipLaunchKernelGGL((mj_step_kernel<E, MI_AUTORESET_DISABLED, false>), g, b, 0, v->stream, v->d, mp); break;
        }
        HIP_TRY(hipGetLastError());
        return MI_OK;
    } else {
    if (v->mj_coop) {
        mi_phys::Args pa;
        pa.state = v->d.state, pa.meta = v->d.meta, pa.needs_reset_mask = kNeedsReset << kFlagShift, pa.N = v->d.N, pa.frame_skip = (int)v->d.P.p[4];
        pa.newton = v->d.solver_newton, pa.act_f64 = mp.act_f64;
        const bool skip_resetting = mode != MI_AUTORESET_SAME_STEP;  // SAME_STEP: every sub-environment steps
        const bool ok = E::COOP_G == 16 ? mi_phys::launch16(v->cfg.kind, pa, skip_resetting, mp.actions, v->d_extras, v->stream)
                                        : mi_phys::launch32(v->cfg.kind, pa, skip_resetting, mp.actions, v->d_extras, v->stream);
        if (!ok) return fail(MI_ERR_UNSUPPORTED, "no cooperative physics kernel for this env kind");
        mp.extras = v->d_extras;
    }
    if (v->mj_coop) {
        switch (mode) {
        case MI_AUTORESET_NEXT_STEP: hipLaunchKernelGGL((mj_step_kernel<E, MI_AUTORESET_NEXT_STEP, true>), g, b, 0, v->stream, v->d, mp); break;
        case MI_AUTORESET_SAME_STEP: hipLaunchKernelGGL((mj_step_kernel<E, MI_AUTORESET_SAME_STEP, true>), g, b, 0, v->stream, v->d, mp); break;
        default: hipLaunchKernelGGL((mj_step_kernel<E, MI_AUTORESET_DISABLED, true>), g, b, 0, v->stream, v->d, mp); break;
        }
    } else {
        switch (mode) {
        case MI_AUTORESET_NEXT_STEP: hipLaunchKernelGGL((mj_step_kernel<E, MI_AUTORESET_NEXT_STEP, false>), g, b, 0, v->stream, v->d, mp); break;
        case MI_AUTORESET_SAME_STEP: hipLaunchKernelGGL((mj_step_kernel<E, MI_AUTORESET_SAME_STEP, false>), g, b, 0, v->stream, v->d, mp); break;
        default: hipLaunchKernelGGL((mj_step_kernel<E, MI_AUTORESET_DISABLED, false>), g, b, 0, v->stream, v->d, mp); break;
        }
    }
    HIP_TRY(hipGetLastError());
    return MI_OK;
    }
}

// T vector steps with the cooperative physics: per step [sample actions ->] physics -> glue, all on the env's stream
template <class E>
int launch_mj_rollout_coop(mi_vecenv *v, const RolloutPtrs &p, const ActionStream &as, int T, bool sample, int in_f64) {
    const size_t N = (size_t)v->cfg.num_envs;
    const dim3 g(v->grid), b(kBlock);
    for (int t = 0; t < T; t++) {
        const void *act;
        if (sample) {
            float *dst = p.actions_out ? static_cast<float *>(p.actions_out) + (size_t)t * N * E::NU : v->d_act_scratch;
            hipLaunchKernelGGL((mj_sample_kernel<E>), g, b, 0, v->stream, v->d, as, t, dst);
            act = dst;
        } else if (in_f64) {
            act = static_cast<const double *>(p.actions_in) + (size_t)t * N * E::NU;
        } else {
            act = static_cast<const float *>(p.actions_in) + (size_t)t * N * E::NU;
        }
        MjStepPtrs mp;
        memset(&mp, 0, sizeof mp);
        mp.actions = act, mp.act_f64 = !sample && in_f64;
        mp.obs = p.obs ? static_cast<double *>(p.obs) + (size_t)t * N * v->lay.obs_dim : static_cast<double *>(v->d_obs_scratch);
        mp.reward = p.reward ? p.reward + (size_t)t * N : nullptr;
        mp.terminated = p.terminated ? p.terminated + (size_t)t * N : nullptr;
        mp.truncated = p.truncated ? p.truncated + (size_t)t * N : nullptr;
        mp.obs_dim = v->lay.obs_dim;
        const int rc = launch_mj_step<E>(v, mp);
        if (rc) return rc;
    }
    return MI_OK;
}

int set_device(const mi_vecenv *v) {
    HIP_TRY(hipSetDevice(v->device));
    return MI_OK;
}

}  // namespace

#pragma GCC visibility push(default)
extern "C" {

int mi_abi_version(void) { return MI355ENV_ABI_VERSION; }
const char *mi_last_error(void) { return mi_internal::g_err; }

int mi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

static int create_buffers(mi_vecenv *v, const mi_config *cfg, int device);

int mi_create(const mi_config *cfg, int device, mi_vecenv **out) {
    if (!cfg || !out || cfg->struct_size != (int32_t)sizeof(mi_config)) return fail(MI_ERR_INVALID_ARGUMENT, "bad mi_config");
    if (cfg->kind < 0 || cfg->kind >= MI_ENV_KIND_COUNT) return fail(MI_ERR_INVALID_ARGUMENT, "unknown env kind");
    if (cfg->num_envs < 1) return fail(MI_ERR_INVALID_ARGUMENT, "num_envs must be >= 1");
    if (cfg->autoreset_mode < 0 || cfg->autoreset_mode > 2) return fail(MI_ERR_INVALID_ARGUMENT, "bad autoreset mode");
    if (cfg->max_episode_steps > (int)kElapsedMask) return fail(MI_ERR_INVALID_ARGUMENT, "max_episode_steps too large");
    if ((cfg->reserved[0] & MI_CFG_SHARED_RNG) && (cfg->kind != MI_ENV_CARTPOLE || cfg->autoreset_mode != MI_AUTORESET_NEXT_STEP))
        return fail(MI_ERR_UNSUPPORTED, "MI_CFG_SHARED_RNG is CartPoleVectorEnv's semantics: MI_ENV_CARTPOLE with NEXT_STEP autoreset only");
    const int ndev = mi_device_count();
    if (ndev == 0)
        return fail(MI_ERR_NO_DEVICE, "no HIP device visible: libmi355env runs on MI355X (gfx950) only and has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(MI_ERR_INVALID_ARGUMENT, "device index out of range");
    mi_vecenv *v = new (std::nothrow) mi_vecenv();
    if (!v) return fail(MI_ERR_HIP, "out of host memory");
    memset(v, 0, sizeof *v);
    v->cfg = *cfg, v->device = device;
    const int rc = create_buffers(v, cfg, device);
    if (rc != MI_OK) {  // a failing allocation half way: release what the earlier ones got (mi_destroy skips the null ones) and keep the message
        const std::string msg = mi_last_error();
        mi_destroy(v);
        return fail(rc, "%s", msg.c_str());
    }
    *out = v;
    return MI_OK;
}

static int create_buffers(mi_vecenv *v, const mi_config *cfg, int device) {
    if (is_mj(cfg->kind)) {
        mi::EnvParams P;
        for (int k = 0; k < 16; k++) P.p[k] = cfg->params[k];
        dispatch_mj(cfg->kind, [&](auto env) -> int {
            using E = decltype(env);
            const mi_layout l = {E::obs_dim_host(P), MI_F64, E::NU, MI_F32, E::S, E::INFO, {0, 0}};
            v->lay = l;
            v->extras_dim = mjx::coop::Sim<typename E::Model, E::COOP_G>::EX_TOTAL;
            return (int)MI_OK;
        });
        // Which physics kernel: the cooperative one (mjx_coop.h) wins where the robot fills its 16 / 32 lanes (Ant 2.4M vs 1.4M
        // env-steps/s, Humanoid 0.76M vs 0.23M); HalfCheetah (9 dofs, 7 bodies, one forward pass per sub-step) is faster on the
        // one-lane kernel (16.4M vs 12.9M).  MI355ENV_MJ_SERIAL=1 / MI355ENV_MJ_COOP=1 force either one (cross-check tests).
        const char *serial = getenv("MI355ENV_MJ_SERIAL"), *coop = getenv("MI355ENV_MJ_COOP");
        // the faster kernel per robot: HalfCheetah 22.0 M (cooperative) vs 18.8 M (one-lane) env-steps/s at 65536 envs
        const bool has_coop = cfg->kind == MI_ENV_HALF_CHEETAH || cfg->kind == MI_ENV_ANT || cfg->kind == MI_ENV_HUMANOID || cfg->kind == MI_ENV_HUMANOID_STANDUP ||
                              cfg->kind == MI_ENV_HOPPER || cfg->kind == MI_ENV_WALKER2D;  // (round 2: the planar walkers joined -- Walker2d 6.6 M one-lane)
        v->mj_coop = has_coop && cfg->kind != MI_ENV_HOPPER;  // measured @65536: Walker2d 8.4 M cooperative vs 6.6 M one-lane; Hopper 9.7 M vs 17.2 M
        if (serial && serial[0] == '1') v->mj_coop = false;
        if (coop && coop[0] == '1') v->mj_coop = true;
        if (!has_coop) v->mj_coop = false;  // the other small robots: one-lane kernel only (cylinder geoms, fluid forces, two-dof arms)
    } else if (cfg->kind == MI_ENV_BLACKJACK) {
        const mi_layout l = {3, MI_I64, 1, MI_I64, 2, 0, {0, 0}};
        v->lay = l;
        v->d.tab.nS = -1, v->d.tab.nA = 2, v->d.tab.K = 0;  // the marker the kernels branch on (is_blackjack)
        v->tab_loaded = true;
    } else if (is_tab(cfg->kind)) {
        const mi_layout l = {1, MI_I64, 1, MI_I64, cfg->params[2] != 0.0 ? 3 : 2, 1, {0, 0}};  // (the fickle Taxi keeps one more word per sub-env)
        v->lay = l;
    } else {
        v->lay = kLayouts[cfg->kind];
    }
    HIP_TRY(hipSetDevice(device));
    // One engine stream per DEVICE, shared by every env of the process and never destroyed.  A stream is a hardware queue with its own
    // scratch-memory reservation, sized for the hungriest kernel launched on it (the one-lane MuJoCo kernels keep ~20 KB per lane there):
    // a stream per env made a long-lived process that creates and closes many envs (the GPU test-suite: > 100) abort inside the runtime
    // at the first large-scratch launch of some later env.  Sub-environments of different envs are independent, so sharing only serialises
    // launches that a caller wanting overlap can still separate with mi_set_stream.
    {
        static std::mutex mu;
        static hipStream_t shared[64] = {};
        std::lock_guard<std::mutex> lock(mu);
        if (device < 64) {
            if (!shared[device]) HIP_TRY(hipStreamCreateWithFlags(&shared[device], hipStreamNonBlocking));
            v->stream = shared[device];
        } else {
            HIP_TRY(hipStreamCreateWithFlags(&v->own_stream, hipStreamNonBlocking));
            v->stream = v->own_stream;
        }
    }
    const size_t N = (size_t)cfg->num_envs;
    v->grid = (int)((N + kBlock - 1) / kBlock);
    DevEnv &d = v->d;
    d.N = cfg->num_envs, d.max_steps = cfg->max_episode_steps;
    d.solver_newton = (cfg->reserved[0] & MI_CFG_SOLVER_NEWTON) ? 1 : 0;
    d.tab_fickle_rows = (cfg->kind == MI_ENV_TABULAR && cfg->params[2] != 0.0) ? 1 : 0;
    for (int k = 0; k < 16; k++) d.P.p[k] = cfg->params[k];
    HIP_TRY(hipMalloc(&d.state, sizeof(double) * v->lay.state_dim * N));
    HIP_TRY(hipMalloc(&d.meta, sizeof(uint32_t) * N));
    HIP_TRY(hipMalloc(&d.rng, sizeof(uint64_t) * 4 * N));
    HIP_TRY(hipMalloc(&d.ep_ret, sizeof(double) * N));
    HIP_TRY(hipMalloc(&d.ep_len, sizeof(int32_t) * N));
    HIP_TRY(hipMalloc(&d.blk_count, sizeof(uint64_t) * 4 * v->grid));
    HIP_TRY(hipMalloc(&d.blk_ret, sizeof(double) * v->grid));
    HIP_TRY(hipMalloc(&v->d_pow2, sizeof(PcgJump) * 64));
    HIP_TRY(hipMalloc(&v->d_act_lane, sizeof(uint64_t) * 2 * N));
    v->shared_rng = (cfg->reserved[0] & MI_CFG_SHARED_RNG) != 0;
    if (v->shared_rng) {
        HIP_TRY(hipMalloc(&v->shared.words, sizeof(uint64_t) * 8));
        HIP_TRY(hipMalloc(&v->d_shared_pow2, sizeof(PcgJump) * 64));
        HIP_TRY(hipMalloc(&v->shared.blk_done, sizeof(uint32_t) * v->grid));
        HIP_TRY(hipMalloc(&v->shared.blk_prefix, sizeof(uint32_t) * v->grid));
        HIP_TRY(hipMemsetAsync(v->shared.words, 0, sizeof(uint64_t) * 8, v->stream));
        HIP_TRY(hipMemsetAsync(v->shared.blk_done, 0, sizeof(uint32_t) * v->grid, v->stream));
        HIP_TRY(hipMemsetAsync(v->shared.blk_prefix, 0, sizeof(uint32_t) * v->grid, v->stream));
        v->shared.pow2 = v->d_shared_pow2, v->shared.low = -0.05, v->shared.high = 0.05;
    }
    HIP_TRY(hipMemsetAsync(d.state, 0, sizeof(double) * v->lay.state_dim * N, v->stream));
    HIP_TRY(hipMemsetAsync(d.meta, 0, sizeof(uint32_t) * N, v->stream));
    HIP_TRY(hipMemsetAsync(d.rng, 0, sizeof(uint64_t) * 4 * N, v->stream));
    HIP_TRY(hipMemsetAsync(d.ep_ret, 0, sizeof(double) * N, v->stream));
    HIP_TRY(hipMemsetAsync(d.ep_len, 0, sizeof(int32_t) * N, v->stream));
    HIP_TRY(hipMemsetAsync(d.blk_count, 0, sizeof(uint64_t) * 4 * v->grid, v->stream));
    HIP_TRY(hipMemsetAsync(d.blk_ret, 0, sizeof(double) * v->grid, v->stream));
    v->act_bytes = N * (v->lay.act_dtype == MI_I64 ? 8 : 4) * v->lay.act_dim;
    v->act_bytes_max = N * 8 * v->lay.act_dim;  // a Box kind may be handed float64 rows (mi_step_io.actions_dtype)
    v->obs_bytes = N * (v->lay.obs_dtype == MI_F32 ? sizeof(float) : sizeof(double)) * v->lay.obs_dim;
    v->info_bytes = N * sizeof(double) * (v->lay.info_dim > 0 ? v->lay.info_dim : 1);
    {
        const size_t sizes[9] = {v->obs_bytes, sizeof(double) * N, N, N, v->info_bytes, sizeof(double) * N, sizeof(int32_t) * N, v->obs_bytes, v->info_bytes};
        size_t off = 256, start[9];  // (the first 256 bytes are unused: the error word moved to its own page-locked word)
        for (int k = 0; k < 9; k++) start[k] = off, off = (off + sizes[k] + 255) & ~(size_t)255, v->off_end[k] = start[k] + sizes[k];
        v->out_bytes = off;
        HIP_TRY(hipMalloc(&v->d_out, v->out_bytes));
        HIP_TRY(hipHostMalloc((void **)&v->h_out, v->out_bytes, hipHostMallocDefault));
        HIP_TRY(hipMemsetAsync(v->d_out, 0, v->out_bytes, v->stream));
        memset(v->h_out, 0, v->out_bytes);
        HIP_TRY(hipHostMalloc((void **)&v->h_err, 256, hipHostMallocDefault));
        *v->h_err = 0;
        void *derr = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&derr, v->h_err, 0));
        d.error = reinterpret_cast<int *>(derr);
        HIP_TRY(hipHostGetDevicePointer((void **)&v->h_out_dev, v->h_out, 0));
        char *D = v->d_out, *H = v->h_out;
        v->d_obs = D + start[0], v->d_reward = (double *)(D + start[1]), v->d_term = (uint8_t *)(D + start[2]), v->d_trunc = (uint8_t *)(D + start[3]);
        v->d_info = (double *)(D + start[4]), v->d_epret = (double *)(D + start[5]), v->d_eplen = (int32_t *)(D + start[6]);
        v->d_final = D + start[7], v->d_final_info = (double *)(D + start[8]);
        HIP_TRY(hipMalloc(&v->d_actions, v->act_bytes_max));
        HIP_TRY(hipHostMalloc(&v->h_actions, v->act_bytes_max, hipHostMallocDefault));
        memset(v->h_actions, 0, v->act_bytes_max);
        v->h_io.actions = v->h_actions, v->h_io.obs = H + start[0], v->h_io.reward = (double *)(H + start[1]);
        v->h_io.terminated = (uint8_t *)(H + start[2]), v->h_io.truncated = (uint8_t *)(H + start[3]), v->h_io.info = (double *)(H + start[4]);
        v->h_io.episode_return = (double *)(H + start[5]), v->h_io.episode_length = (int32_t *)(H + start[6]);
        v->h_io.final_obs = H + start[7], v->h_io.final_info = (double *)(H + start[8]);
    }
    HIP_TRY(hipMalloc(&v->d_mask, N));
    HIP_TRY(hipMalloc(&v->d_words, sizeof(uint64_t) * 4 * N));
    if (is_mj(cfg->kind)) {
        HIP_TRY(hipMalloc(&v->d_extras, sizeof(double) * v->extras_dim * N));
        HIP_TRY(hipMalloc(&v->d_act_scratch, v->act_bytes));
        HIP_TRY(hipMalloc(&v->d_obs_scratch, v->obs_bytes));
        HIP_TRY(hipMemsetAsync(v->d_extras, 0, sizeof(double) * v->extras_dim * N, v->stream));
    }
    HIP_TRY(hipStreamSynchronize(v->stream));
    return MI_OK;
}

void mi_destroy(mi_vecenv *v) {
    if (!v) return;
    (void)hipSetDevice(v->device);
    (void)hipStreamSynchronize(v->stream);
    void *ptrs[] = {v->d.state, v->d.meta, v->d.rng, v->d.ep_ret, v->d.ep_len, v->d.blk_count, v->d.blk_ret, v->d_out,
                    v->d_pow2, v->d_act_lane, v->d_act_stage, v->d_actions, v->d_mask, v->d_words, v->d_extras, v->d_act_scratch, v->d_obs_scratch,
                    v->shared.words, v->d_shared_pow2, v->shared.blk_done, v->shared.blk_prefix};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (v->h_out) (void)hipHostFree(v->h_out);
    if (v->h_err) (void)hipHostFree(v->h_err);
    if (v->h_actions) (void)hipHostFree(v->h_actions);
    for (void *p : v->tab_bufs)
        if (p) (void)hipFree(p);
    if (v->d_epi_partial) (void)hipFree(v->d_epi_partial);
    if (v->own_stream) (void)hipStreamDestroy(v->own_stream);
    delete v;
}

// The reference's stateful vector wrappers as the output stage of the classic-control step kernel (include/mi355env.h mi_step_epilogue).
int mi_set_step_epilogue(mi_vecenv *v, const mi_step_epilogue *e) {
    if (!v) return fail(MI_ERR_INVALID_ARGUMENT, "null env");
    if (v->shared_rng && e) return fail(MI_ERR_UNSUPPORTED, "MI_CFG_SHARED_RNG: the step epilogue belongs to the per-sub-environment step kernel");
    if (set_device(v)) return MI_ERR_HIP;
    HIP_TRY(hipStreamSynchronize(v->stream));
    if (!e) {
        v->has_epi = false;
        return MI_OK;
    }
    if (is_mj(v->cfg.kind) || is_tab(v->cfg.kind))
        return fail(MI_ERR_INVALID_ARGUMENT, "mi_set_step_epilogue: only the classic-control kinds fuse the wrappers into their step kernel (use the mi_normalize_* passes)");
    const mi_running_stats *o = e->obs_rms, *r = e->return_rms;
    if (o && (o->dim != v->lay.obs_dim || o->dtype != MI_F32 || o->device != v->device))
        return fail(MI_ERR_INVALID_ARGUMENT, "mi_set_step_epilogue: obs_rms must be float32 statistics of obs_dim columns on the env's device");
    if (r && (r->dim != 1 || r->dtype != MI_F64 || r->device != v->device || !e->accumulated || !e->prev_done))
        return fail(MI_ERR_INVALID_ARGUMENT, "mi_set_step_epilogue: return_rms must be scalar float64 statistics on the env's device, with accumulated and prev_done");
    if ((o && !(e->obs_epsilon > 0)) || (r && (!(e->reward_epsilon > 0) || !(e->gamma >= 0 && e->gamma <= 1))))
        return fail(MI_ERR_INVALID_ARGUMENT, "mi_set_step_epilogue: epsilon must be positive, gamma in [0, 1]");
    if (!v->d_epi_partial) HIP_TRY(hipMalloc(&v->d_epi_partial, sizeof(double) * kEpiCols * (size_t)v->grid));
    EpiDev d = {};
    d.obs_on = o != nullptr, d.obs_update = e->obs_update != 0, d.ret_on = r != nullptr, d.ret_update = e->reward_update != 0;
    d.same_step = v->cfg.autoreset_mode == MI_AUTORESET_SAME_STEP;
    d.clip_pre = e->clip_pre & 3, d.clip_post = e->clip_post & 3;
    d.gamma = (float)e->gamma, d.obs_eps = e->obs_epsilon, d.ret_eps = e->reward_epsilon;
    d.pre_lo = e->clip_pre_min, d.pre_hi = e->clip_pre_max, d.post_lo = e->clip_post_min, d.post_hi = e->clip_post_max;
    d.acc = e->accumulated, d.prev_done = e->prev_done;
    d.partial = v->d_epi_partial, d.step_blocks = v->grid, d.fold_in_finish = v->grid <= kEpiFoldMax;
    v->epi_host = *e;
    v->epi = d, v->has_epi = d.obs_on || d.ret_on || d.clip_pre || d.clip_post;
    return MI_OK;
}

int mi_get_layout(const mi_vecenv *v, mi_layout *out) {
    if (!v || !out) return fail(MI_ERR_INVALID_ARGUMENT, "null argument");
    *out = v->lay;
    return MI_OK;
}

int mi_set_stream(mi_vecenv *v, void *hip_stream) {
    if (!v) return fail(MI_ERR_INVALID_ARGUMENT, "null env");
    hipStream_t s = (hipStream_t)hip_stream;  // NULL is the legacy default stream (torch's default current stream)
    if (s != v->stream) {
        if (set_device(v)) return MI_ERR_HIP;
        // Order the hand-over between streams with a host synchronise -- unless the NEW stream is being captured into a graph: a synchronising call
        // during a (global-mode) capture is not allowed by the stream-capture rules, even on another stream.  HipVectorEnv.capture_steps synchronises
        // the engine BEFORE it opens the capture, so nothing is pending on the old stream then.
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (s && hipStreamIsCapturing(s, &cap) != hipSuccess) {
            (void)hipGetLastError();
            cap = hipStreamCaptureStatusNone;
        }
        if (cap == hipStreamCaptureStatusNone) HIP_TRY(hipStreamSynchronize(v->stream));
        v->stream = s;
    }
    return MI_OK;
}

int mi_synchronize(mi_vecenv *v) {
    if (!v) return fail(MI_ERR_INVALID_ARGUMENT, "null env");
    if (set_device(v)) return MI_ERR_HIP;
    return check_device_error(v);
}

static int upload_mask(mi_vecenv *v, const uint8_t *mask, const uint8_t **d_mask) {
    *d_mask = nullptr;
    if (!mask) return MI_OK;
    HIP_TRY(hipMemcpyAsync(v->d_mask, mask, (size_t)v->cfg.num_envs, hipMemcpyHostToDevice, v->stream));
    *d_mask = v->d_mask;
    return MI_OK;
}

// MI_CFG_SHARED_RNG: (re)base the one generator -- its words, a zero draw counter and the jump table of its increment
static int shared_rebase(mi_vecenv *v, const Pcg64 &g) {
    v->shared_base = g;
    const uint64_t words[8] = {(uint64_t)(g.state >> 64), (uint64_t)g.state, (uint64_t)(g.inc >> 64), (uint64_t)g.inc, 0, 0, 0, 0};
    PcgJump tab[64];
    for (int j = 0; j < 64; j++) tab[j] = pcg_jump(g.inc, (u128)1 << j);
    HIP_TRY(hipMemcpyAsync(v->shared.words, words, sizeof words, hipMemcpyHostToDevice, v->stream));
    HIP_TRY(hipMemcpyAsync(v->d_shared_pow2, tab, sizeof tab, hipMemcpyHostToDevice, v->stream));
    HIP_TRY(hipStreamSynchronize(v->stream));  // both sources are on this stack frame
    v->seeded = true;
    return MI_OK;
}

int mi_seed(mi_vecenv *v, const uint64_t *pcg, const uint8_t *mask) {
    if (!v || !pcg) return fail(MI_ERR_INVALID_ARGUMENT, "null argument");
    if (set_device(v)) return MI_ERR_HIP;
    if (v->shared_rng) {
        Pcg64 g;
        g.state = make_u128(pcg[0], pcg[1]), g.inc = make_u128(pcg[2], pcg[3]);
        return shared_rebase(v, g);
    }
    const uint8_t *dm;
    if (int rc = upload_mask(v, mask, &dm)) return rc;
    HIP_TRY(hipMemcpyAsync(v->d_words, pcg, sizeof(uint64_t) * 4 * (size_t)v->cfg.num_envs, hipMemcpyHostToDevice, v->stream));
    hipLaunchKernelGGL(seed_words_kernel, dim3(v->grid), dim3(kBlock), 0, v->stream, v->d, (const uint64_t *)v->d_words, dm);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(v->stream));  // the host buffers may be reused by the caller
    v->seeded = true;
    return MI_OK;
}

int mi_seed_sequence(mi_vecenv *v, uint64_t base_seed, uint64_t first_index, const uint8_t *mask) {
    if (!v) return fail(MI_ERR_INVALID_ARGUMENT, "null env");
    if (set_device(v)) return MI_ERR_HIP;
    if (v->shared_rng) {  // Generator(PCG64(SeedSequence(seed))) -- the same host arithmetic seed_sequence_kernel runs per lane
        if (first_index != 0) return fail(MI_ERR_UNSUPPORTED, "MI_CFG_SHARED_RNG: one generator for all sub-environments, they do not shard (first_index must be 0)");
        uint64_t w[4];
        seed_sequence_words(base_seed, w);
        Pcg64 g;
        g.srandom(w);
        return shared_rebase(v, g);
    }
    const uint8_t *dm;
    if (int rc = upload_mask(v, mask, &dm)) return rc;
    hipLaunchKernelGGL(seed_sequence_kernel, dim3(v->grid), dim3(kBlock), 0, v->stream, v->d, base_seed + first_index, dm);
    HIP_TRY(hipGetLastError());
    if (mask) HIP_TRY(hipStreamSynchronize(v->stream));
    v->seeded = true;
    return MI_OK;
}

int mi_get_rng(mi_vecenv *v, uint64_t *pcg) {
    if (!v || !pcg) return fail(MI_ERR_INVALID_ARGUMENT, "null argument");
    if (set_device(v)) return MI_ERR_HIP;
    const size_t N = (size_t)v->cfg.num_envs;
    if (v->shared_rng) {  // the base generator advanced by the draws the device has taken: every row reports the one generator
        uint64_t words[8];
        HIP_TRY(hipMemcpyAsync(words, v->shared.words, sizeof words, hipMemcpyDeviceToHost, v->stream));
        HIP_TRY(hipStreamSynchronize(v->stream));
        const PcgJump j = pcg_jump(v->shared_base.inc, (u128)words[4]);
        const u128 st = j.mult * v->shared_base.state + j.plus;
        for (size_t i = 0; i < N; i++) {
            pcg[4 * i] = (uint64_t)(st >> 64), pcg[4 * i + 1] = (uint64_t)st;
            pcg[4 * i + 2] = (uint64_t)(v->shared_base.inc >> 64), pcg[4 * i + 3] = (uint64_t)v->shared_base.inc;
        }
        return MI_OK;
    }
    std::vector<uint64_t> soa(4 * N);
    HIP_TRY(hipMemcpyAsync(soa.data(), v->d.rng, sizeof(uint64_t) * 4 * N, hipMemcpyDeviceToHost, v->stream));
    HIP_TRY(hipStreamSynchronize(v->stream));
    for (size_t i = 0; i < N; i++)
        for (int k = 0; k < 4; k++) pcg[4 * i + k] = soa[k * N + i];
    return MI_OK;
}

int mi_reset(mi_vecenv *v, const uint8_t *mask, const double *bounds, void *obs, int loc) {
    if (!v) return fail(MI_ERR_INVALID_ARGUMENT, "null env");
    if (!v->seeded) return fail(MI_ERR_STATE, "reset before seeding");
    if (set_device(v)) return MI_ERR_HIP;
    const uint8_t *dm = mask;
    void *dobs = obs;
    // classic kinds, host caller: the pinned block is where the last observations are (the step kernel writes it directly), so that is what
    // a partial reset must leave untouched for the sub-environments it does not reset
    const bool pinned_obs = loc == MI_HOST && !is_mj(v->cfg.kind);
    if (loc == MI_HOST) {
        if (int rc = upload_mask(v, mask, &dm)) return rc;
        dobs = v->d_obs;  // persistent: rows of un-reset sub-envs keep their last observation (sync_vector_env.py:261)
        if (pinned_obs) dobs = v->h_out_dev + ((char *)v->d_obs - v->d_out);  // (pinned() is declared further down)
    }
    const int has_bounds = bounds != nullptr;
    const double b0 = has_bounds ? bounds[0] : 0.0, b1 = has_bounds ? bounds[1] : 0.0;
    if (is_tab(v->cfg.kind)) {
        if (!v->tab_loaded) return fail(MI_ERR_STATE, "tabular environment without a table: call mi_tabular_load first");
        hipLaunchKernelGGL(tab_reset_kernel, dim3(v->grid), dim3(kBlock), 0, v->stream, v->d, dm, (int64_t *)dobs);
        HIP_TRY(hipGetLastError());
        v->was_reset = true;
        if (loc == MI_HOST) {
            HIP_TRY(hipStreamSynchronize(v->stream));
            if (obs && obs != v->h_io.obs) memcpy(obs, v->h_io.obs, v->obs_bytes);
        }
        return MI_OK;
    }
    int rc = is_mj(v->cfg.kind) ? dispatch_mj(v->cfg.kind, [&](auto env) -> int {
        using E = decltype(env);
        hipLaunchKernelGGL((mj_reset_kernel<E>), dim3(v->grid), dim3(kBlock), 0, v->stream, v->d, dm, (double *)dobs, v->lay.obs_dim);
        HIP_TRY(hipGetLastError());
        return (int)MI_OK;
    }) : v->shared_rng ? MI_OK : mi_classic::reset(v, dm, has_bounds, b0, b1, (float *)dobs);
    if (v->shared_rng) {  // CartPoleVectorEnv.reset (cartpole.py:481-503): every sub-environment, bounds kept for the autoresets
        if (mask) return fail(MI_ERR_UNSUPPORTED, "MI_CFG_SHARED_RNG: CartPoleVectorEnv.reset resets every sub-environment (no reset_mask)");
        v->shared.low = has_bounds ? b0 : -0.05, v->shared.high = has_bounds ? b1 : 0.05;
        rc = mi_classic::shared_reset(v, (float *)dobs);
    }
    if (rc) return rc;
    v->was_reset = true;
    if (loc == MI_HOST) {
        if (obs && !pinned_obs) HIP_TRY(hipMemcpyAsync(obs, v->d_obs, v->obs_bytes, hipMemcpyDeviceToHost, v->stream));
        HIP_TRY(hipStreamSynchronize(v->stream));
        if (obs && pinned_obs && obs != v->h_io.obs) memcpy(obs, v->h_io.obs, v->obs_bytes);
    }
    return MI_OK;
}

// the address, as the device sees it, of the place in the page-locked block that mirrors `device_ptr` of the device block
static void *pinned(const mi_vecenv *v, const void *device_ptr) { return v->h_out_dev + ((const char *)device_ptr - v->d_out); }

// ---- the action stream's position: host copy (act_rng) <-> per-lane device states (d_act_lane) ------------------------------------------------
static ActionStream action_stream(const mi_vecenv *v) {
    ActionStream as;
    memset(&as, 0, sizeof as);
    as.state_hi = (uint64_t)(v->act_rng.state >> 64), as.state_lo = (uint64_t)v->act_rng.state;
    as.inc_hi = (uint64_t)(v->act_rng.inc >> 64), as.inc_lo = (uint64_t)v->act_rng.inc;
    as.pow2 = v->d_pow2, as.jump_n = v->jump_n;
    return as;
}
// bring the host copy up to date: lane 0's state is the position stepped once (the state whose output is the next draw)
static int action_sync_host(mi_vecenv *v) {
    if (!v->act_on_device) return MI_OK;
    uint64_t w[2];
    HIP_TRY(hipMemcpyAsync(&w[0], v->d_act_lane, sizeof(uint64_t), hipMemcpyDeviceToHost, v->stream));
    HIP_TRY(hipMemcpyAsync(&w[1], v->d_act_lane + (size_t)v->cfg.num_envs, sizeof(uint64_t), hipMemcpyDeviceToHost, v->stream));
    HIP_TRY(hipStreamSynchronize(v->stream));
    Pcg64 g;
    g.state = make_u128(w[0], w[1]), g.inc = v->act_rng.inc;
    g.unstep();
    v->act_rng.state = g.state;
    v->act_on_device = false;
    return MI_OK;
}
// per-lane states for the host copy's position (skip-ahead by i * act_dim + 1 draws per lane); a no-op while they are current
static int action_prepare_lanes(mi_vecenv *v) {
    if (v->act_lane_valid) return MI_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (v->stream && hipStreamIsCapturing(v->stream, &cap) != hipSuccess) {
        (void)hipGetLastError();
        cap = hipStreamCaptureStatusNone;
    }
    if (cap != hipStreamCaptureStatusNone)
        return fail(MI_ERR_STATE, "the on-device policy's lane states must exist before a stream capture: call mi_action_sample(env, 0, NULL, MI_DEVICE) first");
    hipLaunchKernelGGL(act_init_kernel, dim3(v->grid), dim3(kBlock), 0, v->stream, action_stream(v), v->d_act_lane, v->cfg.num_envs, v->lay.act_dim);
    HIP_TRY(hipGetLastError());
    v->act_lane_valid = true;
    return MI_OK;
}
static int action_spec(const mi_vecenv *v, ActSpec *sp) {
    memset(sp, 0, sizeof *sp);
    sp->D = v->lay.act_dim;
    if (sp->D > 24) return fail(MI_ERR_UNSUPPORTED, "action rows wider than 24");
    const int kind = v->cfg.kind;
    if (is_tab(kind))
        sp->n = (double)v->d.tab.nA;
    else if (kind < kClassicKinds && kNumActions[kind])
        sp->n = (double)kNumActions[kind];
    else if (kind == MI_ENV_PENDULUM)
        sp->lo[0] = -2.0, sp->range[0] = 2.0 - (-2.0);  // Box(-max_torque, max_torque) (pendulum.py:112-114)
    else if (kind == MI_ENV_MOUNTAIN_CAR_CONTINUOUS)
        sp->lo[0] = -1.0, sp->range[0] = 1.0 - (-1.0);  // Box(min_action, max_action) (continuous_mountain_car.py:137-139)
    else
        return dispatch_mj(kind, [&](auto env) -> int {  // Box(low, high, (nu,), float32) from the float32 ctrlrange (mujoco_env.py:113-117)
            using E = decltype(env);
            for (int u = 0; u < E::NU; u++) {
                const double lo = (double)(float)E::Model::actuator_ctrlrange[u][0], hi = (double)(float)E::Model::actuator_ctrlrange[u][1];
                sp->lo[u] = lo, sp->range[u] = hi - lo;
            }
            return (int)MI_OK;
        });
    return MI_OK;
}
// T batches of action_space.sample() into `dst` (device), from the lane states
static int action_sample_device(mi_vecenv *v, int T, void *dst) {
    ActSpec sp;
    if (int rc = action_spec(v, &sp)) return rc;
    if (int rc = action_prepare_lanes(v)) return rc;
    hipLaunchKernelGGL(act_sample_kernel, dim3(v->grid), dim3(kBlock), 0, v->stream, sp, action_stream(v), v->d_act_lane, dst, T, v->cfg.num_envs);
    HIP_TRY(hipGetLastError());
    v->act_on_device = true;
    return MI_OK;
}

// Enqueue one vector step.  loc == MI_HOST: actions go through the pinned staging block (one H2D), the kernel writes the device output
// block, and ONE D2H brings back the prefix of it that the caller asked for -- or, for the classic kinds, the kernel works on the pinned block
// directly and there is no copy at all; nothing is synchronised here.
static int step_enqueue(mi_vecenv *v, const mi_step_io *io, int loc) {
    if (!v || !io) return fail(MI_ERR_INVALID_ARGUMENT, "null argument");
    if (!v->was_reset) return fail(MI_ERR_STATE, "step before reset");
    const bool sample = io->actions == nullptr;  // the on-device policy: step(action_space.sample()) without the host
    if (sample && loc != MI_DEVICE) return fail(MI_ERR_INVALID_ARGUMENT, "actions is NULL (the on-device policy steps device buffers: loc == MI_DEVICE; host callers draw batches with mi_action_sample)");
    if (sample && !v->act_seeded) return fail(MI_ERR_STATE, "a step without actions needs mi_action_seed");
    if (v->has_pending) return fail(MI_ERR_STATE, "mi_step_async: the previous asynchronous step has not been waited for (mi_step_wait)");
    if (set_device(v)) return MI_ERR_HIP;
    // Device callers are asynchronous: an action outside the space (or a finished sub-environment stepped under DISABLED) is recorded by the
    // kernel in the page-locked error word and raised HERE, by the first later step that finds it -- a plain host read, no synchronisation.
    // The word is only CLEARED once the stream is idle (error path only: the common case stays a plain read).
    if (loc == MI_DEVICE && *v->h_err) return check_device_error(v);
    const size_t N = (size_t)v->cfg.num_envs;
    // element type of the action rows: Box kinds take float32 rows or un-rounded float64 rows (include/mi355env.h mi_step_io.actions_dtype)
    const bool box = v->lay.act_dtype == MI_F32;
    if (box && io->actions_dtype != MI_F32 && io->actions_dtype != MI_F64 && io->actions_dtype != MI_F64_WEAK)
        return fail(MI_ERR_INVALID_ARGUMENT, "actions_dtype must be MI_F32, MI_F64 or MI_F64_WEAK");
    const int act_kind = (box && !sample) ? io->actions_dtype : (int)MI_F32;
    const size_t act_bytes = act_kind == MI_F32 ? v->act_bytes : v->act_bytes_max;
    StepPtrs p;
    memset(&p, 0, sizeof p);
    bool zc = false;
    // classic control (without a wrapper epilogue) and ToyText draw inside their step kernel; every other configuration runs the stand-alone
    // sampler first and steps on what it wrote
    const bool fused_sample = sample && !is_mj(v->cfg.kind) && !v->shared_rng && !v->has_epi;
    if (loc == MI_HOST) {
        // validate before anything is mutated (cartpole.py:165-167 asserts action_space.contains(action))
        const int na = is_tab(v->cfg.kind) ? v->d.tab.nA : kNumActions[v->cfg.kind];
        if (na) {
            const int64_t *a = (const int64_t *)io->actions;
            for (size_t i = 0; i < N; i++)
                if (a[i] < 0 || a[i] >= na) return fail(MI_ERR_INVALID_ARGUMENT, "action outside the action space");
        }
        if (io->actions != v->h_actions) memcpy(v->h_actions, io->actions, act_bytes);  // callers that fill the pinned array skip this
        // Classic control and ToyText: the step kernel reads the actions from and writes its outputs to the PINNED block itself (coalesced rows of at most
        // 24 bytes per lane stream over PCIe while the kernel runs): no copy-engine hand-offs, 97 -> 85 us per step at 65 536 sub-environments
        // (73 us when the caller's policy writes into the pinned action array).  MI355ENV_ZEROCOPY=0 restores the staged copies (A/B).  Not
        // with a finishing epilogue pass (it would re-read the batch over PCIe) and not for the wide float64 rows of the MuJoCo kinds.
        static const bool zero_copy = !(getenv("MI355ENV_ZEROCOPY") && getenv("MI355ENV_ZEROCOPY")[0] == '0');
        zc = zero_copy && !is_mj(v->cfg.kind) && !(v->has_epi && (v->epi.obs_on || v->epi.ret_on));
        if (zc) {
            void *da = nullptr;
            HIP_TRY(hipHostGetDevicePointer(&da, v->h_actions, 0));
            p.actions = da;
            p.obs = (float *)pinned(v, v->d_obs), p.reward = (double *)pinned(v, v->d_reward);
            p.terminated = (uint8_t *)pinned(v, v->d_term), p.truncated = (uint8_t *)pinned(v, v->d_trunc);
            p.final_obs = io->final_obs ? (float *)pinned(v, v->d_final) : nullptr;
            p.ep_ret = io->episode_return ? (double *)pinned(v, v->d_epret) : nullptr;
            p.ep_len = io->episode_length ? (int32_t *)pinned(v, v->d_eplen) : nullptr;
        } else {
            HIP_TRY(hipMemcpyAsync(v->d_actions, v->h_actions, act_bytes, hipMemcpyHostToDevice, v->stream));
            p.actions = v->d_actions;
            p.obs = (float *)v->d_obs, p.reward = v->d_reward, p.terminated = v->d_term, p.truncated = v->d_trunc;
            p.final_obs = io->final_obs ? (float *)v->d_final : nullptr;
            p.ep_ret = io->episode_return ? v->d_epret : nullptr;
            p.ep_len = io->episode_length ? v->d_eplen : nullptr;
        }
    } else {
        p.actions = io->actions;
        p.obs = (float *)io->obs, p.reward = io->reward, p.terminated = io->terminated, p.truncated = io->truncated;
        p.final_obs = (float *)io->final_obs, p.ep_ret = io->episode_return, p.ep_len = io->episode_length;
        if (fused_sample) {
            if (int rc = action_prepare_lanes(v)) return rc;
            p.act_lane = v->d_act_lane, p.actions_out = io->actions_out, p.act_jump = v->jump_n;
            v->act_on_device = true;
        } else if (sample) {
            void *dst = io->actions_out ? io->actions_out : (is_mj(v->cfg.kind) ? (void *)v->d_act_scratch : v->d_actions);
            if (int rc = action_sample_device(v, 1, dst)) return rc;
            p.actions = dst;
        }
    }
    double *dinfo = loc == MI_HOST ? (io->info ? v->d_info : nullptr) : io->info;
    double *dfinfo = loc == MI_HOST ? (io->final_info ? v->d_final_info : nullptr) : io->final_info;
    if (zc) dinfo = dinfo ? (double *)pinned(v, dinfo) : nullptr, dfinfo = dfinfo ? (double *)pinned(v, dfinfo) : nullptr;
    int rc;
    if (is_tab(v->cfg.kind)) {
        const TabStepPtrs tp = {(const int64_t *)p.actions, (int64_t *)p.obs, p.reward, p.terminated, p.truncated, (int64_t *)p.final_obs,
                                p.ep_ret, p.ep_len, dinfo, dfinfo, p.act_lane, (int64_t *)p.actions_out, p.act_jump};
        const dim3 g(v->grid), b(kBlock);
        if (fused_sample) {
            switch (v->cfg.autoreset_mode) {
            case MI_AUTORESET_NEXT_STEP: hipLaunchKernelGGL((tab_step_kernel<MI_AUTORESET_NEXT_STEP, true>), g, b, 0, v->stream, v->d, tp); break;
            case MI_AUTORESET_SAME_STEP: hipLaunchKernelGGL((tab_step_kernel<MI_AUTORESET_SAME_STEP, true>), g, b, 0, v->stream, v->d, tp); break;
            default: hipLaunchKernelGGL((tab_step_kernel<MI_AUTORESET_DISABLED, true>), g, b, 0, v->stream, v->d, tp); break;
            }
        } else
        switch (v->cfg.autoreset_mode) {
        case MI_AUTORESET_NEXT_STEP: hipLaunchKernelGGL((tab_step_kernel<MI_AUTORESET_NEXT_STEP>), g, b, 0, v->stream, v->d, tp); break;
        case MI_AUTORESET_SAME_STEP: hipLaunchKernelGGL((tab_step_kernel<MI_AUTORESET_SAME_STEP>), g, b, 0, v->stream, v->d, tp); break;
        default: hipLaunchKernelGGL((tab_step_kernel<MI_AUTORESET_DISABLED>), g, b, 0, v->stream, v->d, tp); break;
        }
        HIP_TRY(hipGetLastError());
        rc = MI_OK;
    } else if (is_mj(v->cfg.kind)) {
        const MjStepPtrs mp = {p.actions, (double *)p.obs, p.reward, p.terminated, p.truncated, (double *)p.final_obs,
                               p.ep_ret, p.ep_len, dinfo, dfinfo, v->lay.obs_dim, nullptr, act_kind != MI_F32};
        if (!mp.obs) return fail(MI_ERR_INVALID_ARGUMENT, "obs is NULL");
        rc = dispatch_mj(v->cfg.kind, [&](auto env) -> int { return launch_mj_step<decltype(env)>(v, mp); });
    } else if (v->shared_rng) {
        if (p.final_obs) return fail(MI_ERR_INVALID_ARGUMENT, "MI_CFG_SHARED_RNG is NEXT_STEP only: no final_obs");
        rc = mi_classic::shared_step(v, p);
    } else {
        rc = mi_classic::step(v, p, act_kind);
    }
    if (rc) return rc;
    if (loc == MI_HOST) {
        // one copy: the block prefix up to the last section somebody wants (sections in h_io order: obs, reward, terminated, truncated,
        // info, episode_return, episode_length, final_obs, final_info)
        const void *want[9] = {io->obs, io->reward, io->terminated, io->truncated, io->info, io->episode_return, io->episode_length, io->final_obs, io->final_info};
        size_t n = 256;
        for (int k = 0; k < 9; k++)
            if (want[k]) n = v->off_end[k];
        if (!zc && n > 256) HIP_TRY(hipMemcpyAsync(v->h_out + 256, v->d_out + 256, n - 256, hipMemcpyDeviceToHost, v->stream));
        v->pending = *io, v->pending_bytes = n, v->has_pending = true;
    }
    return MI_OK;
}

// Wait for the step enqueued by step_enqueue(MI_HOST): synchronise, raise the device error word, and hand the
// sections to the caller's arrays (a plain memcpy out of pinned memory; skipped for arrays that ARE the pinned ones, mi_host_buffers).
static int step_finish(mi_vecenv *v) {
    if (!v->has_pending) return MI_OK;
    v->has_pending = false;
    HIP_TRY(hipStreamSynchronize(v->stream));
    if (int rc = raise_device_error(v)) return rc;
    const size_t N = (size_t)v->cfg.num_envs;
    const mi_step_io &u = v->pending, &h = v->h_io;
    auto out = [](void *dst, const void *src, size_t n) {
        if (dst && dst != src) memcpy(dst, src, n);
    };
    out(u.obs, h.obs, v->obs_bytes), out(u.reward, h.reward, sizeof(double) * N), out(u.terminated, h.terminated, N), out(u.truncated, h.truncated, N);
    out(u.info, h.info, v->info_bytes), out(u.episode_return, h.episode_return, sizeof(double) * N), out(u.episode_length, h.episode_length, sizeof(int32_t) * N);
    out(u.final_obs, h.final_obs, v->obs_bytes), out(u.final_info, h.final_info, v->info_bytes);
    return MI_OK;
}

int mi_step(mi_vecenv *v, const mi_step_io *io, int loc) {
    if (int rc = step_enqueue(v, io, loc)) return rc;
    return loc == MI_HOST ? step_finish(v) : (int)MI_OK;
}

int mi_step_async(mi_vecenv *v, const mi_step_io *io) { return step_enqueue(v, io, MI_HOST); }

int mi_step_wait(mi_vecenv *v) {
    if (!v) return fail(MI_ERR_INVALID_ARGUMENT, "null env");
    if (!v->has_pending) return fail(MI_ERR_STATE, "mi_step_wait without a pending mi_step_async");
    if (set_device(v)) return MI_ERR_HIP;
    return step_finish(v);
}

int mi_host_buffers(mi_vecenv *v, mi_step_io *out) {
    if (!v || !out) return fail(MI_ERR_INVALID_ARGUMENT, "null argument");
    *out = v->h_io;
    return MI_OK;
}

int mi_tabular_load(mi_vecenv *v, const mi_tabular_table *t) {
    if (!v || !t) return fail(MI_ERR_INVALID_ARGUMENT, "null argument");
    if (v->cfg.kind != MI_ENV_TABULAR) return fail(MI_ERR_INVALID_ARGUMENT, "not a tabular environment");
    if (t->num_states < 1 || t->num_actions < 1 || t->max_outcomes < 1 || t->num_tables < 0) return fail(MI_ERR_INVALID_ARGUMENT, "empty table");
    const size_t M = t->num_tables > 1 ? (size_t)t->num_tables : 1;
    if ((M > 1) != (t->env_table != nullptr)) return fail(MI_ERR_INVALID_ARGUMENT, "env_table goes with num_tables > 1");
    if (M > 1 && v->cfg.params[2] != 0.0) return fail(MI_ERR_UNSUPPORTED, "the fickle Taxi has one table");
    for (size_t i = 0; M > 1 && i < (size_t)v->cfg.num_envs; i++)
        if (t->env_table[i] < 0 || (size_t)t->env_table[i] >= M) return fail(MI_ERR_INVALID_ARGUMENT, "env_table entry out of range");
    if (set_device(v)) return MI_ERR_HIP;
    const size_t cells = M * (size_t)t->num_states * t->num_actions, n = cells * t->max_outcomes;
    const void *src[8] = {t->csprob, t->prob, t->reward, t->isd_csprob, t->next_state, t->count, t->terminated, t->env_table};
    const size_t bytes[8] = {n * 8, n * 8, n * 8, M * (size_t)t->num_states * 8, n * 4, cells * 4, n, (size_t)v->cfg.num_envs * 4};
    for (int k = 0; k < 8; k++) {
        if (k == 7 && M == 1) {
            if (v->tab_bufs[k]) (void)hipFree(v->tab_bufs[k]);
            v->tab_bufs[k] = nullptr;
            continue;
        }
        if (!src[k]) return fail(MI_ERR_INVALID_ARGUMENT, "null table array");
        if (v->tab_bufs[k]) (void)hipFree(v->tab_bufs[k]);
        HIP_TRY(hipMalloc(&v->tab_bufs[k], bytes[k]));
        HIP_TRY(hipMemcpyAsync(v->tab_bufs[k], src[k], bytes[k], hipMemcpyHostToDevice, v->stream));
    }
    HIP_TRY(hipStreamSynchronize(v->stream));
    TabTable &tab = v->d.tab;
    tab.nS = t->num_states, tab.nA = t->num_actions, tab.K = t->max_outcomes;
    tab.csprob = (const double *)v->tab_bufs[0], tab.prob = (const double *)v->tab_bufs[1], tab.reward = (const double *)v->tab_bufs[2];
    tab.isd = (const double *)v->tab_bufs[3], tab.next = (const int32_t *)v->tab_bufs[4], tab.count = (const int32_t *)v->tab_bufs[5];
    tab.term = (const uint8_t *)v->tab_bufs[6], tab.env_table = (const int32_t *)v->tab_bufs[7];
    v->tab_start_state = -1;
    if (M == 1) {  // "first state whose cumulative probability exceeds u" is the same state for every u < 1 when that state's sum is already >= 1
        int first = 0;
        while (first < t->num_states && !(t->isd_csprob[first] > 0.0)) first++;
        if (first < t->num_states && t->isd_csprob[first] >= 1.0) v->tab_start_state = first;
    }
    v->tab_loaded = true;
    return MI_OK;
}

int mi_action_seed(mi_vecenv *v, const uint64_t pcg[4]) {
    if (!v || !pcg) return fail(MI_ERR_INVALID_ARGUMENT, "null argument");
    if (set_device(v)) return MI_ERR_HIP;
    const u128 inc = make_u128(pcg[2], pcg[3]);
    const bool same_inc = v->act_seeded && v->act_rng.inc == inc;
    v->act_rng.state = make_u128(pcg[0], pcg[1]);
    v->act_rng.inc = inc;
    v->act_lane_valid = false, v->act_on_device = false;  // the host copy IS the position again
    if (!same_inc) {
        PcgJump tab[64];
        for (int j = 0; j < 64; j++) tab[j] = pcg_jump(inc, (u128)1 << j);
        HIP_TRY(hipMemcpyAsync(v->d_pow2, tab, sizeof tab, hipMemcpyHostToDevice, v->stream));
        HIP_TRY(hipStreamSynchronize(v->stream));
        // per step a lane draws act_dim consecutive values, then skips to its slot in the next step's batch
        v->jump_n = pcg_jump(inc, (u128)v->cfg.num_envs * (u128)v->lay.act_dim - (u128)(v->lay.act_dim - 1));
    }
    v->act_seeded = true;
    return MI_OK;
}

int mi_rollout(mi_vecenv *v, int T, const mi_rollout_io *io) {
    if (!v || !io) return fail(MI_ERR_INVALID_ARGUMENT, "null argument");
    if (!v->was_reset) return fail(MI_ERR_STATE, "rollout before reset");
    if (T < 0) return fail(MI_ERR_INVALID_ARGUMENT, "T must be >= 0");
    if (v->cfg.autoreset_mode == MI_AUTORESET_DISABLED) return fail(MI_ERR_UNSUPPORTED, "rollout needs an autoreset mode");
    const bool sample = io->actions_in == nullptr;
    if (sample && !v->act_seeded) return fail(MI_ERR_STATE, "rollout without actions needs mi_action_seed");
    if (T == 0) return MI_OK;
    if (set_device(v)) return MI_ERR_HIP;
    if (sample)  // (earlier steps / samples may have left the position on the device)
        if (int rc = action_sync_host(v)) return rc;
    RolloutPtrs p = {io->actions_in, io->actions_out, io->obs, io->reward, io->terminated, io->truncated};
    const bool box = v->lay.act_dtype == MI_F32;
    if (box && !sample && io->actions_in_dtype != MI_F32 && io->actions_in_dtype != MI_F64 && io->actions_in_dtype != MI_F64_WEAK)
        return fail(MI_ERR_INVALID_ARGUMENT, "actions_in_dtype must be MI_F32, MI_F64 or MI_F64_WEAK");
    const int in_kind = (box && !sample) ? io->actions_in_dtype : (int)MI_F32, in_f64 = in_kind != MI_F32;
    if (in_f64 && io->actions_out) return fail(MI_ERR_INVALID_ARGUMENT, "actions_out echoes float32 rows: not with float64 actions_in");
    ActionStream as;
    memset(&as, 0, sizeof as);
    if (sample) {
        as.state_hi = (uint64_t)(v->act_rng.state >> 64), as.state_lo = (uint64_t)v->act_rng.state;
        as.inc_hi = (uint64_t)(v->act_rng.inc >> 64), as.inc_lo = (uint64_t)v->act_rng.inc;
        as.pow2 = v->d_pow2, as.jump_n = v->jump_n;
    }
    int rc;
    if (is_tab(v->cfg.kind)) {
        const dim3 g(v->grid), b(kBlock);
        const bool next = v->cfg.autoreset_mode == MI_AUTORESET_NEXT_STEP;
        // table in LDS when it fits next to the 160 KB of a CU and the copy is amortised over enough steps (Blackjack has no table)
        size_t lds = 0;
        if (v->d.tab.nS > 0 && T >= 8 && !v->d.tab.env_table) {  // (per-sub-environment tables stay in HBM / L2)
            const size_t cells = (size_t)v->d.tab.nS * v->d.tab.nA, rows = cells * v->d.tab.K;
            lds = (3 * rows + v->d.tab.nS) * sizeof(double) + (rows + cells) * sizeof(int32_t) + rows;
            lds = (lds + 15) & ~(size_t)15;
            if (lds > 150 * 1024) lds = 0;
        }
        const int lb = (int)lds;
        const int tk = v->d.tab.nS < 0 ? kTabBlackjack : (v->d.tab_fickle_rows ? kTabFickle : kTabPlain);
        auto launch = [&](auto mode, auto smp) {
            constexpr int M = decltype(mode)::value;
            constexpr bool S = decltype(smp)::value;
            if (tk == kTabBlackjack)  // (no table)
                hipLaunchKernelGGL((tab_rollout_kernel<M, S, false, kTabBlackjack>), g, b, 0, v->stream, v->d, p, as, T, lb);
            else if (tk == kTabFickle && lds)
                hipLaunchKernelGGL((tab_rollout_kernel<M, S, true, kTabFickle>), g, b, lds, v->stream, v->d, p, as, T, lb);
            else if (tk == kTabFickle)
                hipLaunchKernelGGL((tab_rollout_kernel<M, S, false, kTabFickle>), g, b, 0, v->stream, v->d, p, as, T, lb);
            else if (lds)
                hipLaunchKernelGGL((tab_rollout_kernel<M, S, true, kTabPlain>), g, b, lds, v->stream, v->d, p, as, T, lb);
            else
                hipLaunchKernelGGL((tab_rollout_kernel<M, S, false, kTabPlain>), g, b, 0, v->stream, v->d, p, as, T, lb);
        };
        typedef std::integral_constant<int, MI_AUTORESET_NEXT_STEP> NextT;
        typedef std::integral_constant<int, MI_AUTORESET_SAME_STEP> SameT;
        // the branch-free kernel for the collector's case: one plain table of <= 3 outcomes per (state, action) that fits into LDS in its packed form
        static const bool lean_on = !(getenv("MI355ENV_TAB_LEAN") && getenv("MI355ENV_TAB_LEAN")[0] == '0');  // "0": tab_rollout_kernel (A/B, tests)
        const int kl = v->d.tab.K == 1 ? 1 : 3;
        const size_t lean_lds = tk == kTabPlain && v->d.tab.nS > 0 ? (tab_lean_lds_bytes(v->d.tab.nS, v->d.tab.nA, kl) + 15) & ~(size_t)15 : 0;
        if (next && sample && lean_on && tk == kTabPlain && !v->d.tab.env_table && v->d.tab.K <= 3 && lean_lds <= 150 * 1024) {
            const bool full = p.actions_out && p.obs && p.reward && p.terminated && p.truncated;
            const int nA = v->d.tab.nA, start = v->tab_start_state;
            const bool apow2 = nA >= 2 && (nA & (nA - 1)) == 0;
            int shift = 64;
            for (int x = nA; apow2 && x > 1; x >>= 1) shift--;
            auto lean = [&](auto kl_c, auto full_c, auto one_c, auto pow_c) {
                hipLaunchKernelGGL((tab_rollout_lean_kernel<decltype(kl_c)::value, decltype(full_c)::value, decltype(one_c)::value, decltype(pow_c)::value>), g, b,
                                   lean_lds, v->stream, v->d, p, as, T, start, shift);
            };
            auto pick_pow = [&](auto kl_c, auto full_c, auto one_c) {
                if (apow2)
                    lean(kl_c, full_c, one_c, std::true_type());
                else
                    lean(kl_c, full_c, one_c, std::false_type());
            };
            auto pick_one = [&](auto kl_c, auto full_c) {
                if (start >= 0)
                    pick_pow(kl_c, full_c, std::true_type());
                else
                    pick_pow(kl_c, full_c, std::false_type());
            };
            auto pick_full = [&](auto kl_c) {
                if (full)
                    pick_one(kl_c, std::true_type());
                else
                    pick_one(kl_c, std::false_type());
            };
            if (kl == 1)
                pick_full(std::integral_constant<int, 1>());
            else
                pick_full(std::integral_constant<int, 3>());
        } else if (next && sample && lean_on && tk == kTabBlackjack) {
            const char *fs = getenv("MI355ENV_BJ_FORCE_SLOW");  // tests: every k-th lane-step through the general routines
            const int force_slow = fs ? atoi(fs) : 0;
            if (p.actions_out && p.obs && p.reward && p.terminated && p.truncated)
                hipLaunchKernelGGL((bj_rollout_lean_kernel<true>), g, b, 0, v->stream, v->d, p, as, T, force_slow);
            else
                hipLaunchKernelGGL((bj_rollout_lean_kernel<false>), g, b, 0, v->stream, v->d, p, as, T, force_slow);
        } else if (next && sample)
            launch(NextT(), std::true_type());
        else if (next)
            launch(NextT(), std::false_type());
        else if (sample)
            launch(SameT(), std::true_type());
        else
            launch(SameT(), std::false_type());
        HIP_TRY(hipGetLastError());
        rc = MI_OK;
    } else if (is_mj(v->cfg.kind)) {
        rc = dispatch_mj(v->cfg.kind, [&](auto env) -> int {
            using E = decltype(env);
            if (v->mj_coop) return launch_mj_rollout_coop<E>(v, p, as, T, sample, in_f64);
            const dim3 g(v->grid), b(kBlock);
            const bool next = v->cfg.autoreset_mode == MI_AUTORESET_NEXT_STEP;
            if (next && sample)
                hipLaunchKernelGGL((mj_rollout_kernel<E, MI_AUTORESET_NEXT_STEP, true>), g, b, 0, v->stream, v->d, p, as, T, v->lay.obs_dim, in_f64);
            else if (next)
                hipLaunchKernelGGL((mj_rollout_kernel<E, MI_AUTORESET_NEXT_STEP, false>), g, b, 0, v->stream, v->d, p, as, T, v->lay.obs_dim, in_f64);
            else if (sample)
                hipLaunchKernelGGL((mj_rollout_kernel<E, MI_AUTORESET_SAME_STEP, true>), g, b, 0, v->stream, v->d, p, as, T, v->lay.obs_dim, in_f64);
            else
                hipLaunchKernelGGL((mj_rollout_kernel<E, MI_AUTORESET_SAME_STEP, false>), g, b, 0, v->stream, v->d, p, as, T, v->lay.obs_dim, in_f64);
            HIP_TRY(hipGetLastError());
            return (int)MI_OK;
        });
    } else if (v->shared_rng) {
        rc = mi_classic::shared_rollout(v, p, as, T, sample);
    } else {
        rc = mi_classic::rollout(v, p, as, T, sample, in_kind);
    }
    if (rc) return rc;
    if (sample) {  // the host copy of the generator moves past the T*N draws the kernel consumes
        const PcgJump j = pcg_jump(v->act_rng.inc, (u128)T * (u128)v->cfg.num_envs * (u128)v->lay.act_dim);
        v->act_rng.state = j.mult * v->act_rng.state + j.plus;
        v->act_lane_valid = false;
    }
    return MI_OK;
}

int mi_action_sample(mi_vecenv *v, int T, void *out, int loc) {
    if (!v) return fail(MI_ERR_INVALID_ARGUMENT, "null env");
    if (T < 0 || (T > 0 && !out)) return fail(MI_ERR_INVALID_ARGUMENT, "mi_action_sample: T >= 0 batches into a non-null array");
    if (!v->act_seeded) return fail(MI_ERR_STATE, "mi_action_sample needs mi_action_seed");
    if (set_device(v)) return MI_ERR_HIP;
    if (T == 0) return action_prepare_lanes(v);
    if (loc == MI_DEVICE) return action_sample_device(v, T, out);
    const size_t bytes = (size_t)T * v->act_bytes;
    if (bytes > v->act_stage_bytes) {
        HIP_TRY(hipStreamSynchronize(v->stream));
        if (v->d_act_stage) (void)hipFree(v->d_act_stage);
        v->d_act_stage = nullptr, v->act_stage_bytes = 0;
        HIP_TRY(hipMalloc(&v->d_act_stage, bytes));
        v->act_stage_bytes = bytes;
    }
    if (int rc = action_sample_device(v, T, v->d_act_stage)) return rc;
    HIP_TRY(hipMemcpyAsync(out, v->d_act_stage, bytes, hipMemcpyDeviceToHost, v->stream));
    HIP_TRY(hipStreamSynchronize(v->stream));
    return MI_OK;
}

int mi_action_get(mi_vecenv *v, uint64_t pcg[4]) {
    if (!v || !pcg) return fail(MI_ERR_INVALID_ARGUMENT, "null argument");
    if (!v->act_seeded) return fail(MI_ERR_STATE, "mi_action_get needs mi_action_seed");
    if (set_device(v)) return MI_ERR_HIP;
    if (int rc = action_sync_host(v)) return rc;
    pcg[0] = (uint64_t)(v->act_rng.state >> 64), pcg[1] = (uint64_t)v->act_rng.state;
    pcg[2] = (uint64_t)(v->act_rng.inc >> 64), pcg[3] = (uint64_t)v->act_rng.inc;
    return MI_OK;
}

int mi_action_skip(mi_vecenv *v, int64_t draws) {
    if (!v) return fail(MI_ERR_INVALID_ARGUMENT, "null env");
    if (!v->act_seeded) return fail(MI_ERR_STATE, "mi_action_skip needs mi_action_seed");
    if (set_device(v)) return MI_ERR_HIP;
    if (int rc = action_sync_host(v)) return rc;
    if (draws == 0) return MI_OK;
    // backwards = forwards by 2^128 - n: the generator's period
    const u128 delta = draws > 0 ? (u128)(uint64_t)draws : (u128)0 - (u128)((uint64_t)0 - (uint64_t)draws);
    const PcgJump j = pcg_jump(v->act_rng.inc, delta);
    v->act_rng.state = j.mult * v->act_rng.state + j.plus;
    v->act_lane_valid = false;
    return MI_OK;
}

int mi_get_stats(mi_vecenv *v, mi_stats *out) {
    if (!v || !out) return fail(MI_ERR_INVALID_ARGUMENT, "null argument");
    if (set_device(v)) return MI_ERR_HIP;
    std::vector<uint64_t> c(4 * (size_t)v->grid);
    std::vector<double> r((size_t)v->grid);
    HIP_TRY(hipMemcpyAsync(c.data(), v->d.blk_count, sizeof(uint64_t) * c.size(), hipMemcpyDeviceToHost, v->stream));
    HIP_TRY(hipMemcpyAsync(r.data(), v->d.blk_ret, sizeof(double) * r.size(), hipMemcpyDeviceToHost, v->stream));
    if (int rc = check_device_error(v)) return rc;
    memset(out, 0, sizeof *out);
    for (int b = 0; b < v->grid; b++) {
        out->env_steps += c[4 * b], out->reset_steps += c[4 * b + 1], out->episodes += c[4 * b + 2];
        out->length_sum += c[4 * b + 3], out->return_sum += r[b];
    }
    return MI_OK;
}

int mi_reset_stats(mi_vecenv *v) {
    if (!v) return fail(MI_ERR_INVALID_ARGUMENT, "null env");
    if (set_device(v)) return MI_ERR_HIP;
    HIP_TRY(hipMemsetAsync(v->d.blk_count, 0, sizeof(uint64_t) * 4 * v->grid, v->stream));
    HIP_TRY(hipMemsetAsync(v->d.blk_ret, 0, sizeof(double) * v->grid, v->stream));
    return MI_OK;
}

int mi_get_state(mi_vecenv *v, double *state, int32_t *elapsed, uint8_t *flags) {
    if (!v) return fail(MI_ERR_INVALID_ARGUMENT, "null env");
    if (set_device(v)) return MI_ERR_HIP;
    const size_t N = (size_t)v->cfg.num_envs, S = (size_t)v->lay.state_dim;
    std::vector<double> soa(S * N);
    std::vector<uint32_t> meta(N);
    HIP_TRY(hipMemcpyAsync(soa.data(), v->d.state, sizeof(double) * S * N, hipMemcpyDeviceToHost, v->stream));
    HIP_TRY(hipMemcpyAsync(meta.data(), v->d.meta, sizeof(uint32_t) * N, hipMemcpyDeviceToHost, v->stream));
    HIP_TRY(hipStreamSynchronize(v->stream));
    for (size_t i = 0; i < N; i++) {
        if (state)
            for (size_t k = 0; k < S; k++) state[i * S + k] = soa[k * N + i];
        if (elapsed) elapsed[i] = (int32_t)(meta[i] & kElapsedMask);
        if (flags) flags[i] = (uint8_t)(meta[i] >> kFlagShift);
    }
    return MI_OK;
}

int mi_set_state(mi_vecenv *v, const double *state, const int32_t *elapsed, const uint8_t *flags) {
    if (!v) return fail(MI_ERR_INVALID_ARGUMENT, "null env");
    if (set_device(v)) return MI_ERR_HIP;
    const size_t N = (size_t)v->cfg.num_envs, S = (size_t)v->lay.state_dim;
    if (state) {
        std::vector<double> soa(S * N);
        for (size_t i = 0; i < N; i++)
            for (size_t k = 0; k < S; k++) soa[k * N + i] = state[i * S + k];
        HIP_TRY(hipMemcpyAsync(v->d.state, soa.data(), sizeof(double) * S * N, hipMemcpyHostToDevice, v->stream));
        HIP_TRY(hipStreamSynchronize(v->stream));
    }
    if (elapsed || flags) {
        std::vector<uint32_t> meta(N);
        HIP_TRY(hipMemcpyAsync(meta.data(), v->d.meta, sizeof(uint32_t) * N, hipMemcpyDeviceToHost, v->stream));
        HIP_TRY(hipStreamSynchronize(v->stream));
        for (size_t i = 0; i < N; i++) {
            uint32_t e = meta[i] & kElapsedMask, f = meta[i] >> kFlagShift;
            if (elapsed) e = (uint32_t)elapsed[i] & kElapsedMask;
            if (flags) f = flags[i] & 3u;
            meta[i] = e | (f << kFlagShift);
        }
        HIP_TRY(hipMemcpyAsync(v->d.meta, meta.data(), sizeof(uint32_t) * N, hipMemcpyHostToDevice, v->stream));
        HIP_TRY(hipStreamSynchronize(v->stream));
        if (v->shared_rng && flags)  // the per-workgroup counts of finished sub-environments follow the new flag words
            if (int rc = mi_classic::shared_recount(v)) return rc;
    }
    v->was_reset = true;
    return MI_OK;
}

}  // extern "C"
#pragma GCC visibility pop
#endif  // MI_CLASSIC_TU
