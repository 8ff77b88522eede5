// coop_phase_bench.hip -- standalone timing harness for the cooperative MuJoCo kernel (gymnasium_amd/csrc/mjx_coop.h).
// Not part of the library: it compiles the same header with MJX_PHASE_TIMING, advances N robots from perturbed initial states
// for a few env-steps (so that contacts and joint limits are active) and prints, per phase of forward(), the share of shader cycles
// (s_memtime deltas summed over wavefronts).  Build + run (on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Igymnasium_amd/csrc scripts/coop_phase_bench.hip -o gpurun_out/coop_phase_bench
//   gpurun_out/coop_phase_bench [ant|humanoid|standup|cheetah] [num_envs]
#ifndef NO_PHASE_TIMING
#define MJX_PHASE_TIMING 1
#endif
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#include "mjx_coop.h"

using namespace mjx;

// the finer marks of the phase selected with -DMJX_DETAIL=<tag> (mjx_coop.h MJX_PHASE_X); the rest of that phase stays in its own slot
#if MJX_DETAIL == 1
#define DETAIL12 "pgs: actuation, passive, M row to registers"
#define DETAIL13 "pgs: Cholesky factor of M"
#elif MJX_DETAIL == 2
#define DETAIL12 "crb: composite inertias"
#define DETAIL13 "crb: I * cdof"
#elif MJX_DETAIL == 3
#define DETAIL12 "RNE: velocity prefix"
#define DETAIL13 "RNE: joint chain, acceleration prefix, body forces"
#elif MJX_DETAIL == 4
#define DETAIL12 "kinematics: joint chains"
#define DETAIL13 "kinematics: pointer-jumping rounds"
#else
#define DETAIL12 "detail 12"
#define DETAIL13 "detail 13"
#endif

template <class M, int G, bool PGS>
__global__ __launch_bounds__(64) void phys(double *state, const float *actions, int N, int nsub, unsigned long long *phase) {
    typedef coop::Sim<M, G, PGS> S;
    constexpr int EPW = 64 / G;
    __shared__ typename S::B boards[EPW];
    const int grp = threadIdx.x / G, lane = threadIdx.x % G;
    const int env = blockIdx.x * EPW + grp;
    if (env >= N) return;
    typename S::B &bb = boards[grp];
    typename S::R r;
#ifdef MJX_PHASE_TIMING
    for (int k = 0; k < 16; k++) r.tphase[k] = 0;
#endif
    S::init(bb, lane);
    for (int k = lane; k < M::NQ; k += G) bb.qpos[k] = state[(size_t)k * N + env];
    for (int k = lane; k < M::NV; k += G) bb.qvel[k] = state[(size_t)(M::NQ + k) * N + env];
    for (int k = lane; k < M::NU; k += G) bb.ctrl[k] = (double)actions[(size_t)env * M::NU + k];
    r.warm = lane < M::NV ? state[(size_t)(M::NQ + M::NV + lane) * N + env] : 0.0;
    r.grp = grp;
#ifdef MJX_COUNT_WORK
    r.work = 0, r.work_wave = 0;
#endif
    coop::coop_sync();
#ifdef MJX_PHASE_TIMING
    r.tmark = __builtin_readcyclecounter();
#endif
    for (int s = 0; s < nsub; s++) S::step(bb, r, lane);
    coop::coop_sync();
    for (int k = lane; k < M::NQ; k += G) state[(size_t)k * N + env] = bb.qpos[k];
    for (int k = lane; k < M::NV; k += G) state[(size_t)(M::NQ + k) * N + env] = bb.qvel[k];
    if (lane < M::NV) state[(size_t)(M::NQ + M::NV + lane) * N + env] = r.warm;
#ifdef MJX_COUNT_WORK
    if (lane == 0) ((int *)phase)[32 + env] = r.work, atomicAdd((unsigned long long *)phase + 14, (unsigned long long)r.work),
        atomicAdd((unsigned long long *)phase + 15, (unsigned long long)r.work_wave);  // per-env passes (buffer sized in run()) + totals
#endif
#ifdef MJX_PHASE_TIMING
    if (threadIdx.x == 0)
        for (int k = 0; k < 16; k++) atomicAdd(&phase[k], r.tphase[k]);
#endif
}

// COOP_DEBUG: ONE forward pass from the initial state; per env 4 NV + NV^2 doubles: qacc, qacc_smooth, bias, qfrc_constraint, mass-matrix rows
template <class M, int G>
__global__ __launch_bounds__(64) void fwd_debug(const double *state, const float *actions, int N, double *out) {
    typedef coop::Sim<M, G> S;
    constexpr int EPW = 64 / G, NV = M::NV;
    __shared__ typename S::B boards[EPW];
    const int grp = threadIdx.x / G, lane = threadIdx.x % G;
    const int env = blockIdx.x * EPW + grp;
    if (env >= N) return;
    typename S::B &bb = boards[grp];
    typename S::R r;
    S::init(bb, lane);
    for (int k = lane; k < M::NQ; k += G) bb.qpos[k] = state[(size_t)k * N + env];
    for (int k = lane; k < M::NV; k += G) bb.qvel[k] = state[(size_t)(M::NQ + k) * N + env];
    for (int k = lane; k < M::NU; k += G) bb.ctrl[k] = (double)actions[(size_t)env * M::NU + k];
    r.warm = 0.0;
    r.grp = grp;
    coop::coop_sync();
    S::forward(bb, r, lane);
    coop::coop_sync();
    double *o = out + (size_t)env * (4 * NV + NV * NV);
    if (lane < NV) {
        o[lane] = r.qacc, o[NV + lane] = r.qacc_smooth, o[2 * NV + lane] = r.bias, o[3 * NV + lane] = r.qfrc_constraint;
        for (int j = 0; j < NV; j++) o[4 * NV + lane * NV + j] = S::mrow(bb, r, lane, j);
    }
}

template <class M, int G, bool PGS = (M::SOLVER == 1)>
int run(int N, int nsub, float amp) {
    const int S = M::NQ + 2 * M::NV;
    std::vector<double> st((size_t)S * N, 0.0);
    std::vector<float> act((size_t)N * M::NU);
    unsigned long long seed = 88172645463325252ull;
    auto rnd = [&]() { seed ^= seed << 13, seed ^= seed >> 7, seed ^= seed << 17; return (double)(seed >> 11) / 9007199254740992.0; };
    for (int i = 0; i < N; i++) {
        for (int k = 0; k < M::NQ; k++) st[(size_t)k * N + i] = M::qpos0[k] + 0.1 * (2 * rnd() - 1) * (k >= 3 && k < 7 ? 0.1 : 1.0);
        for (int k = 0; k < M::NV; k++) st[(size_t)(M::NQ + k) * N + i] = 0.1 * (2 * rnd() - 1);
    }
    double *d_st;
    float *d_act;
    unsigned long long *d_ph;
    hipMalloc(&d_st, sizeof(double) * st.size()), hipMalloc(&d_act, sizeof(float) * act.size()), hipMalloc(&d_ph, 12 * 8 + 256 + sizeof(int) * (size_t)N);
    hipMemcpy(d_st, st.data(), sizeof(double) * st.size(), hipMemcpyHostToDevice);
    const dim3 grid((N + 64 / G - 1) / (64 / G)), block(64);
    hipEvent_t e0, e1, k0, k1;
    hipEventCreate(&e0), hipEventCreate(&e1), hipEventCreate(&k0), hipEventCreate(&k1);
    float kernel_ms = 0;
    if (getenv("COOP_DEBUG")) {
        const size_t W = 4 * M::NV + M::NV * M::NV;
        double *d_out;
        hipMalloc(&d_out, sizeof(double) * W * N);
        hipMemset(d_out, 0, sizeof(double) * W * N);
        for (auto &a : act) a = amp * (float)(2 * rnd() - 1);
        hipMemcpy(d_act, act.data(), sizeof(float) * act.size(), hipMemcpyHostToDevice);
        hipLaunchKernelGGL((fwd_debug<M, G>), grid, block, 0, 0, d_st, d_act, N, d_out);
        std::vector<double> out(W * N);
        hipMemcpy(out.data(), d_out, sizeof(double) * out.size(), hipMemcpyDeviceToHost);
        FILE *f = fopen(getenv("COOP_DEBUG"), "wb");
        fwrite(out.data(), sizeof(double), out.size(), f), fclose(f);
        printf("wrote %zu doubles per env\n", W);
        return 0;
    }
    const int warm = getenv("COOP_WARM") ? atoi(getenv("COOP_WARM")) : 40, timed = getenv("COOP_TIMED") ? atoi(getenv("COOP_TIMED")) : 6;
    if (getenv("COOP_NSUB")) nsub = atoi(getenv("COOP_NSUB"));
    float ms = 0;
    for (int t = 0; t < warm + timed; t++) {
        for (auto &a : act) a = amp * (float)(2 * rnd() - 1);
        hipMemcpy(d_act, act.data(), sizeof(float) * act.size(), hipMemcpyHostToDevice);
        if (t == warm) hipMemset(d_ph, 0, 16 * 8), hipEventRecord(e0);
        hipEventRecord(k0);
        hipLaunchKernelGGL((phys<M, G, PGS>), grid, block, 0, 0, d_st, d_act, N, nsub, d_ph);
        hipEventRecord(k1);
#ifdef MJX_COUNT_WORK
        {  // COOP_SORT=1: regroup the environments between launches by the solver work of the launch just finished (experiment: how much of the
           // waiting inside a wavefront is predictable from one launch to the next).  Environments are independent: permuting the columns is exact.
            float el = 0;
            hipEventSynchronize(k1), hipEventElapsedTime(&el, k0, k1);
            if (t >= warm) kernel_ms += el;
            if (getenv("COOP_SORT")) {
                std::vector<int> w(N), order(N);
                hipMemcpy(w.data(), (const char *)d_ph + 128, sizeof(int) * N, hipMemcpyDeviceToHost);
                hipMemcpy(st.data(), d_st, sizeof(double) * st.size(), hipMemcpyDeviceToHost);
                for (int i = 0; i < N; i++) order[i] = i;
                std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return w[a] < w[b]; });
                std::vector<double> st2(st.size());
                for (int k = 0; k < S; k++)
                    for (int i = 0; i < N; i++) st2[(size_t)k * N + i] = st[(size_t)k * N + order[i]];
                hipMemcpy(d_st, st2.data(), sizeof(double) * st2.size(), hipMemcpyHostToDevice);
            }
        }
#endif
    }
    hipEventRecord(e1), hipEventSynchronize(e1), hipEventElapsedTime(&ms, e0, e1);
    unsigned long long ph[16];
    hipMemcpy(ph, d_ph, sizeof ph, hipMemcpyDeviceToHost);
#ifdef MJX_COUNT_WORK
    {  // how unevenly the solver's work is spread over the sub-environments that share a wavefront (they wait for the slowest)
        std::vector<int> w(N);
        hipMemcpy(w.data(), (const char *)d_ph + 128, sizeof(int) * N, hipMemcpyDeviceToHost);
        const int epw = 64 / G;
        double sum = 0, waves = 0, sorted_waves = 0;
        for (int i = 0; i < N; i++) sum += w[i];
        for (int i = 0; i + epw <= N; i += epw) {
            int m = 0;
            for (int k = 0; k < epw; k++) m = w[i + k] > m ? w[i + k] : m;
            waves += (double)m * epw;
        }
        std::vector<int> s2(w);
        std::sort(s2.begin(), s2.end());
        for (int i = 0; i + epw <= N; i += epw) sorted_waves += (double)s2[i + epw - 1] * epw;
        printf("solver passes per env in the last launch: mean %.2f; wavefront-max grouping costs x%.3f of the mean, sorted grouping x%.3f\n", sum / N,
               waves / sum, sorted_waves / sum);
        unsigned long long tot[2];
        hipMemcpy(tot, (const char *)d_ph + 14 * 8, sizeof tot, hipMemcpyDeviceToHost);
        printf("per forward pass: the slowest sub-environment of a wavefront has x%.3f the solver cost (5 x passes + line-search iterations) of the average one (all launches)\n",
               (double)tot[1] / (double)tot[0]);
        printf("kernel time of the timed launches: %.3f ms per launch%s\n", kernel_ms / timed, getenv("COOP_SORT") ? " (COOP_SORT: regrouped by the previous launch's work)" : "");
    }
#endif
    {  // fingerprint of the final state: lets two builds of this harness (compiler flags, code variants) be compared bit for bit
        hipMemcpy(st.data(), d_st, sizeof(double) * st.size(), hipMemcpyDeviceToHost);
        double sum = 0, asum = 0;
        int bad = 0;
        for (double v : st) sum += v, asum += fabs(v), bad += !(v == v);
        printf("state fingerprint: sum %.17g abs-sum %.17g non-finite %d; env0 qpos[0..3] %.17g %.17g %.17g %.17g\n", sum, asum, bad, st[0], st[(size_t)N], st[(size_t)2 * N],
               st[(size_t)3 * N]);
        if (getenv("COOP_DUMP")) {
            FILE *f = fopen(getenv("COOP_DUMP"), "wb");
            fwrite(st.data(), sizeof(double), st.size(), f), fclose(f);
        }
    }
    const char *newton_names[16] = {"integrator / glue", "kinematics", "com_pos", "collision", "com_vel_and_bias (RNE)", "crb", "make_constraint",
                                    "solver: assemble", "solver: factor + solve", "solver: twist / J dir", "solver: line search", "solver: other", DETAIL12, DETAIL13, "detail 14", "detail 15"};
    const char *pgs_names[16] = {"integrator / glue", "kinematics", "com_pos", "collision", "com_vel_and_bias (RNE)", "crb", "pgs: rows (M^-1 J^T, A, warm start) + make_constraint",
                                 "pgs: M^-1", "pgs: factor M + qacc_smooth", "pgs: J qacc_smooth, J warm", "pgs: sweeps", "other", DETAIL12, DETAIL13, "detail 14", "detail 15"};
    const char **names = PGS ? pgs_names : newton_names;
    double tot = 0;
    for (int k = 0; k < 16; k++) tot += (double)ph[k];
    printf("%d envs, %d sub-steps per launch: %.3f ms per launch (incl. host action upload), %.4g env-steps/s\n", N, nsub, ms / timed, N / (ms / timed * 1e-3));
    const double waves = (double)((N + 64 / G - 1) / (64 / G)) * timed, forwards = waves * nsub * (M::INTEGRATOR ? 4 : 1);
    for (int k = 0; k < 16; k++)
        if (k < 12 || ph[k]) printf("  %-26s %5.1f %%   %8.0f cycles per forward pass\n", names[k], 100.0 * ph[k] / tot, ph[k] / forwards);
    printf("  total %.0f cycles per forward pass and wavefront (%d envs per wavefront)\n", tot / forwards, 64 / G);
    return 0;
}

int main(int argc, char **argv) {
    const char *which = argc > 1 ? argv[1] : "ant";
    const int N = argc > 2 ? atoi(argv[2]) : 32768;
    if (!strcmp(which, "humanoid")) return run<HumanoidModel, 32>(N, 5, 0.4f);              // the MJCF's solver: PGS / 50
    if (!strcmp(which, "humanoid-newton")) return run<HumanoidModel, 32, false>(N, 5, 0.4f);  // opt-in Newton
    if (!strcmp(which, "standup")) return run<HumanoidStandupModel, 32>(N, 5, 0.4f);
    if (!strcmp(which, "standup-newton")) return run<HumanoidStandupModel, 32, false>(N, 5, 0.4f);
    if (!strcmp(which, "cheetah")) return run<HalfCheetahModel, 16>(N, 5, 1.0f);
    return run<AntModel, 16>(N, 5, 1.0f);
}
