// TEST INFRASTRUCTURE ONLY -- CPU emulation of the cooperative MuJoCo kernel (gymnasium_amd/csrc/mjx_coop.h).
//
// The device source is compiled for the host with MJX_HOST_EMU: every lane of a group becomes a fiber (ucontext), and
// coop_sync() yields round-robin to the next lane, which is a barrier as long as all lanes execute the same sequence of
// coop_sync() calls (they do: the kernel's control flow is group-uniform).  This lets tests/test_coop_emu.py compare the
// cooperative algorithm with the C oracle (oracle/mujoco_core.c) without a GPU.  Nothing in the product links this file.
#define MJX_HOST_EMU 1
#include "../../gymnasium_amd/csrc/mjx_coop.h"

#include <ucontext.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

namespace {
constexpr int kMaxLanes = 32;
constexpr size_t kStack = 1 << 20;
ucontext_t g_main, g_ctx[kMaxLanes];
bool g_done[kMaxLanes];
int g_cur = 0, g_n = 0;
long g_syncs = 0;
std::function<void(int)> g_body;
// RACE DETECTOR.  Between two coop_sync() calls the fibers run one after the other, in the order g_perm: ascending (mode 0, the default),
// descending (1), or a fresh random permutation for every interval (2).  The kernel's discipline is that inside one interval no
// blackboard word is written by one lane and read (or written) by another; on the GPU a violation reads old or new data depending on how
// the compiler interleaved the two accesses -- the signature of "results change with the instruction scheduler".  Here a violation makes the
// result depend on the fiber order, so tests/test_coop_emu.py::test_no_cross_lane_dependency_inside_a_sync_interval runs every robot under
// all three orders and requires bit-identical states.
int g_mode = 0, g_perm[kMaxLanes], g_pos = 0;
unsigned long long g_rng = 0x9e3779b97f4a7c15ull;
void make_perm(int n) {
    for (int k = 0; k < n; k++) g_perm[k] = g_mode == 1 ? n - 1 - k : k;
    if (g_mode == 2)
        for (int k = n - 1; k > 0; k--) {
            g_rng ^= g_rng << 13, g_rng ^= g_rng >> 7, g_rng ^= g_rng << 17;
            const int j = (int)(g_rng % (unsigned long long)(k + 1)), t = g_perm[k];
            g_perm[k] = g_perm[j], g_perm[j] = t;
        }
}

void trampoline(int lane) {
    g_body(lane);
    g_done[lane] = true;
    for (int k = 1; k <= g_n; k++) {
        const int nxt = (lane + k) % g_n;
        if (!g_done[nxt]) {
            g_cur = nxt;
            setcontext(&g_ctx[nxt]);
        }
    }
    setcontext(&g_main);
}

void run_group(int n, std::function<void(int)> body) {
    static std::vector<char> stacks;
    stacks.resize(kStack * kMaxLanes);
    g_body = std::move(body), g_n = n;
    for (int l = 0; l < n; l++) {
        g_done[l] = false;
        getcontext(&g_ctx[l]);
        g_ctx[l].uc_stack.ss_sp = stacks.data() + kStack * l, g_ctx[l].uc_stack.ss_size = kStack, g_ctx[l].uc_link = nullptr;
        makecontext(&g_ctx[l], (void (*)())trampoline, 1, l);
    }
    make_perm(n);
    g_pos = 0, g_cur = g_perm[0];
    swapcontext(&g_main, &g_ctx[g_cur]);
}
}  // namespace

long mjx::coop::g_stat[8] = {0};
void mjx::coop::coop_sync() {
    g_syncs++;
    const int me = g_cur;
    if (++g_pos == g_n) {  // every lane has run this interval: the next one starts, in a new order if the mode asks for it
        g_pos = 0;
        if (g_mode == 2) make_perm(g_n);
    }
    const int nxt = g_perm[g_pos];
    g_cur = nxt;
    if (nxt != me) swapcontext(&g_ctx[me], &g_ctx[nxt]);
}

namespace {
using namespace mjx;

template <class M, int G, bool PGS = (M::SOLVER == 1)>
int emu_step(const double *qpos, const double *qvel, const double *ctrl, int nsub, double *qpos_out, double *qvel_out, double *extras,
             double *debug, double *warm) {
    typedef coop::Sim<M, G, PGS> S;
    auto *bb = new typename S::B();
    std::memset((void *)bb, 0, sizeof(*bb));
    g_syncs = 0;
    run_group(G, [&](int lane) {
        typename S::R r;
        std::memset((void *)&r, 0, sizeof(r));
        r.grp = 0;
        if (warm && lane < M::NV) r.warm = warm[lane];
        S::init(*bb, lane);
        for (int k = lane; k < M::NQ; k += G) bb->qpos[k] = qpos[k];
        for (int k = lane; k < M::NV; k += G) bb->qvel[k] = qvel[k];
        for (int k = lane; k < M::NU; k += G) bb->ctrl[k] = ctrl[k];
        coop::coop_sync();
        if (nsub == 0) {
            S::forward(*bb, r, lane);
        } else {
            for (int s = 0; s < nsub; s++) S::step(*bb, r, lane);
        }
        coop::coop_sync();
        S::write_extras(*bb, r, lane, extras);
        if (warm && lane < M::NV) warm[lane] = r.warm;
        if (debug && lane < M::NV) {  // qacc, qacc_smooth, bias, qfrc_constraint, then the mass-matrix rows
            debug[lane] = r.qacc, debug[M::NV + lane] = r.qacc_smooth, debug[2 * M::NV + lane] = r.bias, debug[3 * M::NV + lane] = r.qfrc_constraint;
            for (int j = 0; j < M::NV; j++) debug[4 * M::NV + lane * M::NV + j] = S::mrow(*bb, r, lane, j);
        }
        coop::coop_sync();
    });
    for (int k = 0; k < M::NQ; k++) qpos_out[k] = bb->qpos[k];
    for (int k = 0; k < M::NV; k++) qvel_out[k] = bb->qvel[k];
    const int ncon = bb->ncon;
    delete bb;
    return ncon;
}
// The ONE-LANE simulator (mjx_core.h), compiled for the host as it is: nsub sub-steps (0: one forward pass), qacc and the contact count out.
template <class M>
int core_step(const double *qpos, const double *qvel, const double *ctrl, int nsub, double *qpos_out, double *qvel_out, double *qacc_out,
              double *warm, double *contacts = nullptr) {
    mjx::Data<M> *d = new mjx::Data<M>;
    for (int k = 0; k < M::NQ; k++) d->qpos[k] = qpos[k];
    for (int k = 0; k < M::NV; k++) d->qvel[k] = qvel[k], d->qacc_warm[k] = warm ? warm[k] : 0.0;
    for (int k = 0; k < M::NU; k++) d->ctrl[k] = ctrl[k];
    if (nsub == 0) mjx::forward<M>(*d);
    for (int f = 0; f < nsub; f++) mjx::step<M>(*d);
    for (int k = 0; k < M::NQ; k++) qpos_out[k] = d->qpos[k];
    for (int k = 0; k < M::NV; k++) qvel_out[k] = d->qvel[k], qacc_out[k] = d->qacc[k];
    if (warm)
        for (int k = 0; k < M::NV; k++) warm[k] = d->qacc_warm[k];
    const int ncon = d->ncon;
    if (contacts)  // rows of 17: dist, pos[3], frame[9], pair, D, aref, normal force
        for (int c = 0; c < ncon; c++) {
            double *o = contacts + 17 * c;
            o[0] = d->con[c].dist;
            for (int k = 0; k < 3; k++) o[1 + k] = d->con[c].pos[k];
            for (int k = 0; k < 9; k++) o[4 + k] = d->con[c].frame[k];
            o[13] = d->con[c].pair, o[14] = d->con_D[c], o[15] = d->con_aref[c], o[16] = d->con_force[c][0];
        }
    delete d;
    return ncon;
}
// the one-lane simulator with the opt-in Newton solver for a model whose MJCF asks for PGS
template <class M>
int core_step_newton(const double *qpos, const double *qvel, const double *ctrl, int nsub, double *qpos_out, double *qvel_out, double *qacc_out, double *warm) {
    mjx::Data<M> *d = new mjx::Data<M>;
    for (int k = 0; k < M::NQ; k++) d->qpos[k] = qpos[k];
    for (int k = 0; k < M::NV; k++) d->qvel[k] = qvel[k], d->qacc_warm[k] = warm ? warm[k] : 0.0;
    for (int k = 0; k < M::NU; k++) d->ctrl[k] = ctrl[k];
    if (nsub == 0) mjx::forward<M, false>(*d);
    for (int f = 0; f < nsub; f++) mjx::step<M, false>(*d);
    for (int k = 0; k < M::NQ; k++) qpos_out[k] = d->qpos[k];
    for (int k = 0; k < M::NV; k++) qvel_out[k] = d->qvel[k], qacc_out[k] = d->qacc[k];
    if (warm)
        for (int k = 0; k < M::NV; k++) warm[k] = d->qacc_warm[k];
    const int ncon = d->ncon;
    delete d;
    return ncon;
}
}  // namespace

extern "C" {
// the one-lane simulator; model ids follow oracle/mujoco_envs.h (0 half_cheetah .. 10 pusher)
__attribute__((visibility("default"))) int core_emu_step(int model, const double *qpos, const double *qvel, const double *ctrl, int nsub,
                                                          double *qpos_out, double *qvel_out, double *qacc_out, double *warm, double *contacts) {
    switch (model) {
        case 0: return core_step<HalfCheetahModel>(qpos, qvel, ctrl, nsub, qpos_out, qvel_out, qacc_out, warm, contacts);
        case 1: return core_step<AntModel>(qpos, qvel, ctrl, nsub, qpos_out, qvel_out, qacc_out, warm, contacts);
        case 2: return core_step<HumanoidModel>(qpos, qvel, ctrl, nsub, qpos_out, qvel_out, qacc_out, warm, contacts);
        case 3: return core_step<HopperModel>(qpos, qvel, ctrl, nsub, qpos_out, qvel_out, qacc_out, warm, contacts);
        case 4: return core_step<Walker2dModel>(qpos, qvel, ctrl, nsub, qpos_out, qvel_out, qacc_out, warm, contacts);
        case 5: return core_step<InvertedPendulumModel>(qpos, qvel, ctrl, nsub, qpos_out, qvel_out, qacc_out, warm, contacts);
        case 6: return core_step<InvertedDoublePendulumModel>(qpos, qvel, ctrl, nsub, qpos_out, qvel_out, qacc_out, warm, contacts);
        case 7: return core_step<ReacherModel>(qpos, qvel, ctrl, nsub, qpos_out, qvel_out, qacc_out, warm, contacts);
        case 8: return core_step<HumanoidStandupModel>(qpos, qvel, ctrl, nsub, qpos_out, qvel_out, qacc_out, warm, contacts);
        case 9: return core_step<SwimmerModel>(qpos, qvel, ctrl, nsub, qpos_out, qvel_out, qacc_out, warm, contacts);
        case 10: return core_step<PusherModel>(qpos, qvel, ctrl, nsub, qpos_out, qvel_out, qacc_out, warm, contacts);
    }
    return -1;
}
// model: 0 half_cheetah, 1 ant, 2 humanoid.  nsub = 0: one forward pass only.  Returns the number of contacts of the last pass.
__attribute__((visibility("default"))) int coop_emu_step(int model, const double *qpos, const double *qvel, const double *ctrl, int nsub,
                                                          double *qpos_out, double *qvel_out, double *extras, double *debug, double *warm) {
    switch (model) {  // warm: in/out qacc_warmstart[nv] (NULL = start from zero)
        case 0: return emu_step<HalfCheetahModel, 16>(qpos, qvel, ctrl, nsub, qpos_out, qvel_out, extras, debug, warm);
        case 1: return emu_step<AntModel, 16>(qpos, qvel, ctrl, nsub, qpos_out, qvel_out, extras, debug, warm);
        case 3: return emu_step<HopperModel, 16>(qpos, qvel, ctrl, nsub, qpos_out, qvel_out, extras, debug, warm);
        case 4: return emu_step<Walker2dModel, 16>(qpos, qvel, ctrl, nsub, qpos_out, qvel_out, extras, debug, warm);
        case 2: return emu_step<HumanoidModel, 32>(qpos, qvel, ctrl, nsub, qpos_out, qvel_out, extras, debug, warm);  // the MJCF's solver: PGS / 50
        case 12: return emu_step<HumanoidModel, 32, false>(qpos, qvel, ctrl, nsub, qpos_out, qvel_out, extras, debug, warm);  // opt-in Newton
        case 8: return emu_step<HumanoidStandupModel, 32>(qpos, qvel, ctrl, nsub, qpos_out, qvel_out, extras, debug, warm);  // lying on the floor: many contacts
        case 18: return emu_step<HumanoidStandupModel, 32, false>(qpos, qvel, ctrl, nsub, qpos_out, qvel_out, extras, debug, warm);
    }
    return -1;
}
__attribute__((visibility("default"))) int core_emu_step_newton(int model, const double *qpos, const double *qvel, const double *ctrl, int nsub,
                                                                 double *qpos_out, double *qvel_out, double *qacc_out, double *warm) {
    switch (model) {
        case 2: return core_step_newton<HumanoidModel>(qpos, qvel, ctrl, nsub, qpos_out, qvel_out, qacc_out, warm);
        case 8: return core_step_newton<HumanoidStandupModel>(qpos, qvel, ctrl, nsub, qpos_out, qvel_out, qacc_out, warm);
    }
    return -1;
}
__attribute__((visibility("default"))) int coop_emu_extras_dim(int model) {
    switch (model) {
        case 0: return coop::Sim<HalfCheetahModel, 16>::EX_TOTAL;
        case 1: return coop::Sim<AntModel, 16>::EX_TOTAL;
        case 3: return coop::Sim<HopperModel, 16>::EX_TOTAL;
        case 4: return coop::Sim<Walker2dModel, 16>::EX_TOTAL;
        case 2: return coop::Sim<HumanoidModel, 32>::EX_TOTAL;
        case 8: return coop::Sim<HumanoidStandupModel, 32>::EX_TOTAL;
    }
    return -1;
}
__attribute__((visibility("default"))) long coop_emu_board_bytes(int model) {
    switch (model) {
        case 0: return sizeof(coop::Board<HalfCheetahModel, 16>);
        case 1: return sizeof(coop::Board<AntModel, 16>);
        case 3: return sizeof(coop::Board<HopperModel, 16>);
        case 4: return sizeof(coop::Board<Walker2dModel, 16>);
        case 2: return sizeof(coop::Board<HumanoidModel, 32>);
    }
    return -1;
}
__attribute__((visibility("default"))) long coop_emu_last_syncs() { return g_syncs; }
// order in which the lanes of a group run inside one sync interval: 0 ascending, 1 descending, 2 a fresh random permutation per interval
__attribute__((visibility("default"))) void coop_emu_set_order(int mode, unsigned long long seed) { g_mode = mode, g_rng = seed | 1ull; }
__attribute__((visibility("default"))) void coop_emu_stats(long *out, int reset) {
    for (int k = 0; k < 8; k++) out[k] = mjx::coop::g_stat[k];
    if (reset)
        for (int k = 0; k < 8; k++) mjx::coop::g_stat[k] = 0;
}
}
