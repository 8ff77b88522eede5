// physics16.hip -- cooperative physics kernels of the robots that use 16 lanes per sub-environment (mjx_coop.h, G = 16).
// Compiled with -mllvm -amdgpu-sched-strategy=iterative-maxocc (build.py TU_FLAGS): +19 % on Ant-v5, results bit-identical to the default
// scheduler's (scripts/coop_phase_bench.hip fingerprints, tests/test_gpu_mujoco.py) as long as mjx_coop.h rk4_stage() is not inlined.
#include "mjx_physics.h"

namespace mi_phys {
bool launch16(int kind, const Args &a, bool skip_resetting, const void *actions, double *extras, hipStream_t stream) {
    switch (kind) {
    case MI_ENV_ANT: launch_kind<mjx::MjEnv<mjx::AntModel, mjx::kAnt>>(a, skip_resetting, actions, extras, stream); return true;
    case MI_ENV_HALF_CHEETAH: launch_kind<mjx::MjEnv<mjx::HalfCheetahModel, mjx::kHalfCheetah>>(a, skip_resetting, actions, extras, stream); return true;
    case MI_ENV_HOPPER: launch_kind<mjx::MjEnv<mjx::HopperModel, mjx::kHopper>>(a, skip_resetting, actions, extras, stream); return true;
    case MI_ENV_WALKER2D: launch_kind<mjx::MjEnv<mjx::Walker2dModel, mjx::kWalker2d>>(a, skip_resetting, actions, extras, stream); return true;
    }
    return false;
}
}  // namespace mi_phys
