// physics16.hip -- cooperative physics kernels of the robots that use 16 lanes per sub-environment (mjx_coop.h, G = 16).
// Compiled with -mllvm -amdgpu-sched-strategy=iterative-minreg (build.py TU_FLAGS): +10 % on Ant-v5, results bit-identical to the default
// scheduler's.  NOT iterative-maxocc / -ilp: those are faster still (+19 %) but produce wrong RK4 stage updates for this instantiation
// (found by tests/test_gpu_mujoco.py, narrowed down with scripts/coop_phase_bench.hip).
#include "mjx_physics.h"

namespace mi_phys {
bool launch16(int kind, const Args &a, bool skip_resetting, const float *actions, double *extras, hipStream_t stream) {
    switch (kind) {
    case MI_ENV_ANT: launch_kind<mjx::MjEnv<mjx::AntModel, mjx::kAnt>>(a, skip_resetting, actions, extras, stream); return true;
    case MI_ENV_HALF_CHEETAH: launch_kind<mjx::MjEnv<mjx::HalfCheetahModel, mjx::kHalfCheetah>>(a, skip_resetting, actions, extras, stream); return true;
    }
    return false;
}
}  // namespace mi_phys
