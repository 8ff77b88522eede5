// physics32.hip -- cooperative physics kernels of the robots that use 32 lanes per sub-environment (mjx_coop.h, G = 32).
// Compiled with -mllvm -amdgpu-sched-strategy=iterative-maxocc (build.py TU_FLAGS): +34 % on Humanoid-v5, results bit-identical to the
// default scheduler's (scripts/coop_phase_bench.hip fingerprints, tests/test_gpu_mujoco.py).
#include "mjx_physics.h"

namespace mi_phys {
bool launch32(int kind, const Args &a, bool skip_resetting, const void *actions, double *extras, hipStream_t stream) {
    switch (kind) {
    case MI_ENV_HUMANOID: launch_kind<mjx::MjEnv<mjx::HumanoidModel, mjx::kHumanoid>>(a, skip_resetting, actions, extras, stream); return true;
    case MI_ENV_HUMANOID_STANDUP:
        launch_kind<mjx::MjEnv<mjx::HumanoidStandupModel, mjx::kHumanoidStandup>>(a, skip_resetting, actions, extras, stream);
        return true;
    }
    return false;
}
}  // namespace mi_phys
