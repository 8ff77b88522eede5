// classic.hip -- the classic-control kernels of engine.hip as their own translation unit (their own instruction scheduler: build.py TU_FLAGS,
// the head of engine.hip says why): engine.hip compiled with MI_CLASSIC_TU defines ONLY namespace mi_classic (step / reset / rollout launchers and
// the kernels they instantiate); the other unit holds everything else and calls these three functions.
#define MI_CLASSIC_TU 1
#include "engine.hip"
