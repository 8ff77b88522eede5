#!/bin/bash
# round 6, call AH: the trig polynomials' constants from LDS as well (18 doubles), for every exact-math environment but CartPole: classic parity, A/B against the build
# with Acrobot's eight reduction constants only
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
L=gymnasium_amd/csrc/libmi355env
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_contract.py tests/test_gpu_device_policy.py tests/test_gpu_rollout_roles.py tests/test_gpu_float64_actions.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python scripts/ab_bench.py --libs hot8=${L}_h.so hot18_all=${L}.so --envs Acrobot-v1:65536:128 Acrobot-v1:262144:128 Pendulum-v1:65536:128 MountainCar-v0:65536:128 MountainCarContinuous-v0:65536:128 CartPole-v1:65536:128 --rounds 3 --out gpurun_out/r06_hot_constants_all_ab.txt
