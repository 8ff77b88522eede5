#!/bin/bash
# round 6, call AL: the two-role rollout with one phase loop per ROLE (instead of one loop holding both roles' code): classic parity, A/B on the four environments that use it
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
L=gymnasium_amd/csrc/libmi355env
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_contract.py tests/test_gpu_device_policy.py tests/test_gpu_rollout_roles.py tests/test_gpu_float64_actions.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python scripts/ab_bench.py --libs one_loop=${L}_oneloop.so loop_per_role=${L}.so --envs CartPole-v1:65536:128 Pendulum-v1:65536:128 MountainCar-v0:65536:128 MountainCarContinuous-v0:65536:128 CartPole-v1:262144:128 --rounds 4 --out gpurun_out/r06_duo_loop_per_role_ab.txt
