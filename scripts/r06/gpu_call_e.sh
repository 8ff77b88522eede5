#!/bin/bash
# round 6, call E: aux role derives observation + flags from the state words (CartPole, MountainCar), chunk 8; why the MuJoCo window check disagrees
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_rollout_roles.py tests/test_gpu_parity.py tests/test_gpu_bench_contract.py -x -q -m gpu > gpurun_out/r06_e_tests.log 2>&1; tail -3 gpurun_out/r06_e_tests.log
timeout 1500 python scripts/ab_bench.py --libs base=gymnasium_amd/csrc/libmi355env_base.so c8=gymnasium_amd/csrc/libmi355env_c8.so derive=gymnasium_amd/csrc/libmi355env.so \
   --envs CartPole-v1:65536:128 MountainCar-v0:65536:128 MountainCarContinuous-v0:65536:128 --rounds 3 --out gpurun_out/r06_duo_diet_ab.txt
timeout 300 python scripts/r06/dbg_window.py Ant-v5 2>&1 | grep -v Warn | tail -14
timeout 300 python scripts/r06/dbg_window.py Humanoid-v5 2>&1 | grep -v Warn | tail -14
