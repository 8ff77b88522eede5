#!/bin/bash
# round 6, call P: MountainCarContinuous' reward on the aux role (from the action it drew and the terminated flag): parity, A/B against commit fd84f09's build
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
L=gymnasium_amd/csrc/libmi355env
timeout 900 python -m pytest tests/test_gpu_rollout_roles.py tests/test_gpu_device_policy.py -x -q -m gpu 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
timeout 1200 python scripts/ab_bench.py --libs h=${L}_h.so new=${L}.so --envs MountainCarContinuous-v0:65536:128 Pendulum-v1:65536:128 MountainCarContinuous-v0:262144:128 --rounds 3 --out gpurun_out/r06_mcc_reward_on_aux_ab.txt
