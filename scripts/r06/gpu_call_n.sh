#!/bin/bash
# round 6, call N: Pendulum's angle_normalize on the env role (the aux role was the longer one), MountainCar's episode length from the aux role's TimeLimit counter: parity, A/B against commit fd84f09's build
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
L=gymnasium_amd/csrc/libmi355env
timeout 900 python -m pytest tests/test_gpu_rollout_roles.py tests/test_gpu_device_policy.py -x -q -m gpu 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
timeout 1200 python scripts/ab_bench.py --libs h=${L}_h.so new=${L}.so --envs Pendulum-v1:65536:128 MountainCar-v0:65536:128 CartPole-v1:65536:128 Pendulum-v1:262144:128 --rounds 3 --out gpurun_out/r06_pendulum_second_cut_ab.txt
