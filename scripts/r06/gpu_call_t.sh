#!/bin/bash
# round 6, call T: the env role requests the next step's action one step ahead (ring padded by one row): parity, A/B against commit 485946b's build
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
L=gymnasium_amd/csrc/libmi355env
timeout 900 python -m pytest tests/test_gpu_rollout_roles.py tests/test_gpu_device_policy.py -x -q -m gpu 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
timeout 1200 python scripts/ab_bench.py --libs h=${L}_h.so prefetch=${L}.so --envs CartPole-v1:65536:128 MountainCar-v0:65536:128 MountainCarContinuous-v0:65536:128 Pendulum-v1:65536:128 --rounds 3 --out gpurun_out/r06_act_prefetch_ab.txt
