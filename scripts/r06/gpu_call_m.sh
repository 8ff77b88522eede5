#!/bin/bash
# round 6, call M: the two-role rollout with a barrier per PAIR of wavefronts (duo_pair_sync) instead of __syncthreads(): parity, then A/B against commit fd84f09's build
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
L=gymnasium_amd/csrc/libmi355env
timeout 900 python -m pytest tests/test_gpu_rollout_roles.py tests/test_gpu_device_policy.py -x -q -m gpu 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
timeout 1200 python scripts/ab_bench.py --libs h=${L}_h.so pairsync=${L}.so --envs CartPole-v1:65536:128 MountainCarContinuous-v0:65536:128 Pendulum-v1:65536:128 MountainCar-v0:65536:128 CartPole-v1:262144:128 --rounds 3 --out gpurun_out/r06_duo_pairsync_ab.txt
