#!/bin/bash
# round 6, call K: which of the three changes to the two-role rollout pays -- the loop-carried lane mask (nobool = without), the packed flag word
# (nopacked = without), the limb-wise 128-bit jump (nolimb = without); old3 = none of them (the sources of commit fd84f09 through the same macros); h = that commit's build
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
L=gymnasium_amd/csrc/libmi355env
timeout 600 python -m pytest tests/test_gpu_rollout_roles.py tests/test_gpu_device_policy.py -x -q -m gpu 2>&1 | tail -2
timeout 1200 python scripts/ab_bench.py --libs h=${L}_h.so old3=${L}_old3.so all=${L}.so nopacked=${L}_nopacked.so nobool=${L}_nobool.so nolimb=${L}_nolimb.so --envs CartPole-v1:65536:128 MountainCarContinuous-v0:65536:128 Pendulum-v1:65536:128 MountainCar-v0:65536:128 --rounds 3 --out gpurun_out/r06_duo_three_changes_ab.txt
