#!/bin/bash
# round 6, call J: the two-role rollout with the pending-autoreset flag as a loop-carried lane mask, the packed flag word (one byte per flag) and the
# episode length taken from the TimeLimit counter: parity (roles, digests), then A/B against the build of commit fd84f09 (libmi355env_h.so)
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_rollout_roles.py -x -q -m gpu > gpurun_out/r06_j_tests.log 2>&1; tail -3 gpurun_out/r06_j_tests.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r06_j_tests2.log 2>&1; tail -3 gpurun_out/r06_j_tests2.log
timeout 900 python scripts/ab_bench.py --libs h=gymnasium_amd/csrc/libmi355env_h.so packed=gymnasium_amd/csrc/libmi355env.so --envs CartPole-v1:65536:128 MountainCar-v0:65536:128 Pendulum-v1:65536:128 MountainCarContinuous-v0:65536:128 --rounds 3 --out gpurun_out/r06_duo_packed_ab.txt
