#!/bin/bash
# round 6, call AJ: Pendulum's angle_normalize -- fmod's quotient estimate by the rounded reciprocal (1 instruction) instead of the division (11): classic parity, A/B
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
L=gymnasium_amd/csrc/libmi355env
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_contract.py tests/test_gpu_device_policy.py tests/test_gpu_rollout_roles.py tests/test_gpu_float64_actions.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python scripts/ab_bench.py --libs fmod_div=${L}_fmoddiv.so fmod_rcp=${L}.so --envs Pendulum-v1:65536:128 Pendulum-v1:262144:128 --rounds 4 --out gpurun_out/r06_pendulum_fmod_reciprocal_ab.txt
