#!/bin/bash
# round 6, call AE: Pendulum's two float64 squares: the env role takes the plain products, the aux role runs the pow routine once per pending argument of a chunk: parity, A/B
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
L=gymnasium_amd/csrc/libmi355env
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_contract.py tests/test_gpu_device_policy.py tests/test_gpu_rollout_roles.py tests/test_gpu_float64_actions.py -x -q -m gpu -k "Pendulum or pendulum or Acrobot or acrobot or digest or role" 2>&1 | tail -4
timeout 900 python scripts/ab_bench.py --libs before=${L}_h.so grouped=${L}.so --envs Pendulum-v1:65536:128 Pendulum-v1:262144:128 Acrobot-v1:65536:128 MountainCarContinuous-v0:65536:128 --rounds 3 --out gpurun_out/r06_pendulum_grouped_pow_ab.txt
