#!/bin/bash
# round 6, call AK: Pendulum's float32 `u ** 2` on the aux role -- plain-product test per action of a chunk, the powf routine once per pending action: classic parity, A/B
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
L=gymnasium_amd/csrc/libmi355env
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_contract.py tests/test_gpu_device_policy.py tests/test_gpu_rollout_roles.py tests/test_gpu_float64_actions.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python scripts/ab_bench.py --libs powf_each_step=${L}_sqfeach.so powf_grouped=${L}.so --envs Pendulum-v1:65536:128 Pendulum-v1:262144:128 --rounds 4 --out gpurun_out/r06_pendulum_grouped_powf_ab.txt
