#!/bin/bash
# round 6, call AD: Acrobot's three `** 2` per derivative evaluation through ONE pass of the pow routine (pow_exact.h square3): parity, A/B against the build before
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
L=gymnasium_amd/csrc/libmi355env
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_contract.py tests/test_gpu_device_policy.py -x -q -m gpu -k "Acrobot or acrobot or digest" 2>&1 | tail -4
timeout 900 python scripts/ab_bench.py --libs before=${L}_h.so grouped=${L}.so --envs Acrobot-v1:65536:128 Acrobot-v1:262144:128 Pendulum-v1:65536:128 CartPole-v1:65536:128 --rounds 3 --out gpurun_out/r06_acrobot_grouped_pow_ab.txt
