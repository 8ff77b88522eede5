#!/bin/bash
# round 6, call AG: Acrobot's range-reduction constants from LDS (held in vector registers across the loop) instead of float64 literals: parity, A/B
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
L=gymnasium_amd/csrc/libmi355env
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_contract.py tests/test_gpu_device_policy.py -x -q -m gpu -k "Acrobot or acrobot or digest" 2>&1 | tail -3
timeout 900 python scripts/ab_bench.py --libs shared_divisor=${L}_h.so hot=${L}.so --envs Acrobot-v1:65536:128 Acrobot-v1:262144:128 --rounds 3 --out gpurun_out/${1:-r06_acrobot_hot_constants_ab}.txt
