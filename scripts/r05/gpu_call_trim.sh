#!/bin/bash
# round 5: MI_DUO_TRIM (two instruction-count trims of the two-role rollout's aux role) -- parity, then A/B against the untrimmed build
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rollout_roles.py -q -m gpu -k "digest or roles or rollout or full_size" 2>&1 | tail -2
python scripts/ab_bench.py --libs trim0=gymnasium_amd/csrc/libmi355env_trim0.so trim1=gymnasium_amd/csrc/libmi355env.so --envs CartPole-v1:65536:128 MountainCar-v0:65536:128 MountainCarContinuous-v0:65536:128 --rounds 3 --out gpurun_out/r05_duo_trim_ab.txt
