#!/bin/bash
# round 6, call X: the tabular rollout draws step t + 1's action before step t's table lookups: parity, A/B
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
L=gymnasium_amd/csrc/libmi355env
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_device_policy.py -x -q -m gpu 2>&1 | tail -2
timeout 900 python scripts/ab_bench.py --libs h=${L}_h.so ahead=${L}.so --envs FrozenLake-v1:65536:128 Taxi-v4:65536:128 Blackjack-v1:65536:128 CliffWalking-v1:65536:128 --rounds 3 --out gpurun_out/r06_tab_draw_ahead_ab.txt
