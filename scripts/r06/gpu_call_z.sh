#!/bin/bash
# round 6, call Z: Blackjack's branch-free rollout (bj_rollout_lean_kernel): parity against the oracle incl. the forced general routines, A/B against tab_rollout_kernel
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
L=gymnasium_amd/csrc/libmi355env.so
timeout 900 python -m pytest tests/test_gpu_tabular_lean.py -x -q -m gpu 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_device_policy.py tests/test_gpu_bench_contract.py -x -q -m gpu -k "toytext or Taxi or FrozenLake or tab or digest or lackjack" 2>&1 | tail -3
timeout 900 python scripts/ab_bench.py --libs branchy=${L}@MI355ENV_TAB_LEAN=0 lean=${L} --envs Blackjack-v1:65536:128 --rounds 3 --out gpurun_out/r06_blackjack_lean_ab.txt
