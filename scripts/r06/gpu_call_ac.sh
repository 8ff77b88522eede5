#!/bin/bash
# round 6, call AC: lean tabular rollout with the record's reads issued before the two generator steps (scheduling barriers) and the initial-state reads unconditional: parity, A/B
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
L=gymnasium_amd/csrc/libmi355env
timeout 900 python -m pytest tests/test_gpu_tabular_lean.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_contract.py -x -q -m gpu -k "toytext or Taxi or FrozenLake or tab or digest" 2>&1 | tail -2
timeout 900 python scripts/ab_bench.py --libs before=${L}_h.so reads_first=${L}.so --envs FrozenLake-v1:65536:128 FrozenLake8x8-v1:65536:128 Taxi-v4:65536:128 CliffWalking-v1:65536:128 --rounds 3 --out gpurun_out/r06_tab_lean_reads_first_ab.txt
