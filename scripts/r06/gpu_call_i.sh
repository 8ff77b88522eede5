#!/bin/bash
# round 6, call I: Acrobot's rollout as two wavefronts split by function (rollout_acrobot_pair_kernel): parity, then A/B against the one-role kernel; the full-batch cooperative vs one-lane test
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_rollout_roles.py -x -q -m gpu -k "Acrobot" > gpurun_out/r06_i_tests.log 2>&1; tail -3 gpurun_out/r06_i_tests.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r06_i_tests2.log 2>&1; tail -3 gpurun_out/r06_i_tests2.log
timeout 900 python scripts/ab_bench.py --libs onerole=gymnasium_amd/csrc/libmi355env_h.so pair=gymnasium_amd/csrc/libmi355env.so --envs Acrobot-v1:65536:128 Acrobot-v1:262144:128 --rounds 3 --out gpurun_out/r06_acrobot_pair_ab.txt
timeout 900 python scripts/ab_bench.py --libs onerole=gymnasium_amd/csrc/libmi355env_h.so pair=gymnasium_amd/csrc/libmi355env.so --envs Acrobot-v1:65536:128 --rounds 2 --env-kwargs '{"fast_math": true}' --out gpurun_out/r06_acrobot_pair_ab.txt
timeout 1200 python -m pytest tests/test_gpu_mujoco.py -x -q -m gpu -k "full_batch" -s 2>&1 | grep -E "passed|failed|max diff|Error" | tail -6
