#!/bin/bash
# round 6, call C: the instruction diet of the two-role rollout A/B against the build before it, on one box; the parity tests that
# cover it; the protocol legs again with the short host path of step() / sample().
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_rollout_roles.py tests/test_gpu_parity.py tests/test_gpu_bench_contract.py tests/test_gpu_device_policy.py tests/test_gpu_graph_capture.py tests/test_gpu_cartpole_shared_rng.py -x -q -m gpu > gpurun_out/r06_c_tests.log 2>&1; tail -3 gpurun_out/r06_c_tests.log
timeout 300 python scripts/bench_extras.py --out gpurun_out/r06_c_policy.json --policy-only 2> gpurun_out/r06_c_policy.err | tail -2
timeout 1200 python scripts/ab_bench.py --libs base=gymnasium_amd/csrc/libmi355env_base.so diet=gymnasium_amd/csrc/libmi355env.so \
   --envs CartPole-v1:65536:128 MountainCar-v0:65536:128 MountainCarContinuous-v0:65536:128 CartPole-v1:262144:128 --rounds 3 --out gpurun_out/r06_duo_diet_ab.txt
PROF_STEPS=default timeout 900 scripts/gpu_profile.sh r06_cartpole_rollout > /dev/null 2>&1; grep "rollout_duo" gpurun_out/r06_cartpole_rollout.txt | cut -c1-50,140-250
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary 2>/dev/null | cut -c1-600
