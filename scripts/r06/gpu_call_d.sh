#!/bin/bash
# round 6, call D: the two-role rollout without loop-carried lane masks (phase overhead), chunk 8 against 4; the MuJoCo window verification; the refused batch of the shared mode
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_rollout_roles.py tests/test_gpu_parity.py tests/test_gpu_bench_contract.py tests/test_gpu_cartpole_shared_rng.py -x -q -m gpu > gpurun_out/r06_d_tests.log 2>&1; tail -3 gpurun_out/r06_d_tests.log
timeout 1500 python scripts/ab_bench.py --libs base=gymnasium_amd/csrc/libmi355env_base.so diet=gymnasium_amd/csrc/libmi355env_diet.so nomask=gymnasium_amd/csrc/libmi355env.so c8=gymnasium_amd/csrc/libmi355env_c8.so \
   --envs CartPole-v1:65536:128 MountainCar-v0:65536:128 MountainCarContinuous-v0:65536:128 --rounds 3 --out gpurun_out/r06_duo_diet_ab.txt
for e in Ant-v5 Humanoid-v5; do
  timeout 600 python bench.py --env $e --num-envs 32768 --inner 4 --steps 10 --warmup 3 --no-secondary --pmc off --no-cpu-baseline 2>gpurun_out/r06_d_$e.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], json.dumps(d['verified']))"
done
