#!/bin/bash
# round 6, call B: the on-device policy of the per-step path (new GPU tests + the protocol legs), the ToyText rollouts with the table as a real
# LDS array (was: generic pointers -> flat loads), and the regression run of the GPU suite.
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_device_policy.py -x -q -m gpu > gpurun_out/r06_b_policy_tests.log 2>&1; tail -5 gpurun_out/r06_b_policy_tests.log
timeout 300 python scripts/bench_extras.py --out gpurun_out/r06_b_policy.json --policy-only 2> gpurun_out/r06_b_policy.err | tail -2
for e in FrozenLake-v1 Taxi-v4 Blackjack-v1 FrozenLake8x8-v1; do
  echo "== $e"; timeout 300 python bench.py --env $e --steps 20 --warmup 3 --no-secondary --pmc off --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('verified'), d.get('output_sha256','')[:16])"
done
for e in FrozenLake-v1 Taxi-v4; do
  t=$(echo $e | tr 'A-Z' 'a-z' | sed 's/-v.//')
  PROF_STEPS=20 PROF_WARMUP=3 timeout 600 scripts/gpu_profile.sh r06_${t}_rollout_lds --env $e > /dev/null 2>&1
  grep "tab_rollout" gpurun_out/r06_${t}_rollout_lds.txt | grep -E "\| [0-9]+ \| [0-9.]+ \| [0-9.]+ \| [0-9.]+$|VMEM_RD|INSTS_LDS|WAIT_ANY|WAVE_CYCLES" | cut -c1-60,150-260
done
timeout 2400 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_device_policy.py > gpurun_out/r06_b_pytest_gpu.log 2>&1; tail -5 gpurun_out/r06_b_pytest_gpu.log
