#!/bin/bash
# round 6, call AF: Acrobot's three quotients by d1 through one refined reciprocal (SharedDivisor): parity, A/B against the build with the grouped squares only;
# role timers of the Pendulum rollout after the third cut
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
L=gymnasium_amd/csrc/libmi355env
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_contract.py tests/test_gpu_device_policy.py -x -q -m gpu -k "Acrobot or acrobot or digest" 2>&1 | tail -3
timeout 900 python scripts/ab_bench.py --libs grouped=${L}_h.so shared_divisor=${L}.so --envs Acrobot-v1:65536:128 Acrobot-v1:262144:128 --rounds 3 --out gpurun_out/r06_acrobot_shared_divisor_ab.txt
for env in Pendulum-v1; do
  echo "## timing $env"; MI355ENV_LIBRARY=${L}_timing.so timeout 300 python scripts/r04/duo_timing.py $env 2>&1 | grep "duo timing" | tail -8 | sort
done > gpurun_out/r06_duo_timing_third_cut.txt 2>&1
cat gpurun_out/r06_duo_timing_third_cut.txt
