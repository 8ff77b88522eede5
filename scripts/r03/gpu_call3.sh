#!/bin/bash
# round-3 GPU call 3: classic-control parity after the trig / pow trims + their bench lines and SQ instruction counters; MFMA microbench (tree fixed, fma-chain check)
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wrappers.py tests/test_gpu_bench_contract.py -m gpu -q > gpurun_out/r03c_pytest.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/r03c_pytest.log
tail -12 gpurun_out/r03c_pytest.log
echo "=== MFMA microbenchmark"; timeout 120 scripts/humanoid_mfma.bin | tee gpurun_out/r03c_mfma_humanoid.txt
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[2])); print(sys.argv[1], "value %.4g" % r["value"], "frac %.3g" % r["roofline"]["frac"], "kernel_ms %.4g" % r["roofline"]["avg_kernel_ms"], "opt_in %.4g" % r.get("opt_in", {}).get("value", 0))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for E in CartPole-v1 Pendulum-v1 Acrobot-v1 MountainCar-v0 MountainCarContinuous-v0; do
  timeout 300 python bench.py --no-api --no-cpu-baseline --no-secondary --pmc off --env $E > gpurun_out/r03c_bench_$E.json 2>> gpurun_out/r03c_bench.err; show $E gpurun_out/r03c_bench_$E.json
done
for E in Pendulum-v1 Acrobot-v1; do
  PROF_STEPS=30 PROF_WARMUP=3 timeout 600 scripts/gpu_profile.sh r03c_${E}_rollout --env $E --no-secondary --pmc off
done
tail -3 gpurun_out/r03c_bench.err
