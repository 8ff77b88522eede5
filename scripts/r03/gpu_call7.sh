#!/bin/bash
# round-3 GPU call 7: MFMA block Cholesky in the PGS kernels -- MuJoCo GPU tests, then A/B (phase harness + bench.py on two libraries)
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_scheduler_guard.py tests/test_gpu_parity.py -m gpu -q > gpurun_out/r03e_pytest.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/r03e_pytest.log
tail -6 gpurun_out/r03e_pytest.log
for W in 3 12; do
  for V in h0 h1; do
    echo "=== phase $V (h1 = MFMA Schur update) warm=$W"; COOP_WARM=$W timeout 120 scripts/phase_$V.bin humanoid 32768 | tee -a gpurun_out/r03e_phase_$V.txt | grep -v "^  \(integrator\|kinematics\|com_\|collision\|crb\|other\)"
  done
done
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[2])); print(sys.argv[1], "value %.4g" % r["value"], "kernel_ms %.4g" % r["roofline"]["avg_kernel_ms"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for rep in 1 2 3; do
  for V in nomfma product; do
    L=gymnasium_amd/csrc/libmi355env_$V.so; [ $V = product ] && L=gymnasium_amd/csrc/libmi355env.so
    MI355ENV_LIBRARY=$PWD/$L timeout 300 python bench.py --no-api --no-cpu-baseline --no-secondary --pmc off --env Humanoid-v5 --num-envs 32768 --inner 4 > gpurun_out/r03e_hum_${V}_$rep.json 2>/dev/null
    show "Humanoid $V rep$rep" gpurun_out/r03e_hum_${V}_$rep.json
  done
done
