#!/bin/bash
# round-3 GPU call 21: flat per-body / per-dof static records (MJX_FLAT_JOINTS) against the chained model tables
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_scheduler_guard.py -m gpu -q > gpurun_out/r03q_pytest.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/r03q_pytest.log
tail -3 gpurun_out/r03q_pytest.log
for R in humanoid ant; do
  N=32768; [ $R = ant ] && N=65536
  for V in f0 f1; do
    echo "=== $V (f0 = model tables, f1 = flat records) $R warm=3"; COOP_WARM=3 timeout 120 scripts/phase_$V.bin $R $N | tee -a gpurun_out/r03q_phase_$V.txt | grep -v "pgs:\|solver:"
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
for rep in 1 2; do
  for V in prev32 product; do
    L=gymnasium_amd/csrc/libmi355env_$V.so; [ $V = product ] && L=gymnasium_amd/csrc/libmi355env.so
    MI355ENV_LIBRARY=$PWD/$L timeout 300 python bench.py --no-api --no-cpu-baseline --no-secondary --pmc off --spinup 0.2 --env Humanoid-v5 --num-envs 32768 --inner 4 > gpurun_out/r03q_hum_${V}_$rep.json 2>/dev/null
    show "Humanoid $V rep$rep" gpurun_out/r03q_hum_${V}_$rep.json
  done
  for V in prev16 product; do
    L=gymnasium_amd/csrc/libmi355env_$V.so; [ $V = product ] && L=gymnasium_amd/csrc/libmi355env.so
    MI355ENV_LIBRARY=$PWD/$L timeout 300 python bench.py --no-api --no-cpu-baseline --no-secondary --pmc off --spinup 0.2 --env Ant-v5 --num-envs 65536 --inner 4 > gpurun_out/r03q_ant_${V}_$rep.json 2>/dev/null
    show "Ant $V rep$rep" gpurun_out/r03q_ant_${V}_$rep.json
  done
done
for E in HalfCheetah-v5 Walker2d-v5 Hopper-v5; do
  for V in prev16 product; do
    L=gymnasium_amd/csrc/libmi355env_$V.so; [ $V = product ] && L=gymnasium_amd/csrc/libmi355env.so
    MI355ENV_LIBRARY=$PWD/$L timeout 300 python bench.py --no-api --no-cpu-baseline --no-secondary --pmc off --spinup 0.2 --env $E --num-envs 65536 --inner 4 > gpurun_out/r03q_tmp.json 2>/dev/null
    show "$E $V" gpurun_out/r03q_tmp.json
  done
done
