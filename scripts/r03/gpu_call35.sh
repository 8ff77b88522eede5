#!/bin/bash
# round-3 GPU call 35: PGS keeps M^-1 J_c^T of 15 contacts (not 7) on the blackboard (MJX_PGS_MORE_BLOCKS)
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_scheduler_guard.py -m gpu -q -k "umanoid" > gpurun_out/r03ac_pytest.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/r03ac_pytest.log
tail -3 gpurun_out/r03ac_pytest.log
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[2])); print(sys.argv[1], "value %.4g" % r["value"], "kernel_ms %.4g" % r["roofline"]["avg_kernel_ms"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for rep in 1 2; do
  for E in Humanoid-v5 HumanoidStandup-v5; do
    for V in prev32 product; do
      LL=gymnasium_amd/csrc/libmi355env_$V.so; [ $V = product ] && LL=gymnasium_amd/csrc/libmi355env.so
      MI355ENV_LIBRARY=$PWD/$LL timeout 300 python bench.py --no-api --no-cpu-baseline --no-secondary --pmc off --spinup 0.2 --env $E --num-envs 32768 --inner 4 > gpurun_out/r03ac_tmp.json 2>/dev/null
      show "$E $V rep$rep" gpurun_out/r03ac_tmp.json
    done
  done
done
