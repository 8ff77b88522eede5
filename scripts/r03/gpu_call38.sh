#!/bin/bash
# round-3 GPU call 38 (run twice: union layout, then the zero-length array): the final library -- MuJoCo tests + guard, Ant and Humanoid against the call-35 objects
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_scheduler_guard.py -m gpu -q > gpurun_out/r03af_pytest.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/r03af_pytest.log
tail -3 gpurun_out/r03af_pytest.log
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[2])); print(sys.argv[1], "value %.4g" % r["value"], "kernel_ms %.4g" % r["roofline"]["avg_kernel_ms"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for V in old16 product; do
  LL=gymnasium_amd/csrc/libmi355env_$V.so; [ $V = product ] && LL=gymnasium_amd/csrc/libmi355env.so
  MI355ENV_LIBRARY=$PWD/$LL timeout 300 python bench.py --no-api --no-cpu-baseline --no-secondary --pmc off --spinup 0.2 --env Ant-v5 --num-envs 65536 --inner 4 > gpurun_out/r03af_tmp.json 2>/dev/null
  show "Ant-v5 $V" gpurun_out/r03af_tmp.json
done
for V in old16 product; do   # (old16 carries the call-35 physics32.o: Humanoid with the 40 128-byte Board)
  LL=gymnasium_amd/csrc/libmi355env_$V.so; [ $V = product ] && LL=gymnasium_amd/csrc/libmi355env.so
  MI355ENV_LIBRARY=$PWD/$LL timeout 300 python bench.py --no-api --no-cpu-baseline --no-secondary --pmc off --spinup 0.2 --env Humanoid-v5 --num-envs 32768 --inner 4 > gpurun_out/r03af_tmp.json 2>/dev/null
  show "Humanoid-v5 $V" gpurun_out/r03af_tmp.json
done
