#!/bin/bash
# round-3 GPU call 25: does physics16.hip still miscompile with -sink-insts-to-avoid-spills (the flag set of physics32.hip) after the round-3 rewrites?
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
L=$PWD/gymnasium_amd/csrc/libmi355env_sink16.so
timeout 600 python scripts/r03/guard_variant.py $L 2>&1 | tail -6 | tee gpurun_out/r03v_guard.txt
MI355ENV_LIBRARY=$L timeout 900 python -m pytest tests/test_gpu_mujoco.py -m gpu -q -k "not humanoid" > gpurun_out/r03v_pytest.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/r03v_pytest.log
tail -5 gpurun_out/r03v_pytest.log
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[2])); print(sys.argv[1], "value %.4g" % r["value"], "kernel_ms %.4g" % r["roofline"]["avg_kernel_ms"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for E in Ant-v5 HalfCheetah-v5 Walker2d-v5 Hopper-v5; do
  for V in product sink16; do
    LL=gymnasium_amd/csrc/libmi355env_$V.so; [ $V = product ] && LL=gymnasium_amd/csrc/libmi355env.so
    MI355ENV_LIBRARY=$PWD/$LL timeout 300 python bench.py --no-api --no-cpu-baseline --no-secondary --pmc off --spinup 0.2 --env $E --num-envs 65536 --inner 4 > gpurun_out/r03v_tmp.json 2>/dev/null
    show "$E $V" gpurun_out/r03v_tmp.json
  done
done
