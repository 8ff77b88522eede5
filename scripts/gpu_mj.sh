#!/bin/bash
# MuJoCo-family GPU check: parity tests + bench lines for the cooperative kernel (and the one-lane kernel for comparison).
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-mj}
timeout 900 python -m pytest tests/test_gpu_mujoco.py -x -q -s > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest exit $?"; grep -v "^$" gpurun_out/${TAG}_pytest.log | tail -25
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[2])); print(sys.argv[1], "value", r["value"], "ms_per_step", r["ms_per_step"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for spec in "Ant-v5 65536" "Ant-v5 32768" "HalfCheetah-v5 65536" "Humanoid-v5 32768"; do
  set -- $spec
  timeout 300 python bench.py --no-api --no-cpu-baseline --env $1 --num-envs $2 --inner 4 --steps 3 --warmup 1 > gpurun_out/${TAG}_bench_$1_N$2.json 2>> gpurun_out/${TAG}_bench.err; show "coop $1 N=$2" gpurun_out/${TAG}_bench_$1_N$2.json
done
tail -5 gpurun_out/${TAG}_bench.err
