#!/bin/bash
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r02j}
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/${TAG}_pytest.log
tail -8 gpurun_out/${TAG}_pytest.log
for env in CartPole-v1 Pendulum-v1 Acrobot-v1 MountainCar-v0 MountainCarContinuous-v0; do
  for fast in false true; do
    timeout 300 python bench.py --env $env --num-envs 65536 --no-secondary --pmc off --no-cpu-baseline --no-api --sustained 1.0 --env-kwargs "{\"fast_math\": $fast}" > gpurun_out/${TAG}_${env}_${fast}.json 2> gpurun_out/${TAG}_${env}_${fast}.err
    python - <<PY
import json
try:
    r = json.load(open("gpurun_out/${TAG}_${env}_${fast}.json"))
    print("$env fast=$fast value %.4g sustained %.4g frac %.3g kernel_ms %.4g" % (r["value"], r.get("sustained_value", 0), r["roofline"]["frac"], r["roofline"]["avg_kernel_ms"]))
except Exception as e:
    print("$env fast=$fast FAILED", e); print(open("gpurun_out/${TAG}_${env}_${fast}.err").read()[-800:])
PY
  done
done
