#!/bin/bash
# One gpurun call: parity tests, the bench line, profiles, and the N / env sweeps.  Everything lands in gpurun_out/.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r01_b}
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/${TAG}_pytest.log
tail -5 gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench exit $?"; cat gpurun_out/${TAG}_bench.json
timeout 900 scripts/gpu_profile.sh ${TAG}_cartpole
for N in 131072 262144 1048576; do
  timeout 300 python bench.py --no-api --no-cpu-baseline --num-envs $N > gpurun_out/${TAG}_bench_N$N.json 2>> gpurun_out/${TAG}_bench.err
  python - <<PY
import json; r=json.load(open("gpurun_out/${TAG}_bench_N$N.json")); print("N=$N", r["value"], r["roofline"]["frac"], r["roofline"]["avg_kernel_ms"])
PY
done
for E in Pendulum-v1 Acrobot-v1 MountainCar-v0 MountainCarContinuous-v0; do
  timeout 300 python bench.py --no-api --no-cpu-baseline --env $E > gpurun_out/${TAG}_bench_$E.json 2>> gpurun_out/${TAG}_bench.err
  python - <<PY
import json; r=json.load(open("gpurun_out/${TAG}_bench_$E.json")); print("$E", r["value"], r["roofline"]["frac"], r["roofline"]["avg_kernel_ms"])
PY
done
