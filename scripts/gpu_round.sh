#!/bin/bash
# One gpurun call: parity tests, the bench line, profiles, and the N / env sweeps.  Everything lands in gpurun_out/.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r01}
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/${TAG}_pytest.log
tail -5 gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench exit $?"; cat gpurun_out/${TAG}_bench.json
timeout 900 scripts/gpu_profile.sh ${TAG}_cartpole
show() { python - "$1" "$2" <<'PY'
import json, sys
r = json.load(open(sys.argv[2])); print(sys.argv[1], "value", r["value"], "frac", r["roofline"]["frac"], "kernel_ms", r["roofline"]["avg_kernel_ms"])
PY
}
for N in 131072 262144 1048576; do
  timeout 300 python bench.py --no-api --no-cpu-baseline --num-envs $N > gpurun_out/${TAG}_bench_N$N.json 2>> gpurun_out/${TAG}_bench.err; show "CartPole N=$N" gpurun_out/${TAG}_bench_N$N.json
done
for E in Pendulum-v1 Acrobot-v1 MountainCar-v0 MountainCarContinuous-v0; do
  timeout 300 python bench.py --no-api --no-cpu-baseline --env $E > gpurun_out/${TAG}_bench_$E.json 2>> gpurun_out/${TAG}_bench.err; show $E gpurun_out/${TAG}_bench_$E.json
done
for N in 32768 65536 131072; do
  for E in HalfCheetah-v5 Ant-v5; do
    timeout 600 python bench.py --no-api --env $E --num-envs $N --inner 8 --steps 5 --warmup 1 > gpurun_out/${TAG}_bench_${E}_N$N.json 2>> gpurun_out/${TAG}_bench.err; show "$E N=$N" gpurun_out/${TAG}_bench_${E}_N$N.json
  done
done
PROF_STEPS=3 PROF_WARMUP=1 timeout 900 scripts/gpu_profile.sh ${TAG}_ant --env Ant-v5 --num-envs 32768 --inner 8
