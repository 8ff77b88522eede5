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
try:
    r = json.load(open(sys.argv[2])); print(sys.argv[1], "value %.4g" % r["value"], "frac %.3g" % r["roofline"]["frac"], "kernel_ms %.4g" % r["roofline"]["avg_kernel_ms"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for N in 262144 1048576; do
  timeout 300 python bench.py --no-api --no-cpu-baseline --num-envs $N > gpurun_out/${TAG}_bench_N$N.json 2>> gpurun_out/${TAG}_bench.err; show "CartPole N=$N" gpurun_out/${TAG}_bench_N$N.json
done
for E in Pendulum-v1 Acrobot-v1 MountainCar-v0 MountainCarContinuous-v0 FrozenLake-v1 FrozenLake8x8-v1 CliffWalking-v1 Taxi-v4 Blackjack-v1; do
  timeout 300 python bench.py --no-api --no-cpu-baseline --env $E > gpurun_out/${TAG}_bench_$E.json 2>> gpurun_out/${TAG}_bench.err; show $E gpurun_out/${TAG}_bench_$E.json
done
# MuJoCo family: Ant / Humanoid at the BASELINE.json sizes with the CPU oracle beside them, the others at 65536
timeout 600 python bench.py --no-api --env Ant-v5 --num-envs 32768 --inner 4 --steps 5 --warmup 1 > gpurun_out/${TAG}_bench_Ant-v5_N32768.json 2>> gpurun_out/${TAG}_bench.err; show "Ant-v5 N=32768" gpurun_out/${TAG}_bench_Ant-v5_N32768.json
timeout 600 python bench.py --no-api --env Humanoid-v5 --num-envs 32768 --inner 4 --steps 3 --warmup 1 > gpurun_out/${TAG}_bench_Humanoid-v5_N32768.json 2>> gpurun_out/${TAG}_bench.err; show "Humanoid-v5 N=32768" gpurun_out/${TAG}_bench_Humanoid-v5_N32768.json
for E in Ant-v5 HalfCheetah-v5 Hopper-v5 Walker2d-v5 InvertedPendulum-v5 InvertedDoublePendulum-v5 Reacher-v5 Swimmer-v5 Pusher-v5; do
  timeout 600 python bench.py --no-api --no-cpu-baseline --env $E --num-envs 65536 --inner 4 --steps 3 --warmup 1 > gpurun_out/${TAG}_bench_${E}_N65536.json 2>> gpurun_out/${TAG}_bench.err; show "$E N=65536" gpurun_out/${TAG}_bench_${E}_N65536.json
done
timeout 600 python bench.py --no-api --no-cpu-baseline --env HumanoidStandup-v5 --num-envs 32768 --inner 4 --steps 3 --warmup 1 > gpurun_out/${TAG}_bench_HumanoidStandup-v5_N32768.json 2>> gpurun_out/${TAG}_bench.err; show "HumanoidStandup-v5 N=32768" gpurun_out/${TAG}_bench_HumanoidStandup-v5_N32768.json
PROF_STEPS=3 PROF_WARMUP=1 timeout 900 scripts/gpu_profile.sh ${TAG}_ant --env Ant-v5 --num-envs 32768 --inner 4
PROF_STEPS=2 PROF_WARMUP=1 timeout 900 scripts/gpu_profile.sh ${TAG}_humanoid --env Humanoid-v5 --num-envs 32768 --inner 2
tail -3 gpurun_out/${TAG}_bench.err
