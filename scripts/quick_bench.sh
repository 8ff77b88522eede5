show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[2])); print(sys.argv[1], "value %.4g" % r["value"], "kernel_ms %.4g" % r["roofline"]["avg_kernel_ms"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
T=${1:-x}
python bench.py --no-api --no-cpu-baseline > gpurun_out/q_${T}_cartpole.json 2>/dev/null; show CartPole gpurun_out/q_${T}_cartpole.json
for E in Pendulum-v1 Acrobot-v1 MountainCar-v0 FrozenLake-v1; do python bench.py --no-api --no-cpu-baseline --env $E > gpurun_out/q_${T}_$E.json 2>/dev/null; show $E gpurun_out/q_${T}_$E.json; done
python bench.py --no-api --no-cpu-baseline --env Ant-v5 --num-envs 32768 --inner 4 --steps 5 --warmup 1 > gpurun_out/q_${T}_ant32.json 2>/dev/null; show Ant32768 gpurun_out/q_${T}_ant32.json
python bench.py --no-api --no-cpu-baseline --env Humanoid-v5 --num-envs 32768 --inner 4 --steps 3 --warmup 1 > gpurun_out/q_${T}_hum.json 2>/dev/null; show Humanoid32768 gpurun_out/q_${T}_hum.json
for E in Ant-v5 HalfCheetah-v5 Hopper-v5 Walker2d-v5 Pusher-v5 Swimmer-v5 InvertedDoublePendulum-v5; do python bench.py --no-api --no-cpu-baseline --env $E --num-envs 65536 --inner 4 --steps 3 --warmup 1 > gpurun_out/q_${T}_$E.json 2>/dev/null; show $E gpurun_out/q_${T}_$E.json; done
