#!/bin/bash
# round 5, GPU call C: new tests, the driver's bench command, the r05 profiles and the occupancy sweep that prices the lane-split idea
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
bash scripts/gpu_call.sh r05c tests -k "per_sub_environment or correctly_rounded or drivers_own or shared_rng or short_explicit"
cp gpurun_out/bench_full.json gpurun_out/r05c_bench_full.json 2>/dev/null
PROF_STEPS=default timeout 600 scripts/gpu_profile.sh r05_cartpole_rollout > /dev/null 2>&1; grep -i "rollout_duo" gpurun_out/r05_cartpole_rollout.txt | head -8
PROF_STEPS=10 PROF_WARMUP=2 timeout 600 scripts/gpu_profile.sh r05_Acrobot-v1_rollout --env Acrobot-v1 > /dev/null 2>&1; grep -i "rollout_kernel" gpurun_out/r05_Acrobot-v1_rollout.txt | head -4
PROF_STEPS=20 PROF_WARMUP=2 timeout 600 scripts/gpu_profile.sh r05_Pendulum-v1_rollout --env Pendulum-v1 > /dev/null 2>&1; grep -i "rollout_kernel" gpurun_out/r05_Pendulum-v1_rollout.txt | head -4
{
echo "occupancy sweep (round 5): the SAME rollout kernel at 1 / 2 / 4 wavefronts per SIMD = num_envs 65536 / 131072 / 262144; env-steps/s and the events' time per launch"
for E in Acrobot-v1 Pendulum-v1 CartPole-v1; do
  for N in 65536 131072 262144; do
    timeout 300 python bench.py --env $E --num-envs $N --no-extras --no-verify --no-cpu-baseline --pmc off --steps 30 --warmup 3 --sustained 0 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('$E', $N, '%.4g env-steps/s' % r['value'], 'kernel_ms %.4f' % r['roofline']['avg_kernel_ms'], 'hbm frac %.3f' % r['roofline']['frac'])"
  done
done
for E in Acrobot-v1 Pendulum-v1; do
  for N in 65536 262144; do
    timeout 300 python bench.py --env $E --num-envs $N --env-kwargs '{"fast_math": true}' --no-extras --no-verify --no-cpu-baseline --pmc off --steps 30 --warmup 3 --sustained 0 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('$E fast_math', $N, '%.4g env-steps/s' % r['value'], 'kernel_ms %.4f' % r['roofline']['avg_kernel_ms'])"
  done
done
} > gpurun_out/r05_occupancy_sweep.txt 2>&1
cat gpurun_out/r05_occupancy_sweep.txt
