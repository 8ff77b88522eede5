#!/bin/bash
# round 5: bench.py's in-run verification (first timed launch vs the oracle) for every family -- classic, ToyText, one-lane and cooperative MuJoCo kinds
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
{
for spec in "Pendulum-v1 65536 128" "Acrobot-v1 65536 128" "MountainCar-v0 65536 128" "MountainCarContinuous-v0 65536 128" "FrozenLake-v1 65536 128" "Taxi-v4 65536 128" "Blackjack-v1 65536 128" \
            "CliffWalking-v1 65536 128" "HalfCheetah-v5 32768 4" "Hopper-v5 32768 4" "Walker2d-v5 32768 4" "Ant-v5 32768 4" "Humanoid-v5 32768 4" "HumanoidStandup-v5 32768 4" "Reacher-v5 32768 4" \
            "Swimmer-v5 32768 4" "Pusher-v5 32768 4" "InvertedPendulum-v5 32768 4" "InvertedDoublePendulum-v5 32768 4"; do
  set -- $spec
  timeout 300 python bench.py --env $1 --num-envs $2 --inner $3 --steps 5 --warmup 2 --no-extras --no-cpu-baseline --pmc off --sustained 0 2>/dev/null | python -c "
import json,sys
try:
    r=json.loads(sys.stdin.readline()); v=r['verified']
    print('%-28s value %.4g  verified ok=%s envs=%s compare=%s max_abs_diff=%.3g policy=%s  sha %s' % ('$1', r['value'], v.get('ok'), v.get('envs'), v.get('compare'), v.get('max_abs_diff', float('nan')), v.get('policy_equals_host_sample'), r['output_sha256'][:12]))
except Exception as e:
    print('$1 FAILED', e)"
done
} > gpurun_out/r05_bench_verification_all_envs.txt 2>&1
cat gpurun_out/r05_bench_verification_all_envs.txt
