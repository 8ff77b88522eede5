#!/bin/bash
# one steady-state bench line per env id (README / DESIGN section 9 table)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$ROOT
: > gpurun_out/r02_allenvs.txt
for spec in "CartPole-v1 65536 128" "Pendulum-v1 65536 128" "Acrobot-v1 65536 128" "MountainCar-v0 65536 128" "MountainCarContinuous-v0 65536 128" \
            "FrozenLake-v1 65536 128" "FrozenLake8x8-v1 65536 128" "CliffWalking-v1 65536 128" "Taxi-v4 65536 128" "Blackjack-v1 65536 128" \
            "HalfCheetah-v5 65536 4" "Hopper-v5 65536 4" "Walker2d-v5 65536 4" "Pusher-v5 65536 4" "Swimmer-v5 65536 4" "Reacher-v5 65536 4" \
            "InvertedPendulum-v5 65536 4" "InvertedDoublePendulum-v5 65536 4" "Ant-v5 65536 4" "Humanoid-v5 32768 4" "HumanoidStandup-v5 32768 4"; do
  set -- $spec
  timeout 200 python bench.py --env $1 --num-envs $2 --inner $3 --no-secondary --pmc off --no-cpu-baseline --no-api 2>/dev/null | python -c "
import json,sys
try:
    r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s N=%-6s value %.4g  K=%d  ms_per_launch %.4g  opt_in %s' % ('$1', '$2', r['value'], r['steps'], r['ms_per_step'], (r.get('opt_in') or {}).get('value')))
except Exception as e: print('$1 FAILED', e)
" | tee -a gpurun_out/r02_allenvs.txt
done
