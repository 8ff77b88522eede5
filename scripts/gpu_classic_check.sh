#!/bin/bash
# classic-control change check: the parity suite + one steady-state bench line per classic env (exact and fast_math)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wrappers.py -m gpu -q 2>&1 | tail -3
for env in CartPole-v1 Pendulum-v1 Acrobot-v1 MountainCar-v0 MountainCarContinuous-v0; do
  python bench.py --env $env --num-envs 65536 --no-secondary --pmc off --no-cpu-baseline --no-api 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-26s exact %.4g  fast_math %.4g' % ('$env', r['value'], (r.get('opt_in') or {}).get('value', 0)))"
done
