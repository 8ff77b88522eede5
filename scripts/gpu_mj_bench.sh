#!/bin/bash
# steady-state bench lines of the cooperative-kernel robots (A/B of builds: MI355ENV_LIBRARY selects another library)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
export TMPDIR=/tmp PYTHONPATH=$ROOT
for spec in "Ant-v5 65536" "Ant-v5 32768" "HalfCheetah-v5 65536" "Humanoid-v5 32768" "HumanoidStandup-v5 32768"; do set -- $spec
  python bench.py --env $1 --num-envs $2 --inner 4 --no-secondary --pmc off --no-cpu-baseline --no-api 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-22s N=%-6s %.4g env-steps/s' % ('$1', '$2', r['value']))"
done
