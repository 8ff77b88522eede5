#!/bin/bash
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$ROOT
timeout 1200 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_scheduler_guard.py tests/test_mujoco_reference_pins.py tests/test_gpu_parity.py -m gpu -q > gpurun_out/r02x_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r02x_pytest.log
for e in "Ant-v5 65536" "HalfCheetah-v5 65536" "Humanoid-v5 32768"; do set -- $e
  python bench.py --env $1 --num-envs $2 --inner 4 --steps 6 --no-secondary --pmc off --no-cpu-baseline --no-api --sustained 1.0 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '%.4g burst %.4g sustained' % (r['value'], r.get('sustained_value',0)))"
done
