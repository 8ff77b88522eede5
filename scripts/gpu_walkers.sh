#!/bin/bash
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$ROOT
timeout 900 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_scheduler_guard.py tests/test_mujoco_reference_pins.py tests/test_gpu_parity.py -m gpu -q -k "opper or alker or guard or ragged" 2>&1 | tail -4
for spec in "Hopper-v5 65536" "Walker2d-v5 65536"; do set -- $spec
  for mode in coop serial; do
    if [ $mode = serial ]; then export MI355ENV_MJ_SERIAL=1; else unset MI355ENV_MJ_SERIAL; fi
    python bench.py --env $1 --num-envs $2 --inner 4 --no-secondary --pmc off --no-cpu-baseline --no-api 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-14s %-7s %.4g env-steps/s' % ('$1', '$mode', r['value']))"
  done
done
