#!/bin/bash
# GPU tests file by file (an abort in one file does not hide the others); optional first argument: a -k expression run FIRST, alone.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-split}; ALONE=${2:-}
export TMPDIR=/tmp
if [ -n "$ALONE" ]; then
  AMD_LOG_LEVEL=1 timeout 600 python -m pytest tests/test_gpu_mujoco.py -m gpu -q -x -k "$ALONE" > gpurun_out/${TAG}_alone.log 2>&1; echo "alone exit $?"; tail -25 gpurun_out/${TAG}_alone.log
fi
for f in tests/test_gpu_distributed.py tests/test_gpu_mujoco.py tests/test_gpu_parity.py tests/test_gpu_scheduler_guard.py tests/test_gpu_wrappers.py tests/test_mujoco_reference_pins.py; do
  b=$(basename $f .py)
  timeout 1200 python -m pytest $f -m gpu -q > gpurun_out/${TAG}_$b.log 2>&1; echo "$b exit $?"; grep -E "passed|failed|Fatal|error" gpurun_out/${TAG}_$b.log | tail -3
done
