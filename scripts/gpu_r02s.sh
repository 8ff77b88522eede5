#!/bin/bash
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$ROOT
for i in 1 2; do
echo "copy engine:"; python scripts/numpy_step_bench.py 2>/dev/null | tail -1
echo "zero copy:"; MI355ENV_ZEROCOPY=1 python scripts/numpy_step_bench.py 2>/dev/null | tail -1
done
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "torch_mode_errors or step_async" 2>&1 | tail -3
