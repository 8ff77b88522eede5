#!/bin/bash
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$ROOT
for i in 1 2; do
echo "zero copy (default):"; python scripts/numpy_step_bench.py 2>/dev/null | tail -1
echo "staged copies:"; MI355ENV_ZEROCOPY=0 python scripts/numpy_step_bench.py 2>/dev/null | tail -1
done
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r02t_pytest.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/r02t_pytest.log
