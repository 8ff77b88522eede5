#!/bin/bash
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$ROOT
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r02u_pytest.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/r02u_pytest.log
MI355ENV_ZEROCOPY=0 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wrappers.py -m gpu -q > gpurun_out/r02u_pytest_staged.log 2>&1; echo "pytest (staged copies) exit $?"; tail -3 gpurun_out/r02u_pytest_staged.log
