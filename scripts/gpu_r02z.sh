#!/bin/bash
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "ragged or torch_mode" > gpurun_out/r02z_pytest.log 2>&1; echo "pytest exit $?"; tail -15 gpurun_out/r02z_pytest.log
