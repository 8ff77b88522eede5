#!/bin/bash
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r02m}
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_wrappers.py -m gpu -q -x > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest exit $?"; tail -30 gpurun_out/${TAG}_pytest.log
