#!/bin/bash
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r02l}
export TMPDIR=/tmp
for W in 3 40; do for M in humanoid; do echo "== $M warm=$W"; COOP_WARM=$W timeout 300 scripts/coop_phase_bench.bin $M 32768 2>&1 | tail -16; done; done > gpurun_out/${TAG}_phases.txt 2>&1
cat gpurun_out/${TAG}_phases.txt
timeout 900 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_scheduler_guard.py -m gpu -q -k "umanoid or guard" > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/${TAG}_pytest.log
