#!/bin/bash
# cooperative-kernel change check: phase timing (ant, humanoid PGS / Newton; standing and fallen robots) + the MuJoCo GPU tests + scheduler guard
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-coop}
export TMPDIR=/tmp PYTHONPATH=$ROOT
for W in 3 40; do for M in humanoid humanoid-newton ant cheetah; do echo "== $M warm=$W"; COOP_WARM=$W timeout 300 scripts/coop_phase_bench.bin $M $([ $M = ant -o $M = cheetah ] && echo 65536 || echo 32768) 2>&1 | tail -16; done; done > gpurun_out/${TAG}_phases.txt 2>&1
grep -E "^==|env-steps/s|kinematics|RNE|total" gpurun_out/${TAG}_phases.txt
timeout 900 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_scheduler_guard.py tests/test_mujoco_reference_pins.py -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/${TAG}_pytest.log
