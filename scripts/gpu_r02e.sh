#!/bin/bash
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
for M in humanoid humanoid-newton; do echo "== $M warm=3"; COOP_WARM=3 timeout 300 scripts/coop_phase_bench.bin $M 32768 2>&1 | tail -15; done > gpurun_out/r02e_phases.txt 2>&1
cat gpurun_out/r02e_phases.txt
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r02e_all.log 2>&1; echo "single-process suite exit $?"; tail -4 gpurun_out/r02e_all.log | cut -c1-300
