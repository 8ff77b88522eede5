#!/bin/bash
# cooperative-kernel phase timing: scripts/gpu_phases.sh <tag> <warm> <model> [<model> ...]
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
TAG=$1; W=$2; shift 2
for M in "$@"; do echo "== $M warm=$W"; COOP_WARM=$W timeout 300 scripts/coop_phase_bench.bin $M 32768 2>&1 | tail -15; done > gpurun_out/${TAG}_phases.txt 2>&1
cat gpurun_out/${TAG}_phases.txt
