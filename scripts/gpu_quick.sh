#!/bin/bash
# Quick GPU check: fused-rollout parity tests + one bench line (no secondary measurements).
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-quick}; shift || true
timeout 900 python -m pytest tests -m gpu -x -q  > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/${TAG}_pytest.log
timeout 300 python bench.py --no-api --no-cpu-baseline "$@" > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench exit $?"
python - <<PY
import json; r=json.load(open("gpurun_out/${TAG}_bench.json")); print("value", r["value"], "frac", r["roofline"]["frac"], "kernel_ms", r["roofline"]["avg_kernel_ms"])
PY
