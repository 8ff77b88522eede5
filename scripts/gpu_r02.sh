#!/bin/bash
# Round-2 GPU call: parity tests, then the default bench line (timed), everything under gpurun_out/<tag>_*.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r02}; shift || true
export TMPDIR=/tmp
timeout ${PYTEST_TIMEOUT:-1500} python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/${TAG}_pytest.log
tail -15 gpurun_out/${TAG}_pytest.log
S=$(date +%s)
timeout 900 python bench.py "$@" > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench exit $? in $(( $(date +%s) - S )) s"
tail -3 gpurun_out/${TAG}_bench.err
python - <<PY
import json
r = json.load(open("gpurun_out/${TAG}_bench.json"))
print("value %.4g sustained %.4g frac %.3g kernel_ms %.4g traffic %s (%s)" % (r["value"], r.get("sustained_value", 0), r["roofline"]["frac"], r["roofline"]["avg_kernel_ms"], r["roofline"]["traffic"], (r["roofline"]["traffic_source"] or "")[:40]))
for s in r.get("secondary", []):
    rf = s["roofline"]
    print("  %-26s N=%-6d value %.4g sustained %.4g bound %s frac %s traffic/algo %s cpu %.4g opt-in %s" % (s["env"], s["num_envs"], s["value"], s.get("sustained_value", 0), rf["bound"], rf["frac"], rf.get("traffic_over_algorithmic"), (s.get("cpu_baseline") or {}).get("value", 0), json.dumps(s.get("opt_in"))))
for k in ("opt_in", "api_step_device", "api_step_numpy", "api_step_wrapped", "cpu_baseline"):
    print(" ", k, json.dumps(r.get(k))[:260])
PY
