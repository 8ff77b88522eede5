#!/bin/bash
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
for W in 3 40; do for M in humanoid humanoid-newton; do
  echo "== $M warm=$W"; COOP_WARM=$W timeout 300 scripts/coop_phase_bench.bin $M 32768 2>&1 | tail -16
done; done > gpurun_out/r02d_phases.txt 2>&1
cat gpurun_out/r02d_phases.txt
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r02d_all.log 2>&1; echo "single-process suite exit $?"; tail -4 gpurun_out/r02d_all.log
S=$(date +%s); timeout 900 python bench.py --no-secondary > gpurun_out/r02d_bench.json 2> gpurun_out/r02d_bench.err; echo "bench exit $? in $(( $(date +%s) - S )) s"
python -c "
import json; r=json.load(open('gpurun_out/r02d_bench.json')); print('value %.4g sustained %.4g' % (r['value'], r['sustained_value'])); print(json.dumps(r['api_step_numpy'])); print(json.dumps(r['api_step_device']))"
