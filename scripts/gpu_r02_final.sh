#!/bin/bash
# Round-2 closing run: full GPU test-suite (single process), the default bench line, then the profile set.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$ROOT
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02_final_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r02_final_pytest.log
S=$(date +%s); timeout 900 python bench.py > gpurun_out/r02_final_bench.json 2> gpurun_out/r02_final_bench.err; echo "bench exit $? in $(( $(date +%s) - S )) s"
python -c "
import json; r=json.load(open('gpurun_out/r02_final_bench.json'))
print('value %.4g frac %.3g kernel_ms %.4g K=%d' % (r['value'], r['roofline']['frac'], r['roofline']['avg_kernel_ms'], r['steps']))
for s in r['secondary']: print('  %-26s N=%-6d value %.4g frac %s traffic/algo %s opt-in %s' % (s['env'], s['num_envs'], s['value'], s['roofline']['frac'], s['roofline'].get('traffic_over_algorithmic'), (s.get('opt_in') or {}).get('value')))
"
scripts/gpu_r02_profiles.sh > gpurun_out/r02_final_profiles.log 2>&1; tail -8 gpurun_out/r02_final_profiles.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()"
