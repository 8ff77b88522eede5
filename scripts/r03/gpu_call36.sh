#!/bin/bash
# round-3 GPU call 36: last validation of the final library (the Board grew by one block array in call 35: every cooperative unit was recompiled)
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_scheduler_guard.py -m gpu -q > gpurun_out/r03ad_pytest.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/r03ad_pytest.log
tail -3 gpurun_out/r03ad_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for spec in "Ant-v5 65536" "Humanoid-v5 32768"; do
  set -- $spec
  timeout 300 python bench.py --no-api --no-cpu-baseline --no-secondary --pmc off --spinup 0.2 --env $1 --num-envs $2 --inner 4 2>/dev/null | python -c "import json,sys; r=json.load(sys.stdin); print('$1', '%.4g env-steps/s' % r['value'])"
done
