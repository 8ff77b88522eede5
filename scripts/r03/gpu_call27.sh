#!/bin/bash
# round-3 GPU call 27: both cooperative units on one flag set (iterative scheduler + sink), blended mass-matrix rows everywhere: MuJoCo tests + guard + lines
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_scheduler_guard.py -m gpu -q > gpurun_out/r03x_pytest.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/r03x_pytest.log
tail -3 gpurun_out/r03x_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[2])); print(sys.argv[1], "%.4g env-steps/s" % r["value"], "(%.2f s timed)" % (r["ms_per_step"] * r["steps"] * 1e-3))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
: > gpurun_out/r03x_lines.txt
for E in Ant-v5 HalfCheetah-v5 Hopper-v5 Walker2d-v5; do
  timeout 300 python bench.py --no-api --no-cpu-baseline --no-secondary --pmc off --spinup 0.2 --env $E --num-envs 65536 --inner 4 > gpurun_out/tmp_b.json 2>/dev/null; show "$E@65536" gpurun_out/tmp_b.json | tee -a gpurun_out/r03x_lines.txt
done
timeout 300 python bench.py --no-api --no-cpu-baseline --no-secondary --pmc off --spinup 0.2 --env Ant-v5 --num-envs 32768 --inner 4 > gpurun_out/tmp_b.json 2>/dev/null; show "Ant-v5@32768" gpurun_out/tmp_b.json | tee -a gpurun_out/r03x_lines.txt
PROF_STEPS=3 PROF_WARMUP=1 timeout 600 scripts/gpu_profile.sh r03_ant_coop_physics --env Ant-v5 --num-envs 65536 --inner 4 --no-secondary --pmc off > /dev/null
rm -f gpurun_out/tmp_b.json
