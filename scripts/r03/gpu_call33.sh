#!/bin/bash
# round-3 final closing validation (after calls 24-32: one flag set for both cooperative units, blended mass-matrix rows everywhere)
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r03ab_pytest.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/r03ab_pytest.log
tail -4 gpurun_out/r03ab_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python scripts/r03/wrapped_mujoco_step.py Ant-v5 65536 2>&1 | tail -1 | tee gpurun_out/r03_wrapped_mujoco_step.txt
timeout 300 python scripts/r03/wrapped_mujoco_step.py Humanoid-v5 32768 2>&1 | tail -1 | tee -a gpurun_out/r03_wrapped_mujoco_step.txt
timeout 900 python bench.py > gpurun_out/r03_final_bench.json 2> gpurun_out/r03_final_bench.err; echo "bench exit $?"
python - <<'PY'
import json
r = json.load(open("gpurun_out/r03_final_bench.json"))
print("value %.4g frac %.3f ms %.4f steps %d" % (r["value"], r["roofline"]["frac"], r["ms_per_step"], r["steps"]), "sustained %.4g" % r.get("sustained_value", 0), "traffic/alg", r["roofline"].get("traffic_over_algorithmic"))
print("cpu_baseline %.4g cores %s" % (r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"]))
for s in r.get("secondary", []):
    rf = s["roofline"]
    print(" ", s["env"], s["num_envs"], "%.4g" % s["value"], "opt %.4g" % s.get("opt_in", {}).get("value", 0), "frac %.3g" % (rf.get("frac") or 0), "traffic/alg", rf.get("traffic_over_algorithmic"), "cpu %.4g" % s.get("cpu_baseline", {}).get("value", 0))
PY
tail -2 gpurun_out/r03_final_bench.err
PROF_STEPS=default timeout 600 scripts/gpu_profile.sh r03_cartpole_rollout --no-secondary --pmc off > /dev/null
PROF_STEPS=3 PROF_WARMUP=1 timeout 600 scripts/gpu_profile.sh r03_ant_coop_physics --env Ant-v5 --num-envs 65536 --inner 4 --no-secondary --pmc off > /dev/null
PROF_STEPS=2 PROF_WARMUP=1 timeout 600 scripts/gpu_profile.sh r03_humanoid_pgs_coop_physics --env Humanoid-v5 --num-envs 32768 --inner 4 --no-secondary --pmc off > /dev/null
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[2])); print(sys.argv[1], "%.4g env-steps/s" % r["value"], "(%.2f s timed)" % (r["ms_per_step"] * r["steps"] * 1e-3))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
: > gpurun_out/r03_allenvs.txt
for E in Ant-v5 HalfCheetah-v5 Hopper-v5 Walker2d-v5 InvertedPendulum-v5 InvertedDoublePendulum-v5 Reacher-v5 Swimmer-v5 Pusher-v5; do
  timeout 300 python bench.py --no-api --no-cpu-baseline --no-secondary --pmc off --spinup 0.2 --env $E --num-envs 65536 --inner 4 > gpurun_out/tmp_b.json 2>/dev/null; show "$E@65536" gpurun_out/tmp_b.json | tee -a gpurun_out/r03_allenvs.txt
done
for E in Ant-v5 Humanoid-v5 HumanoidStandup-v5; do
  timeout 300 python bench.py --no-api --no-cpu-baseline --no-secondary --pmc off --spinup 0.2 --env $E --num-envs 32768 --inner 4 > gpurun_out/tmp_b.json 2>/dev/null; show "$E@32768" gpurun_out/tmp_b.json | tee -a gpurun_out/r03_allenvs.txt
done
rm -f gpurun_out/tmp_b.json
