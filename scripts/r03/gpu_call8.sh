#!/bin/bash
# round-3 GPU call 8: full GPU suite after the glue (np.linalg.norm) change; smoke; the driver's bench command
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r03f_pytest.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/r03f_pytest.log
tail -4 gpurun_out/r03f_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03f_bench_driver.json 2> gpurun_out/r03f_bench_driver.err; echo "bench exit $?"
python - <<'PY'
import json
r = json.load(open("gpurun_out/r03f_bench_driver.json"))
print("value %.4g frac %.3f ms %.4f" % (r["value"], r["roofline"]["frac"], r["ms_per_step"]), "sustained %.4g" % r.get("sustained_value", 0), "traffic", r["roofline"].get("traffic_over_algorithmic"))
print("cpu_baseline %.4g cores %s" % (r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"]))
for s in r.get("secondary", []):
    rf = s["roofline"]
    print(s["env"], s["num_envs"], "%.4g" % s["value"], "opt %.4g" % s.get("opt_in", {}).get("value", 0), "frac %.3g" % (rf.get("frac") or 0), "traffic/alg", rf.get("traffic_over_algorithmic"), (rf.get("traffic_source") or "")[:40], "cpu %.4g" % s.get("cpu_baseline", {}).get("value", 0))
PY
tail -2 gpurun_out/r03f_bench_driver.err
