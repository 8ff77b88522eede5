#!/bin/bash
# round-3 GPU call 1: the whole GPU suite on the current build + the driver's bench command + the default bench (steady state)
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r03a_pytest.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/r03a_pytest.log
tail -15 gpurun_out/r03a_pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03a_bench_driver.json 2> gpurun_out/r03a_bench_driver.err; echo "bench(driver flags) exit $?"
python - <<'PY'
import json
r = json.load(open("gpurun_out/r03a_bench_driver.json"))
print("value %.4g frac %.3f ms %.4f" % (r["value"], r["roofline"]["frac"], r["ms_per_step"]), "sustained %.4g" % r.get("sustained_value", 0))
print("cpu_baseline", r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"], r["cpu_baseline"].get("host_cpu_count"))
for s in r.get("secondary", []):
    print(s["env"], s["num_envs"], "%.4g" % s["value"], "opt %.4g" % s.get("opt_in", {}).get("value", 0), "frac", s["roofline"].get("frac"), "cpu", s.get("cpu_baseline", {}).get("value"), s.get("cpu_baseline", {}).get("cores"))
for k in ("api_step_device", "api_step_numpy", "api_step_wrapped"):
    print(k, r.get(k))
PY
tail -3 gpurun_out/r03a_bench_driver.err
nproc; free -g | head -2
