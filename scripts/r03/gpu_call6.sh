#!/bin/bash
# round-3 GPU call 6: full GPU suite on the build with Acrobot on the builtin-fma policy; Acrobot determinism loop; classic bench lines + SQ counters
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r03d_pytest.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/r03d_pytest.log
tail -8 gpurun_out/r03d_pytest.log
for rep in 1 2 3; do echo "=== determinism rep $rep"; timeout 120 python scripts/debug_acrobot3.py 2>&1 | grep "^T=" | grep -v "bad lanes \[\], given-actions t=0 bad \[\], sampled vs given per t: \[0\(, 0\)*\]$" ; done; echo "(lines above = launches that differed; none expected)"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[2])); print(sys.argv[1], "value %.4g" % r["value"], "frac %.3g" % r["roofline"]["frac"], "kernel_ms %.4g" % r["roofline"]["avg_kernel_ms"], "opt_in %.4g" % r.get("opt_in", {}).get("value", 0))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for E in CartPole-v1 Pendulum-v1 Acrobot-v1; do
  timeout 300 python bench.py --no-api --no-cpu-baseline --no-secondary --pmc off --env $E > gpurun_out/r03d_bench_$E.json 2>> gpurun_out/r03d_bench.err; show $E gpurun_out/r03d_bench_$E.json
done
for E in Pendulum-v1 Acrobot-v1; do
  PROF_STEPS=30 PROF_WARMUP=3 timeout 600 scripts/gpu_profile.sh r03d_${E}_rollout --env $E --no-secondary --pmc off > /dev/null
done
grep -h "SQ_INSTS_VALU\|SQ_INSTS_SALU\|SQ_WAVE_CYCLES\|SQ_WAVES \|SQ_INSTS_LDS" gpurun_out/r03d_*_rollout.txt | grep "ExactMath" | cut -c1-60,200-
