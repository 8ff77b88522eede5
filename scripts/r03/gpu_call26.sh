#!/bin/bash
# round-3 GPU call 26: physics16.hip on physics32.hip's flag set (iterative scheduler + sink) with the blended mass-matrix rows for the Ant as well
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
L=$PWD/gymnasium_amd/csrc/libmi355env_sink16b.so
timeout 600 python scripts/r03/guard_variant.py $L 2>&1 | tail -6 | tee gpurun_out/r03w_guard.txt
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[2])); print(sys.argv[1], "value %.4g" % r["value"], "kernel_ms %.4g" % r["roofline"]["avg_kernel_ms"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for rep in 1 2; do
  for V in product sink16 sink16b; do
    LL=gymnasium_amd/csrc/libmi355env_$V.so; [ $V = product ] && LL=gymnasium_amd/csrc/libmi355env.so
    MI355ENV_LIBRARY=$PWD/$LL timeout 300 python bench.py --no-api --no-cpu-baseline --no-secondary --pmc off --spinup 0.2 --env Ant-v5 --num-envs 65536 --inner 4 > gpurun_out/r03w_tmp.json 2>/dev/null
    show "Ant-v5 $V rep$rep" gpurun_out/r03w_tmp.json
  done
done
