#!/bin/bash
# round-3 GPU call 37: did the 16-lane kernels change speed when the Board grew by the (unused, one-double) block array of MJX_PGS_MORE_BLOCKS?
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[2])); print(sys.argv[1], "value %.4g" % r["value"], "kernel_ms %.4g" % r["roofline"]["avg_kernel_ms"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for rep in 1 2; do
  for V in old16 product; do
    LL=gymnasium_amd/csrc/libmi355env_$V.so; [ $V = product ] && LL=gymnasium_amd/csrc/libmi355env.so
    MI355ENV_LIBRARY=$PWD/$LL timeout 300 python bench.py --no-api --no-cpu-baseline --no-secondary --pmc off --spinup 0.2 --env Ant-v5 --num-envs 65536 --inner 4 > gpurun_out/r03ae_tmp.json 2>/dev/null
    show "Ant-v5 $V rep$rep" gpurun_out/r03ae_tmp.json
  done
done
