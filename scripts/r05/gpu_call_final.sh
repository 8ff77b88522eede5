#!/bin/bash
# round 5, closing GPU call: what the driver runs at round end -- the whole -m gpu suite, smoke(), and bench.py with the driver's own flags
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
bash scripts/gpu_call.sh r05final tests -- smoke
S=$(date +%s); python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05final_bench.json 2> gpurun_out/r05final_bench.err; echo "bench exit $? in $(( $(date +%s) - S )) s; line $(wc -c < gpurun_out/r05final_bench.json) bytes"
cp gpurun_out/bench_full.json gpurun_out/r05final_bench_full.json
python - <<'PY'
import json
r = json.load(open("gpurun_out/r05final_bench.json"))
print("value %.4g frac %.3f kernel_ms %.4f verified %s sha %s" % (r["value"], r["roofline"]["frac"], r["roofline"]["avg_kernel_ms"], r["verified"]["ok"], r["output_sha256"][:12]))
print(json.dumps(r["secondary"]))
PY
