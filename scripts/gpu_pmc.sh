#!/bin/bash
# One extra PMC pass for a bench.py configuration:  scripts/gpu_pmc.sh <tag> "<counters>" [bench args...]
set -u
TAG=$1; CTRS=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
CMD="python $ROOT/bench.py --steps ${PROF_STEPS:-3} --warmup ${PROF_WARMUP:-1} --no-api --no-cpu-baseline $*"
rocprofv3 --pmc $CTRS --kernel-trace -d "$OUT/${TAG}_pmc" -- $CMD > "$OUT/${TAG}_pmc.log" 2>&1
cd "$ROOT"
python scripts/rocpd_summary.py --pmc "$OUT/${TAG}_pmc" --cmd "$CMD" -o "$OUT/${TAG}.txt" | grep "${FILTER:-mj_physics}" | sed 's/(anonymous namespace):://g' | awk -F'|' '{print $2, $3}'
rm -rf "$OUT/${TAG}_pmc"
