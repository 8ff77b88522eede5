#!/bin/bash
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r02n}
export TMPDIR=/tmp PYTHONPATH=$ROOT
cd /tmp
for mode in fused standalone; do
  rm -rf /tmp/prof_$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -o w -- python $ROOT/scripts/wrappers_bench.py --mode $mode --output torch --steps 200 > /tmp/prof_$mode.log 2>&1; tail -3 /tmp/prof_$mode.log
  f=$(find /tmp/prof_$mode -name "*kernel_stats.csv" | head -1)
  echo "== $mode: $f"; cp "$f" $ROOT/gpurun_out/${TAG}_${mode}_kernel_stats.csv 2>/dev/null; head -15 "$f"
done
