#!/bin/bash
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r02o}
export TMPDIR=/tmp PYTHONPATH=$ROOT
timeout 900 python -m pytest tests/test_gpu_wrappers.py -m gpu -q -x > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/${TAG}_pytest.log
for out in torch numpy; do for mode in fused standalone; do python scripts/wrappers_bench.py --mode $mode --output $out 2>/dev/null | tail -1; done; done | tee gpurun_out/${TAG}_wrappers.txt
python scripts/wrappers_bench.py --mode fused --output torch --num-envs 1048576 --steps 100 2>/dev/null | tail -1 | tee -a gpurun_out/${TAG}_wrappers.txt
python scripts/wrappers_bench.py --mode standalone --output torch --num-envs 1048576 --steps 100 2>/dev/null | tail -1 | tee -a gpurun_out/${TAG}_wrappers.txt
cd /tmp
for mode in fused standalone; do
  rm -rf /tmp/prof_$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -o w -- python $ROOT/scripts/wrappers_bench.py --mode $mode --output torch --steps 200 > /tmp/prof_$mode.log 2>&1
  f=$(find /tmp/prof_$mode -name "*kernel_stats.csv" | head -1)
  echo "== $mode"; cp "$f" $ROOT/gpurun_out/${TAG}_${mode}_kernel_stats.csv 2>/dev/null; head -6 "$f" | cut -c1-200
done
