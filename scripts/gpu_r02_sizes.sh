#!/bin/bash
# throughput vs batch size (steady-state bench line per size)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$ROOT
: > gpurun_out/r02_sizes.txt
for spec in "CartPole-v1 128 4096 16384 65536 262144 1048576 4194304" "Ant-v5 4 8192 16384 32768 65536 131072" "Humanoid-v5 4 8192 16384 32768 65536"; do
  set -- $spec; env=$1; inner=$2; shift 2
  for n in "$@"; do
    timeout 200 python bench.py --env $env --num-envs $n --inner $inner --no-secondary --pmc off --no-cpu-baseline --no-api 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-14s N=%-8s %.4g env-steps/s  frac %s  ms_per_launch %.4g' % ('$env', '$n', r['value'], r['roofline'].get('frac'), r['ms_per_step']))" | tee -a gpurun_out/r02_sizes.txt
  done
done
