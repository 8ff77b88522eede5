#!/bin/bash
set -u
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_bench_contract.py -m gpu -q 2>&1 | tail -2
for S in 0 0.5; do
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-api --no-cpu-baseline --pmc off --spinup $S | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('spinup $S: value %.4g frac %.3f ms %.4f sustained %.4g' % (r['value'], r['roofline']['frac'], r['ms_per_step'], r.get('sustained_value',0)), r['clock_spinup'])"
done
