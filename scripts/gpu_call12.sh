#!/bin/bash
set -u
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
timeout 600 python bench.py --no-api --no-cpu-baseline --no-secondary > gpurun_out/r03h_bench.json 2>/dev/null
python -c "
import json; r=json.load(open('gpurun_out/r03h_bench.json')); rf=r['roofline']; print('value %.4g frac %.3f' % (r['value'], rf['frac'])); print(json.dumps(rf.get('issue_bound'), indent=1))"
