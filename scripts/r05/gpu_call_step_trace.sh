#!/bin/bash
# round 5: kernel trace of the per-launch step() path (step_kernel's own duration vs the 5 us per step the API measures)
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05_step_api_stats -- python $R/scripts/bench_extras.py --out $R/gpurun_out/r05_step_api.json --api-only > $R/gpurun_out/r05_step_api_stats.log 2>&1
cd $R && python scripts/rocpd_summary.py --stats gpurun_out/r05_step_api_stats --cmd "python scripts/bench_extras.py --api-only" -o gpurun_out/r05_step_api_kernel_trace.txt > /dev/null 2>&1
rm -rf gpurun_out/r05_step_api_stats
head -12 gpurun_out/r05_step_api_kernel_trace.txt | cut -c1-260
python -c "
import json; d=json.load(open('gpurun_out/r05_step_api.json')); print({k: {kk: (round(vv,3) if isinstance(vv,float) else vv) for kk,vv in v.items() if 'us_' in kk or kk=='roofline_frac'} for k,v in d.items() if k.startswith('api_')})"
