#!/bin/bash
set -u
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wrappers.py -m gpu -q 2>&1 | tail -3
timeout 300 python bench.py --no-api --no-cpu-baseline --no-secondary --pmc off --env Pendulum-v1 > gpurun_out/r03g_bench_Pendulum-v1.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/r03g_bench_Pendulum-v1.json')); print('Pendulum %.4g frac %.3f opt_in %.4g' % (r['value'], r['roofline']['frac'], r['opt_in']['value']))"
PROF_STEPS=30 PROF_WARMUP=3 timeout 600 scripts/gpu_profile.sh r03g_Pendulum-v1_rollout --env Pendulum-v1 --no-secondary --pmc off > /dev/null
grep -h "SQ_INSTS_VALU\|SQ_INSTS_SALU" gpurun_out/r03g_Pendulum-v1_rollout.txt | grep "ExactMath" | cut -c1-50,190-
