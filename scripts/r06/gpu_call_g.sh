#!/bin/bash
# round 6, call G: Pendulum as a balanced two-role rollout (reward on the aux role); CartPole back on the flag-word protocol, MountainCar on the state words; the driver's command end to end
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_rollout_roles.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r06_g_tests.log 2>&1; tail -3 gpurun_out/r06_g_tests.log
timeout 1500 python scripts/ab_bench.py --libs base=gymnasium_amd/csrc/libmi355env_base.so c8=gymnasium_amd/csrc/libmi355env_c8.so product=gymnasium_amd/csrc/libmi355env.so \
   --envs CartPole-v1:65536:128 Pendulum-v1:65536:128 MountainCar-v0:65536:128 --rounds 3 --out gpurun_out/r06_duo_diet_ab.txt
SECONDS=0; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_g_bench_line.json 2> gpurun_out/r06_g_bench.err; echo "bench.py exit $? after $SECONDS s"; cut -c1-200 gpurun_out/r06_g_bench_line.json
cp gpurun_out/bench_full.json gpurun_out/r06_g_bench_full.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_g_bench_full.json'))
print(d.get('traffic_passes'))
for l in d['secondary']:
    r=l.get('roofline',{})
    print(l.get('env'), l.get('num_envs'), l.get('regime','')[:10], '%.4g'%l.get('value',0), 'frac %.3f'%r.get('frac',0), 'tr/algo', r.get('traffic_over_algorithmic'), 'f64peak', r.get('frac_of_f64_peak'), 'verified', (l.get('verified') or {}).get('ok'))
print({k:v for k,v in d['headline'].items() if k not in ('hbm_frac','opt_in','verified')})
PY
