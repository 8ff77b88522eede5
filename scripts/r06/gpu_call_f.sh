#!/bin/bash
# round 6, call F: the aux-fed reset FIFO A/B; MuJoCo window verification with a thread-safe oracle; the driver's bench command end to end (batched live traffic passes)
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_rollout_roles.py tests/test_gpu_parity.py tests/test_gpu_bench_contract.py -x -q -m gpu > gpurun_out/r06_f_tests.log 2>&1; tail -3 gpurun_out/r06_f_tests.log
timeout 1500 python scripts/ab_bench.py --libs c8=gymnasium_amd/csrc/libmi355env_c8.so derive=gymnasium_amd/csrc/libmi355env_derive.so fifo=gymnasium_amd/csrc/libmi355env.so \
   --envs CartPole-v1:65536:128 MountainCar-v0:65536:128 --rounds 3 --out gpurun_out/r06_duo_diet_ab.txt
for e in Ant-v5 Humanoid-v5; do
  timeout 600 python bench.py --env $e --num-envs 32768 --inner 4 --steps 10 --warmup 3 --no-secondary --pmc off --no-cpu-baseline 2>gpurun_out/r06_f_$e.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], json.dumps(d['verified']))"
done
/usr/bin/time -v timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_f_bench_line.json 2> gpurun_out/r06_f_bench.err; grep -E "Elapsed|Exit" gpurun_out/r06_f_bench.err; cut -c1-300 gpurun_out/r06_f_bench_line.json
cp gpurun_out/bench_full.json gpurun_out/r06_f_bench_full.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_f_bench_full.json'))
print(d.get('traffic_passes'))
for l in d['secondary']:
    r=l.get('roofline',{})
    print(l.get('env'), l.get('num_envs'), l.get('regime','')[:20], '%.4g'%l.get('value',0), 'frac %.3f'%r.get('frac',0), 'traffic', r.get('traffic'), (r.get('traffic_source') or '')[:40], 'verified', (l.get('verified') or {}).get('ok'), (l.get('verified') or {}).get('max_abs_diff'))
print(d.get('headline'))
PY
