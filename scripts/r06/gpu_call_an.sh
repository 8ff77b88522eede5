#!/bin/bash
# round 6, call AN (closing after Pendulum's fmod / powf cuts and the loop per role): whole GPU suite, smoke(), the randomised soak against the oracle,
# rocprofv3 summaries of the four two-role rollouts, the driver's bench command
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06_final_pytest_gpu.log 2>&1; tail -3 gpurun_out/r06_final_pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v Warn | tail -3
timeout 300 python scripts/r06/gpu_soak.py 100 17 2>&1 | tail -3 | tee gpurun_out/r06_soak_role_loops.txt
PROF_STEPS=default timeout 900 scripts/gpu_profile.sh r06_cartpole_rollout > /dev/null 2>&1; grep -c rollout_duo gpurun_out/r06_cartpole_rollout.txt
for e in Pendulum-v1 MountainCar-v0 MountainCarContinuous-v0; do PROF_STEPS=20 PROF_WARMUP=3 timeout 600 scripts/gpu_profile.sh r06_${e}_rollout --env $e > /dev/null 2>&1; grep -c "rollout" gpurun_out/r06_${e}_rollout.txt; done
SECONDS=0; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_line_driver_cmd.json 2> gpurun_out/r06_an_bench.err; echo "bench.py exit $? after $SECONDS s"; cut -c1-300 gpurun_out/r06_bench_line_driver_cmd.json
cp gpurun_out/bench_full.json gpurun_out/r06_bench_full.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_bench_full.json'))
print(d.get('traffic_passes'))
for l in d['secondary']:
    r=l.get('roofline',{})
    print(l.get('env'), l.get('num_envs'), l.get('regime','')[:10], '%.4g'%l.get('value',0), 'frac %.3f'%r.get('frac',0), 'tr/algo', r.get('traffic_over_algorithmic'), 'verified', (l.get('verified') or {}).get('ok'))
print({k:v for k,v in d['headline'].items() if k not in ('hbm_frac','opt_in','verified')})
PY
