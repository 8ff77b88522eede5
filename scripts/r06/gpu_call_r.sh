#!/bin/bash
# round 6, call R (closing, after the role re-balancing and the limb-wise generator step): smoke(), the rocprofv3 summaries of every bench configuration (kernel trace + FETCH / WRITE / SQ passes), the driver's bench command
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v Warn | tail -3
PROF_STEPS=default timeout 900 scripts/gpu_profile.sh r06_cartpole_rollout > /dev/null 2>&1; grep -c rollout_duo gpurun_out/r06_cartpole_rollout.txt
for e in Pendulum-v1 Acrobot-v1 MountainCarContinuous-v0 MountainCar-v0; do PROF_STEPS=20 PROF_WARMUP=3 timeout 600 scripts/gpu_profile.sh r06_${e}_rollout --env $e > /dev/null 2>&1; grep -c "rollout" gpurun_out/r06_${e}_rollout.txt; done
for e in FrozenLake-v1 Taxi-v4 Blackjack-v1; do t=$(echo $e | tr 'A-Z' 'a-z' | sed 's/-v.//'); PROF_STEPS=20 PROF_WARMUP=3 timeout 600 scripts/gpu_profile.sh r06_${t}_rollout --env $e > /dev/null 2>&1; grep -c "tab_rollout" gpurun_out/r06_${t}_rollout.txt; done
PROF_STEPS=3 PROF_WARMUP=1 timeout 900 scripts/gpu_profile.sh r06_ant_coop_physics --env Ant-v5 --num-envs 32768 --inner 4 > /dev/null 2>&1; grep -c "mj_physics" gpurun_out/r06_ant_coop_physics.txt
PROF_STEPS=2 PROF_WARMUP=1 timeout 900 scripts/gpu_profile.sh r06_humanoid_pgs_coop_physics --env Humanoid-v5 --num-envs 32768 --inner 2 > /dev/null 2>&1; grep -c "mj_physics" gpurun_out/r06_humanoid_pgs_coop_physics.txt
PROF_STEPS=2 PROF_WARMUP=40 timeout 900 scripts/gpu_profile.sh r06_humanoid_pgs_on_the_ground --env Humanoid-v5 --num-envs 32768 --inner 4 --env-kwargs '{"terminate_when_unhealthy":false}' > /dev/null 2>&1; grep -c "mj_physics" gpurun_out/r06_humanoid_pgs_on_the_ground.txt
export TMPDIR=/tmp; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/r06_policy_trace -- python $OLDPWD/scripts/bench_extras.py --out $OLDPWD/gpurun_out/r06_h_policy.json --policy-only > $OLDPWD/gpurun_out/r06_h_policy.log 2>&1); python scripts/rocpd_summary.py --stats gpurun_out/r06_policy_trace --cmd "python scripts/bench_extras.py --policy-only" -o gpurun_out/r06_step_api_kernel_trace.txt > /dev/null; rm -rf gpurun_out/r06_policy_trace; tail -1 gpurun_out/r06_h_policy.log | cut -c1-300
SECONDS=0; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_line_driver_cmd.json 2> gpurun_out/r06_h_bench.err; echo "bench.py exit $? after $SECONDS s"; cut -c1-200 gpurun_out/r06_bench_line_driver_cmd.json
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
