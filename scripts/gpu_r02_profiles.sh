#!/bin/bash
# Round-2 profile set: phase timing of the cooperative kernels + rocprofv3 summaries (kernel trace + PMC passes) of the bench workloads.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
for W in 3 40; do for M in humanoid humanoid-newton ant; do echo "== $M warm=$W (COOP_WARM: env-steps taken before the timed launch; 40 = fallen robots, many contacts)"; COOP_WARM=$W timeout 300 scripts/coop_phase_bench.bin $M $([ $M = ant ] && echo 65536 || echo 32768) 2>&1 | tail -16; done; done > gpurun_out/r02_coop_phases.txt 2>&1
grep -E "^==|env-steps/s|total" gpurun_out/r02_coop_phases.txt
COMMON="--no-secondary --pmc off --sustained 0"
PROF_STEPS=default scripts/gpu_profile.sh r02_cartpole_rollout $COMMON
PROF_STEPS=default scripts/gpu_profile.sh r02_ant_coop_physics $COMMON --env Ant-v5 --num-envs 65536 --inner 4
PROF_STEPS=default scripts/gpu_profile.sh r02_humanoid_pgs_coop_physics $COMMON --env Humanoid-v5 --num-envs 32768 --inner 4
PROF_STEPS=default scripts/gpu_profile.sh r02_humanoid_newton_coop_physics $COMMON --env Humanoid-v5 --num-envs 32768 --inner 4 --env-kwargs '{"solver":"Newton"}'
ls -la gpurun_out/r02_*.txt
