#!/bin/bash
# round 5: rocprofv3 summaries of the cooperative MuJoCo kernels (kernel trace + FETCH / WRITE / SQ counters), random-policy regime and Humanoid on the ground
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
PROF_STEPS=3 PROF_WARMUP=1 timeout 900 scripts/gpu_profile.sh r05_ant_coop_physics --env Ant-v5 --num-envs 32768 --inner 4 > /dev/null 2>&1; grep -c "mj_physics" gpurun_out/r05_ant_coop_physics.txt
PROF_STEPS=2 PROF_WARMUP=1 timeout 900 scripts/gpu_profile.sh r05_humanoid_pgs_coop_physics --env Humanoid-v5 --num-envs 32768 --inner 2 > /dev/null 2>&1; grep -c "mj_physics" gpurun_out/r05_humanoid_pgs_coop_physics.txt
PROF_STEPS=2 PROF_WARMUP=40 timeout 900 scripts/gpu_profile.sh r05_humanoid_pgs_on_the_ground --env Humanoid-v5 --num-envs 32768 --inner 4 --env-kwargs '{"terminate_when_unhealthy":false}' > /dev/null 2>&1; grep -c "mj_physics" gpurun_out/r05_humanoid_pgs_on_the_ground.txt
grep "mj_physics" gpurun_out/r05_humanoid_pgs_on_the_ground.txt | head -3 | cut -c1-60,150-260
