#!/bin/bash
# round 6, call S: the rocprofv3 summary of MountainCar's rollout once more (call R's pass ran into its 600 s limit)
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
PROF_STEPS=20 PROF_WARMUP=3 timeout 1500 scripts/gpu_profile.sh r06_MountainCar-v0_rollout --env MountainCar-v0 > gpurun_out/r06_s.log 2>&1; tail -3 gpurun_out/r06_s.log; grep -c "rollout" gpurun_out/r06_MountainCar-v0_rollout.txt
