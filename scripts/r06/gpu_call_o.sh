#!/bin/bash
# round 6, call O: role timing after Pendulum's second cut
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
L=gymnasium_amd/csrc/libmi355env
for env in Pendulum-v1 MountainCarContinuous-v0 MountainCar-v0; do
  echo "## timing $env"; MI355ENV_LIBRARY=${L}_timing.so timeout 300 python scripts/r04/duo_timing.py $env 2>&1 | grep "duo timing" | tail -8 | sort
done > gpurun_out/r06_duo_timing_after.txt 2>&1
cat gpurun_out/r06_duo_timing_after.txt
