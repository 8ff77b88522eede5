#!/bin/bash
# round 6, call L: where each role of the two-role rollout spends its cycles (-DMI_DUO_TIMING builds of the current sources and of them with the three changes off)
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
L=gymnasium_amd/csrc/libmi355env
for lib in timing timing_old; do for env in CartPole-v1 MountainCar-v0 MountainCarContinuous-v0 Pendulum-v1; do
  echo "## $lib $env"; MI355ENV_LIBRARY=${L}_${lib}.so timeout 300 python scripts/r04/duo_timing.py $env 2>&1 | grep "duo timing" | tail -8
done; done > gpurun_out/r06_duo_timing.txt 2>&1
cat gpurun_out/r06_duo_timing.txt
