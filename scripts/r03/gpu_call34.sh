#!/bin/bash
# round-3 GPU call 34: phase tables of the final cooperative kernels (scripts/coop_phase_bench.hip built with the product's flags)
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
: > gpurun_out/r03_coop_phases.txt
for spec in "ant 65536 3" "ant 65536 40" "cheetah 65536 40" "humanoid 32768 3" "humanoid 32768 40" "humanoid-newton 32768 40"; do
  set -- $spec
  echo "##### $1, $2 envs, COOP_WARM=$3" | tee -a gpurun_out/r03_coop_phases.txt
  COOP_WARM=$3 timeout 200 scripts/phase_final.bin $1 $2 | tee -a gpurun_out/r03_coop_phases.txt | grep "env-steps\|total"
done
