#!/bin/bash
# round-3 GPU call 24: how much of the waiting inside a wavefront is predictable?  The harness regroups the environments between launches by the solver
# work of the previous launch (COOP_SORT=1; built with -DMJX_COUNT_WORK -DNO_PHASE_TIMING) and times the kernel alone.
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
for R in ant cheetah; do
  for S in 0 1; do
    echo "=== $R sort=$S"
    if [ $S = 1 ]; then export COOP_SORT=1; else unset COOP_SORT; fi
    COOP_WARM=12 COOP_TIMED=10 timeout 300 scripts/phase_work.bin $R 65536 | tee -a gpurun_out/r03t_sort.txt | grep "solver passes\|per forward pass\|kernel time\|fingerprint"
  done
done
