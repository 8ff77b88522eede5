#!/bin/bash
# round-3 GPU call 19: finer phase marks (MJX_DETAIL = 2 crb, 3 RNE, 4 kinematics) of the cooperative kernel, Humanoid PGS and Ant
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
for R in humanoid ant; do
  N=32768; [ $R = ant ] && N=65536
  for V in d2 d3 d4; do
    echo "=== $V $R warm=3"; COOP_WARM=3 timeout 120 scripts/phase_$V.bin $R $N | tee -a gpurun_out/r03o_phase_$V.txt | grep -v "fingerprint\|pgs:\|solver:"
  done
done
