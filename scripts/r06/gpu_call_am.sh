#!/bin/bash
# round 6, call AM: three switches of the two-role rollout measured again now that each role has its own loop (A/B only): CartPole's aux role deriving the flags,
# the action of the next step requested a step ahead (everywhere), chunks of 4 steps
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
L=gymnasium_amd/csrc/libmi355env
timeout 900 python scripts/ab_bench.py --libs shipped=${L}.so cp_derive=${L}_cpderive.so act_ahead=${L}_actahead.so chunk4=${L}_chunk4.so --envs CartPole-v1:65536:128 Pendulum-v1:65536:128 MountainCar-v0:65536:128 MountainCarContinuous-v0:65536:128 --rounds 3 --out gpurun_out/r06_duo_switches_after_role_loops_ab.txt
