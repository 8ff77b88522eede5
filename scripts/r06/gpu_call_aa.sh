#!/bin/bash
# round 6, call AA: where the one-role rollout kernel stands after the round's instruction diet (MI355ENV_ROLLOUT_DUO=0 against the two-role kernel)
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
L=gymnasium_amd/csrc/libmi355env.so
timeout 900 python scripts/ab_bench.py --libs duo=${L} one=${L}@MI355ENV_ROLLOUT_DUO=0 --envs CartPole-v1:65536:128 MountainCar-v0:65536:128 MountainCarContinuous-v0:65536:128 Pendulum-v1:65536:128 --rounds 3 --out gpurun_out/r06_one_role_after_diet_ab.txt
