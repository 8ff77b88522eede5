#!/bin/bash
# round 6, call Q: the generator's step written limb by limb (pcg64_dev.h pcg_muladd) -- it is on the env role (the reset-state refill) and in every reset / seeding kernel:
# full GPU suite, then A/B against commit 0c58991's build
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
L=gymnasium_amd/csrc/libmi355env
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06_q_pytest_gpu.log 2>&1; tail -3 gpurun_out/r06_q_pytest_gpu.log
timeout 1200 python scripts/ab_bench.py --libs h=${L}_h.so limbstep=${L}.so --envs CartPole-v1:65536:128 MountainCar-v0:65536:128 MountainCarContinuous-v0:65536:128 Pendulum-v1:65536:128 Acrobot-v1:65536:128 Blackjack-v1:65536:128 --rounds 3 --out gpurun_out/r06_limb_step_ab.txt
