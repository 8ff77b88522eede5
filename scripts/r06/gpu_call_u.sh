#!/bin/bash
# round 6, call U: randomised soak of the closing build against the oracle (classic + ToyText incl. the on-device policy; the shared-generator CartPole; MuJoCo within 2e-8)
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 400 python scripts/r06/gpu_soak.py 240 11 2>&1 | tail -3 | tee gpurun_out/r06_soak.txt
timeout 200 python scripts/r06/gpu_soak.py shared 60 2>&1 | tail -2 | tee -a gpurun_out/r06_soak.txt
timeout 300 python scripts/r06/gpu_soak.py mujoco 120 2>&1 | tail -2 | tee -a gpurun_out/r06_soak.txt
