#!/bin/bash
# round 5, GPU call D: the pipelined PGS sweep (MJX_PGS_PIPELINE) -- parity tests, then A/B against the sequential form (libmi355env_seq.so) in both contact regimes
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_scheduler_guard.py tests/test_mujoco_reference_pins.py -q -m gpu -k "umanoid or guard or pins or statistics" > gpurun_out/r05d_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r05d_pytest.log | cut -c1-200
python scripts/ab_bench.py --libs seq=gymnasium_amd/csrc/libmi355env_seq.so pipelined=gymnasium_amd/csrc/libmi355env.so --envs Humanoid-v5:32768:4 HumanoidStandup-v5:32768:4 --rounds 2 --out gpurun_out/r05d_ab_standing.txt
python scripts/ab_bench.py --libs seq=gymnasium_amd/csrc/libmi355env_seq.so pipelined=gymnasium_amd/csrc/libmi355env.so --envs Humanoid-v5:32768:4 --rounds 2 --env-kwargs '{"terminate_when_unhealthy": false}' --warmup 40 --out gpurun_out/r05d_ab_ground.txt
