#!/bin/bash
# round 5, GPU call E: PGS sweep variants against the shipped sequential form -- pf (next contact's Jacobian column requested early), edge (the four edge rows
# relaxed in edge space: a 4-operation dependent chain per edge instead of ~8), edgepf (both) -- standing and on-the-ground regimes, two interleaved rounds
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
L="product=gymnasium_amd/csrc/libmi355env.so pf=gymnasium_amd/csrc/libmi355env_pf.so edge=gymnasium_amd/csrc/libmi355env_edge.so edgepf=gymnasium_amd/csrc/libmi355env_edgepf.so"
python scripts/ab_bench.py --libs $L --envs Humanoid-v5:32768:4 --rounds 2 --env-kwargs '{"terminate_when_unhealthy": false}' --warmup 40 --out gpurun_out/r05e_ab_ground.txt
python scripts/ab_bench.py --libs $L --envs Humanoid-v5:32768:4 HumanoidStandup-v5:32768:4 --rounds 2 --out gpurun_out/r05e_ab_standing.txt
for V in edge edgepf; do
  MI355ENV_LIBRARY=$PWD/gymnasium_amd/csrc/libmi355env_$V.so timeout 600 python -m pytest tests/test_gpu_mujoco.py tests/test_mujoco_reference_pins.py -q -m gpu -k "umanoid or statistics" > gpurun_out/r05e_pytest_$V.log 2>&1; echo "$V pytest exit $?"; tail -2 gpurun_out/r05e_pytest_$V.log | cut -c1-160
done
