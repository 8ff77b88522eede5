#!/bin/bash
# round 6, call AO: CartPole's rare lanes (angle outside the short routine's range, a division operand outside the three-FMA range) redo their accelerations in an
# OUT-OF-LINE function: the hot path loses the spill of the saved exec mask and the general routines' scalar registers.  A/B first, then the whole GPU suite, smoke(),
# the rocprofv3 summary of the bench's kernel and the driver's bench command on this build
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
L=gymnasium_amd/csrc/libmi355env
timeout 300 python scripts/ab_bench.py --libs rare_inline=${L}_inline_rare.so rare_out_of_line=${L}.so --envs CartPole-v1:65536:128 CartPole-v1:262144:128 --rounds 3 --out gpurun_out/r06_cartpole_rare_out_of_line_ab.txt
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r06_final_pytest_gpu_ao.log 2>&1; tail -3 gpurun_out/r06_final_pytest_gpu_ao.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v Warn | tail -3
SECONDS=0; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_line_driver_cmd_ao.json 2> gpurun_out/r06_ao_bench.err; echo "bench.py exit $? after $SECONDS s"; cut -c1-300 gpurun_out/r06_bench_line_driver_cmd_ao.json
cp gpurun_out/bench_full.json gpurun_out/r06_bench_full_ao.json
PROF_STEPS=default timeout 300 scripts/gpu_profile.sh r06_cartpole_rollout_ao > /dev/null 2>&1; grep -c rollout_duo gpurun_out/r06_cartpole_rollout_ao.txt
