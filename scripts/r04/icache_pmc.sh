#!/bin/bash
# Instruction-cache and LDS counters of the cooperative physics kernels (one wavefront per SIMD): is the issue stall an instruction-fetch stall?
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd "$ROOT"
export FILTER=mj_physics
run() { # lib env n
  if [ $1 = product ]; then export MI355ENV_LIBRARY=$ROOT/gymnasium_amd/csrc/libmi355env.so; else export MI355ENV_LIBRARY=$ROOT/gymnasium_amd/csrc/libmi355env_$1.so; fi
  echo "== $1 $2 $3"
  bash scripts/gpu_pmc.sh r04_ic_$1_$2_a "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_ICACHE_BUSY_CYCLES" --env $2 --num-envs $3 --inner 4 --no-secondary --pmc off
  bash scripts/gpu_pmc.sh r04_ic_$1_$2_b "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_BUSY_CYCLES" --env $2 --num-envs $3 --inner 4 --no-secondary --pmc off
}
MI355ENV_MJ_COOP=1 run product Hopper-v5 65536
MI355ENV_MJ_COOP=1 run w2 Hopper-v5 65536
run product Ant-v5 65536
run product Humanoid-v5 32768
