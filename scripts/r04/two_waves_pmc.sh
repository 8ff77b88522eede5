#!/bin/bash
# SQ counters of the cooperative Hopper physics kernel: product build (one wavefront per SIMD) vs the 256-register build (two per SIMD)
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd "$ROOT"
export MI355ENV_MJ_COOP=1 FILTER=mj_physics
for lib in product w2; do
  if [ $lib = product ]; then export MI355ENV_LIBRARY=$ROOT/gymnasium_amd/csrc/libmi355env.so; else export MI355ENV_LIBRARY=$ROOT/gymnasium_amd/csrc/libmi355env_w2.so; fi
  for env in "${@:-Hopper-v5}"; do
    echo "== $lib $env"
    bash scripts/gpu_pmc.sh r04_w2_${lib}_${env}_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY" --env $env --num-envs 65536 --inner 4 --no-secondary --pmc off
    bash scripts/gpu_pmc.sh r04_w2_${lib}_${env}_b "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU" --env $env --num-envs 65536 --inner 4 --no-secondary --pmc off
  done
done
