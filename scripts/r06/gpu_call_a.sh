#!/bin/bash
# round 6, call A: what bounds the exact-libm Acrobot rollout and the ToyText rollouts (item 2): LDS bank conflicts,
# instruction-cache misses, resident wavefronts and the VALU mix, next to the same counters for Pendulum and CartPole's two-role kernel.
# Summaries: gpurun_out/r06_<env>_<group>.txt (one rocprofv3 --pmc pass per group), gpurun_out/r06_<toytext>_rollout.txt (trace + traffic + SQ).
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
export FILTER=rollout
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"
G2="SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_WAVES SQ_INSTS_VALU"
G3="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL"
G4="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INST_CYCLES_SALU"
G5="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_SMEM"
pmc() { # tag env n
  for g in 1 2 3 4 5; do
    eval "C=\$G$g"
    echo "== $1 group $g"
    timeout 300 bash scripts/gpu_pmc.sh r06_$1_g$g "$C" --env $2 --num-envs $3 --no-secondary --pmc off
  done
}
pmc acrobot Acrobot-v1 65536
pmc acrobot_x4 Acrobot-v1 262144
pmc pendulum Pendulum-v1 65536
pmc cartpole CartPole-v1 65536
for e in FrozenLake-v1 Taxi-v3 Blackjack-v1; do
  t=$(echo $e | tr 'A-Z' 'a-z' | sed 's/-v.//')
  PROF_STEPS=20 PROF_WARMUP=3 timeout 600 scripts/gpu_profile.sh r06_${t}_rollout --env $e > /dev/null 2>&1
  grep -c "tab_rollout" gpurun_out/r06_${t}_rollout.txt
  pmc $t $e 65536
done
# the same rollouts at 4x the sub-environments: does occupancy help the tabular kernels?
for e in FrozenLake-v1 Taxi-v3 Blackjack-v1 Acrobot-v1; do
  for n in 65536 262144; do
    echo "== occupancy $e $n"; timeout 300 python bench.py --env $e --num-envs $n --steps 20 --warmup 3 --no-secondary --pmc off --no-cpu-baseline --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
  done
done
