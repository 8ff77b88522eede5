#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for one bench.py configuration, then
# summarise into gpurun_out/<tag>.txt (copy the summary to profiles/ afterwards).
#   scripts/gpu_profile.sh <tag> [bench.py args...]
set -u
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
# PROF_STEPS=default: bench.py's own default timed region (~1 s), so that the kernel trace's average duration is the one of the bench line
if [ "${PROF_STEPS:-20}" = default ]; then STEPS=""; else STEPS="--steps ${PROF_STEPS:-20} --warmup ${PROF_WARMUP:-3}"; fi
# PROF_API=1: keep the per-launch step() API section of bench.py in the profiled command (step_kernel durations)
if [ "${PROF_API:-0}" = 1 ]; then NOAPI=""; else NOAPI="--no-api"; fi
CMD="python $ROOT/bench.py $STEPS $NOAPI --no-cpu-baseline --no-extras --no-verify $*"
rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_stats" -- $CMD > "$OUT/${TAG}_stats.log" 2>&1
PMC=()
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d "$OUT/${TAG}_$C" -- $CMD > "$OUT/${TAG}_$C.log" 2>&1 && PMC+=("$OUT/${TAG}_$C")
done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace \
  -d "$OUT/${TAG}_SQ" -- $CMD > "$OUT/${TAG}_SQ.log" 2>&1 && PMC+=("$OUT/${TAG}_SQ")
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace \
  -d "$OUT/${TAG}_SQ2" -- $CMD > "$OUT/${TAG}_SQ2.log" 2>&1 && PMC+=("$OUT/${TAG}_SQ2")
cd "$ROOT"
python scripts/rocpd_summary.py --stats "$OUT/${TAG}_stats" --pmc "${PMC[@]}" --cmd "$CMD" -o "$OUT/${TAG}.txt" > /dev/null
# keep the merge-back small: the raw databases are not needed once summarised
rm -rf "$OUT/${TAG}_stats" "$OUT/${TAG}_FETCH_SIZE" "$OUT/${TAG}_WRITE_SIZE" "$OUT/${TAG}_SQ" "$OUT/${TAG}_SQ2"
tail -n +1 "$OUT/${TAG}.txt" | grep -i "rollout_kernel\|rollout_duo\|mj_\|step_kernel" | head -40
