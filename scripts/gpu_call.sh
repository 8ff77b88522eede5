#!/bin/bash
# The ONE parameterised GPU-box script (round 4; replaces the 30 one-off scripts/r03/gpu_call*.sh logs).  Run through gpurun from the repo root:
#
#   gpurun --timeout 900 -- 'bash scripts/gpu_call.sh <tag> <task> [args...] [-- <task> [args...]] ...'
#
# tasks (outputs under gpurun_out/<tag>_*):
#   tests [pytest args]          python -m pytest tests -m gpu -q [args]
#   smoke                        __graft_entry__.smoke()  (CartPole vs oracle + the two-build miscompile guard)
#   bench [bench.py args]        one bench.py line -> <tag>_bench.json (+ a one-line summary)
#   profile <name> [bench args]  rocprofv3 --kernel-trace --stats + PMC passes of a bench.py configuration (scripts/gpu_profile.sh) -> <tag>_<name>.txt
#   pmc <name> "<counters>" [bench args]   one extra counter pass (scripts/gpu_pmc.sh)
#   ab <ab_bench.py args>        interleaved A/B of library builds (scripts/ab_bench.py; build variants with scripts/build_variant.py first)
#   py <script.py> [args]        any script of the repo
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
TAG=$1; shift
run_task() {
  local task=$1; shift
  case $task in
    tests) timeout 1500 python -m pytest tests -m gpu -q "$@" > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/${TAG}_pytest.log | cut -c1-220; grep -n "^FAILED\|^ERROR" gpurun_out/${TAG}_pytest.log | head -20 ;;
    smoke) python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "WARN\|logger.warn\|amdgpu.ids" | tail -5 ;;
    bench) local S=$(date +%s); timeout 900 python bench.py "$@" > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench exit $? in $(( $(date +%s) - S )) s"
           python - gpurun_out/${TAG}_bench.json <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1]))
    print("value %.4g %s frac %.3f kernel_ms %.4g n_gpus %s" % (r["value"], r["unit"], r["roofline"]["frac"] or 0, r["roofline"]["avg_kernel_ms"], r["n_gpus"]))
    for s in r.get("secondary", []):
        rf = s["roofline"]
        print("  %-26s %6d %-22s %.4g frac %s f64peak %s" % (s["env"], s["num_envs"], (s.get("regime") or "")[:22], s["value"], rf.get("frac"), rf.get("frac_of_f64_peak")))
except Exception as e:
    print("no bench line:", e)
PY
           ;;
    profile) local name=$1; shift; timeout 900 scripts/gpu_profile.sh ${TAG}_${name} "$@" ;;
    pmc) local name=$1; local ctrs=$2; shift 2; bash scripts/gpu_pmc.sh ${TAG}_${name} "$ctrs" "$@" ;;
    ab) python scripts/ab_bench.py "$@" --out gpurun_out/${TAG}_ab.txt ;;
    py) python "$@" ;;
    *) echo "unknown task $task"; return 2 ;;
  esac
}
args=()
for a in "$@"; do
  if [ "$a" = "--" ]; then run_task "${args[@]}"; args=(); else args+=("$a"); fi
done
[ ${#args[@]} -gt 0 ] && run_task "${args[@]}"
