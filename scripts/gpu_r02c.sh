#!/bin/bash
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
scripts/gpu_tests_split.sh r02c
for S in PGS Newton; do
  timeout 300 python bench.py --no-api --no-cpu-baseline --no-secondary --pmc off --sustained 0 --env Humanoid-v5 --num-envs 32768 --inner 4 --steps 3 --warmup 1 --env-kwargs "{\"solver\": \"$S\"}" > gpurun_out/r02c_hum_$S.json 2> gpurun_out/r02c_hum_$S.err
  python -c "import json; r=json.load(open('gpurun_out/r02c_hum_$S.json')); print('Humanoid $S value %.4g ms/vector-step %.4g' % (r['value'], r['roofline']['avg_vector_step_ms']))"
done
timeout 300 python bench.py --no-api --no-cpu-baseline --no-secondary --pmc off --sustained 0 --env HumanoidStandup-v5 --num-envs 32768 --inner 4 --steps 3 --warmup 1 > gpurun_out/r02c_standup.json 2> gpurun_out/r02c_standup.err
python -c "import json; r=json.load(open('gpurun_out/r02c_standup.json')); print('HumanoidStandup PGS value %.4g ms/vector-step %.4g' % (r['value'], r['roofline']['avg_vector_step_ms']))"
timeout 600 python -m pytest tests -m gpu -q -x > gpurun_out/r02c_all.log 2>&1; echo "single-process suite exit $?"; tail -3 gpurun_out/r02c_all.log
