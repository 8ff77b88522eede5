#!/bin/bash
# round 6, call AP: the driver's 20-launch bench line, A/B on ONE box: CartPole's rare path inline against out of line (call AO's line on another box read 1.198e11)
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
L=$PWD/gymnasium_amd/csrc/libmi355env
: > gpurun_out/r06_cartpole_rare_out_of_line_bench_line_ab.txt
for r in 1 2 3; do for v in inline_rare.so:rare_inline so:rare_out_of_line; do lib=${v%%:*}; name=${v##*:}; [ "$lib" = so ] && p=${L}.so || p=${L}_${lib}
  MI355ENV_LIBRARY=$p timeout 120 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --pmc off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$name', 'round $r', '%.4g'%d['value'], 'env-steps/s', '%.2f us/launch'%(1e3*d['ms_per_step']))" | tee -a gpurun_out/r06_cartpole_rare_out_of_line_bench_line_ab.txt
done; done
