#!/bin/bash
# round 6, call AQ (the last GPU minutes): smoke() and the driver's 20-launch line on the from-scratch build of the committed tree (bit-identical to call AO's binaries)
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
sha256sum gymnasium_amd/csrc/libmi355env.so gymnasium_amd/csrc/libmi355env_ref.so > gpurun_out/r06_aq_final.txt
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v Warn | tail -3 | tee -a gpurun_out/r06_aq_final.txt
for r in 1 2 3; do timeout 60 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --pmc off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('line', $r, '%.4g'%d['value'], 'env-steps/s', '%.2f us/launch'%(1e3*d['ms_per_step']), 'frac', '%.3f'%d['roofline']['frac'], 'verified', d.get('verified',{}).get('ok') if isinstance(d.get('verified'),dict) else d.get('verified'))" | tee -a gpurun_out/r06_aq_final.txt
done
