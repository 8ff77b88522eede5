#!/bin/bash
set -u
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
echo "=== product"; timeout 300 python scripts/debug_acrobot.py 2>&1 | grep -v Warn | tail -12
echo "=== nosplit"; MI355ENV_LIBRARY=$PWD/gymnasium_amd/csrc/libmi355env_nosplit.so timeout 300 python scripts/debug_acrobot.py 2>&1 | grep -v Warn | tail -12
MI355ENV_LIBRARY=$PWD/gymnasium_amd/csrc/libmi355env_nosplit.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "acrobot" 2>&1 | tail -5
MI355ENV_LIBRARY=$PWD/gymnasium_amd/csrc/libmi355env_nosplit.so timeout 300 python bench.py --no-api --no-cpu-baseline --no-secondary --pmc off --env Acrobot-v1 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('nosplit Acrobot %.4g' % r['value'])"
