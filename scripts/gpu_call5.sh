#!/bin/bash
cd "$(dirname "$0")/.."
for V in product nofmak product nofmak; do
  L=gymnasium_amd/csrc/libmi355env_$V.so; [ $V = product ] && L=gymnasium_amd/csrc/libmi355env.so
  echo "=== $V"; MI355ENV_LIBRARY=$PWD/$L timeout 300 python scripts/debug_acrobot3.py 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -12
done
