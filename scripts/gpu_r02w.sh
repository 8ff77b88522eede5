#!/bin/bash
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wrappers.py -m gpu -q > gpurun_out/r02w_pytest.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/r02w_pytest.log
python - <<'PY'
import time, numpy as np, gymnasium_amd
for eid in ("FrozenLake-v1", "Taxi-v4", "Blackjack-v1"):
    env = gymnasium_amd.make_vec(eid, num_envs=65536, copy=False)
    env.reset(seed=0); env.action_space.seed(0)
    acts = [env.action_space.sample() for _ in range(8)]
    for k in range(10): env.step(acts[k % 8])
    t0 = time.perf_counter()
    for k in range(200): env.step(acts[k % 8])
    print(eid, "us per NumPy step", (time.perf_counter() - t0) / 200 * 1e6)
PY
