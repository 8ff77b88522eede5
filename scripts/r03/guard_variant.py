#!/usr/bin/env python3
"""tests/test_gpu_scheduler_guard.py's comparison for an A/B library: `python scripts/r03/guard_variant.py <libmi355env_variant.so>` runs the 16-lane robots
for 25 steps on the variant and on libmi355env_ref.so (hipcc's default scheduler, same sources) and requires every output bit to agree."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_scheduler_guard as G  # noqa: E402

lib = os.path.abspath(sys.argv[1])
bad = 0
for env_id in ("Ant-v5", "HalfCheetah-v5", "Hopper-v5", "Walker2d-v5"):
    kw = {} if env_id == "HalfCheetah-v5" else dict(terminate_when_unhealthy=False)
    with tempfile.TemporaryDirectory() as d:
        a = G.run_build(lib, env_id, kw, os.path.join(d, "a.npz"))
        b = G.run_build(G.REF, env_id, kw, os.path.join(d, "b.npz"))
        diff = [k for k in a.files if not np.array_equal(a[k], b[k], equal_nan=True)]
        nan = sum(int(np.isnan(a[k]).sum()) for k in a.files if a[k].dtype.kind == "f")
        print(env_id, "arrays that differ:", len(diff), "of", len(a.files), "NaNs:", nan, flush=True)
        bad += bool(diff) or nan > 0
sys.exit(1 if bad else 0)
