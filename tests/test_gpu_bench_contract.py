"""-m gpu: bench.py prints ONE JSON line with the fields the driver's contract names (and the two objects this tier adds), for a short explicit
run of the default workload.  Guards the contract, not the numbers."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-secondary", "--pmc", "off",
                          "--sustained", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"exactly one line on stdout, got {len(lines)}"
    r = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in r, key
    assert r["n_gpus"] == 1 and r["steps"] == 5 and r["warmup"] == 2 and r["higher_is_better"] is True and r["scaling"] == "weak"
    assert r["vs_baseline"] is None and r["data"] == "synthetic" and r["dtype"] == "f64" and "workload" in r["config"] and "model" not in r["config"]
    assert r["value"] > 1e9 and abs(r["ms_per_step"] - 65536 * 128 / r["value"] * 1e3) / r["ms_per_step"] < 0.2  # env-steps/s over 65536 x 128 steps per launch
    assert r["clock_spinup"]["seconds"] == 0.5 and r["clock_spinup"]["launches"] > 0  # untimed, before the warm-up
    rf = r["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0 < rf["frac"] < 1
    cb = r["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    # the oracle sharded over the host's logical CPUs (one single-threaded process each): `cores` = the processes actually used
    assert cb["kind"] == "port" and 1 <= cb["cores"] <= (os.cpu_count() or 1) and cb["value"] > 1e6 and cb["host_cpu_count"] == os.cpu_count()
