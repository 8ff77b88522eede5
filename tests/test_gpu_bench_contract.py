"""-m gpu: bench.py's stdout is ONE strict-JSON line under 4 KB with the fields the driver's contract names (and the two objects this tier adds),
for the DRIVER'S OWN command (`--gpus 1 --steps 20 --warmup 5`, extras on) and for a short explicit run.  Guards the contract and the
known-answer digest, not the numbers."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_DIGEST = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_digest.json")))["CartPole-v1:65536:128:rank0"]


def _nan_guard(name):
    raise ValueError(f"non-finite constant {name} in the bench line")


def _run(extra, timeout):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra], capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"exactly one line on stdout, got {len(lines)}"
    assert len(lines[0]) < 4096, f"the line is {len(lines[0])} bytes"
    return json.loads(lines[0], parse_constant=_nan_guard)


def _check_contract(r, K, W):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline", "output_sha256", "verified", "rccl_ranks"):
        assert key in r, key
    assert r["n_gpus"] == 1 and r["steps"] == K and r["warmup"] == W and r["higher_is_better"] is True and r["scaling"] == "weak"
    assert r["vs_baseline"] is None and r["data"] == "synthetic" and r["dtype"] == "f64" and "workload" in r["config"] and "model" not in r["config"]
    assert r["value"] > 1e9 and abs(r["ms_per_step"] - 65536 * 128 / r["value"] * 1e3) / r["ms_per_step"] < 0.2  # env-steps/s over 65536 x 128 steps per launch
    rf = r["roofline"]
    for key in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_kernel_ms"):
        assert key in rf, key
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0 < rf["frac"] < 1
    assert rf["algorithmic_bytes_per_launch"] == (34 * 128 + 96) * 65536
    # the first timed launch is the known-answer rollout: the digest the REFERENCE produces for it (tests/golden/make_bench_digest.py), and the
    # in-run comparison with the oracle over every sub-environment
    assert r["output_sha256"] == GOLDEN_DIGEST
    v = r["verified"]
    assert v["ok"] is True and v["envs"] == 65536 and v["steps"] == 128 and v["compare"] == "array_equal" and v["policy_equals_host_sample"] is True


def test_short_explicit_run():
    r = _run(["--gpus", "1", "--steps", "5", "--warmup", "2", "--no-extras", "--pmc", "off", "--sustained", "0", "--cpu-budget", "2"], 600)
    _check_contract(r, 5, 2)
    assert r["clock_spinup"]["seconds"] == 0.5 and r["clock_spinup"]["launches"] > 0  # untimed, before the warm-up
    cb = r["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    # the oracle sharded over the host's usable CPUs (one single-threaded process each): `cores` = the processes actually used
    assert cb["kind"] == "port" and 1 <= cb["cores"] <= (os.cpu_count() or 1) and cb["value"] > 1e6 and cb["host_cpu_count"] == os.cpu_count()
    assert "secondary" not in r


def test_the_drivers_own_command_line():
    """`python bench.py --gpus 1 --steps 20 --warmup 5` with everything on: the sidecar file is written, the line carries the headline values only."""
    r = _run(["--gpus", "1", "--steps", "20", "--warmup", "5"], 900)
    _check_contract(r, 20, 5)
    assert r["roofline"]["traffic"] is None or 0.8 < r["roofline"]["traffic_over_algorithmic"] < 1.5
    sec = r["secondary"]
    assert "failed" not in sec and "skipped" not in sec, sec
    for key in ("Pendulum-v1@65536", "Acrobot-v1@65536", "MountainCarContinuous-v0@65536", "Ant-v5@32768", "Humanoid-v5@32768"):
        assert sec[key] > 1e5, (key, sec)
    full = json.load(open(os.path.join(ROOT, r["full"])))
    assert full["primary"]["output_sha256"] == GOLDEN_DIGEST and len(full["secondary"]) >= 6 and "api_step_device" in full
    # BASELINE.json configs[2]: the secondaries' known-answer launches against the digests the REFERENCE produced at the same shape
    ref2 = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_digest_configs2.json")))
    seen = 0
    for line in full["secondary"]:
        key = f"{line['env']}:{line['num_envs']}:{line['vector_steps_per_launch']}:rank0"
        if key in ref2 and "regime" not in line:
            assert line["output_sha256"] == ref2[key], line["env"]
            seen += 1
    assert seen >= 3  # configs[2]; plus the ToyText lines when the run was young enough to measure them
