"""bench.py's N > 1 control flow, executed end to end before the driver's first multi-GPU run.

`python -m torch.distributed.run --nproc-per-node N tests/bench_dryrun.py --gpus N ...` -- launched as the driver launches bench.py; the wrapper
imports bench.main and injects a Config subclass on the CPU checker plus the gloo transport (there is no GPU in the build container; bench.py
itself has no flag that selects the checker).  What runs is everything that had never executed anywhere: the pilot's MAX all-reduce that fixes K on every rank, the
barrier + synchronise bracket, both `sustained` branches, the one statistics all-reduce, rank 0's CPU legs and the single JSON line with
`n_gpus: N`.  The numbers are NOT measurements (the line says engine = oracle)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(cmd_for_port, **kw):
    """Launch with a fresh rendezvous port; a port another (parallel) test grabbed between _free_port() and torchrun's bind is retried."""
    for _ in range(4):
        p = subprocess.run(cmd_for_port(_free_port()), capture_output=True, text=True, **kw)
        if p.returncode == 0 or not any(m in p.stderr for m in ("Address already in use", "EADDRINUSE", "address already in use")):
            break
    return p


def _run(world, extra, num_envs=256, inner=8):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", MI355ENV_CPU_WORKERS="2")
    cmd = lambda port: [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",  # noqa: E731
                        "--master-port", str(port), os.path.join(ROOT, "tests", "bench_dryrun.py"), "--gpus", str(world),
                        "--num-envs", str(num_envs), "--inner", str(inner), "--cpu-budget", "0.5", *extra]
    p = _torchrun(cmd, env=env, cwd=ROOT, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line from rank 0, got {len(lines)}: {p.stdout[-2000:]}"
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [2, 3])
def test_default_flags_pilot_sets_k_on_every_rank(world):
    """No --steps: K comes from the pilot, all-reduced MAX over the ranks; a timed region that is long enough IS the sustained figure."""
    r = _run(world, ["--pilot-seconds", "0.9", "--sustained", "0.5"])
    assert r["n_gpus"] == world and r["scaling"] == "weak" and r["higher_is_better"] is True and r["unit"] == "env-steps/s"
    # the rank count comes out of the collective, and every rank reported its own device
    assert r["rccl_ranks"] == world == r["world_size_env"] and r["distinct_devices"] == world and sorted(d["rank"] for d in r["devices"]) == list(range(world))
    assert r["steps"] >= 20 and r["warmup"] >= 5
    # every rank timed the same K launches of 8 vector steps over its own 256 sub-environments; autoreset steps are not counted
    lanes = world * 256 * 8 * r["steps"]
    assert 0.8 * lanes < r["value"] * r["ms_per_step"] * 1e-3 * r["steps"] <= lanes
    assert r["config"]["parallelism"] == f"env-sharded x{world} (no data-path collective)" and "oracle" in r["engine"]
    assert r["sustained_value"] == r["value"] or r["sustained"]["launches"] >= r["steps"]
    # every rank's first timed launch was compared with the oracle on its own global indices, and the verdicts travelled with the census
    assert r["verified"]["ok"] is True and r["verified_all_ranks"] is True and all(d["verified"] is True for d in r["devices"])
    assert len({d["output_sha256"] for d in r["devices"]}) == world and r["devices"][0]["output_sha256"] == r["output_sha256"][:16]
    assert len(json.dumps(r, separators=(",", ":"))) < 4096
    assert r["roofline"]["bound"] == "hbm" and r["roofline"]["algorithmic_bytes_per_launch"] == (34 * 8 + 96) * 256
    assert r["cpu_baseline"] is None  # (the CPU leg is timed at N = 1 only)
    assert "secondary" not in r and "api_step_device" not in r  # one-GPU extras stay out of the N > 1 line


def test_driver_style_explicit_steps_takes_the_separate_sustained_loop():
    """--steps K --warmup W as the driver passes them: EXACTLY K timed launches; a short timed region is followed by the separate sustained loop."""
    r = _run(2, ["--steps", "6", "--warmup", "2", "--sustained", "0.3"])
    assert r["steps"] == 6 and r["warmup"] == 2 and r["n_gpus"] == 2
    assert r["sustained"]["launches"] > 6 and r["sustained"]["seconds"] > 0  # (a launch count, not a wall-clock threshold: the oracle's speed varies with the host's load)
    assert r["episodes"] > 0 and 9.0 < r["mean_episode_return"] < 60.0  # CartPole under the random policy: ~22 steps per episode


def test_baseline_config4_command_line_humanoid_sharded():
    """BASELINE.json configs[4] -- Humanoid-v5 sharded over the ranks -- with the driver's flags (`--env Humanoid-v5 --num-envs N --inner 4`): the MuJoCo
    branch of the line (algorithmic bytes of a cooperative robot, `bound: "valu"`, the CPU leg's bounded sample) on two gloo ranks."""
    r = _run(2, ["--env", "Humanoid-v5", "--steps", "2", "--warmup", "1", "--sustained", "0", "--cpu-baseline-at-any-n"], num_envs=6, inner=2)
    assert r["n_gpus"] == 2 and r["rccl_ranks"] == 2 and r["config"]["env"] == "Humanoid-v5" and r["config"]["num_envs_per_gpu"] == 6
    lanes = 2 * 6 * 2 * r["steps"]
    assert 0 < r["value"] * r["ms_per_step"] * 1e-3 * r["steps"] <= lanes * (1 + 1e-9)  # (no episode ends in 4 steps: exactly `lanes` env-steps, up to rounding)
    assert r["roofline"]["bound"] == "valu" and r["roofline"]["kernel"] == "mj_physics_kernel" and r["roofline"]["algorithmic_bytes_per_launch"] > 0
    assert r["verified"]["ok"] is True and r["verified"]["compare"] == "atol 1e-8" and r["verified_all_ranks"] is True
    assert r["cpu_baseline"]["value"] > 0 and "Humanoid-v5" in r["cpu_baseline"]["sample"]


def test_baseline_config4_at_its_real_world_size():
    """BASELINE.json configs[4] as the driver will launch it -- `torchrun --nproc-per-node 8 bench.py --gpus 8 --env Humanoid-v5 --num-envs 32768 --inner 4` -- at
    world size 8, with a tiny shard per rank: eight ranks in the collective, eight distinct digests (every rank owns its own global
    env indices), every rank's first timed launch verified against the oracle, one line under 4 KB."""
    r = _run(8, ["--env", "Humanoid-v5", "--steps", "2", "--warmup", "1", "--sustained", "0"], num_envs=2, inner=4)
    assert r["n_gpus"] == 8 and r["rccl_ranks"] == 8 and r["world_size_env"] == 8 and r["distinct_devices"] == 8
    assert sorted(d["rank"] for d in r["devices"]) == list(range(8)) and len({d["output_sha256"] for d in r["devices"]}) == 8
    assert r["verified_all_ranks"] is True and all(d["verified"] is True for d in r["devices"])
    assert r["config"]["env"] == "Humanoid-v5" and r["config"]["vector_steps_per_launch"] == 4 and r["config"]["parallelism"] == "env-sharded x8 (no data-path collective)"
    assert 0 < r["value"] * r["ms_per_step"] * 1e-3 * r["steps"] <= 8 * 2 * 4 * r["steps"] * (1 + 1e-9)
    assert len(json.dumps(r, separators=(",", ":"))) < 4096


def test_bench_py_cannot_be_pointed_at_the_checker():
    """The product script has no --engine / --backend switch any more: the metric line can only come from the HIP engine."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "--engine" not in src and "--backend" not in src and "from oracle import oracle" in src  # (the cpu_baseline leg is the one allowed use)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--engine", "oracle"], capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "unrecognized arguments" in p.stderr
