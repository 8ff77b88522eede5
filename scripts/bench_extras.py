#!/usr/bin/env python3
"""Everything bench.py's one line leaves out, measured on ONE GPU by a child process of `python bench.py --gpus 1` (or by hand):

    python scripts/bench_extras.py --out gpurun_out/bench_full.json [--primary primary.json] [--pmc auto|full|off] [--budget 150]

  secondary         BASELINE.json configs[2..4]: Pendulum / Acrobot / MountainCarContinuous @65536, Ant-v5 @32768 and @65536, Humanoid-v5 @32768
                    (per GPU), each with its own roofline (bench.Config.roofline) and, for the cooperative MuJoCo kernels, the SQ activity
                    share and the executed-fp64 rate; then the on-the-ground regime of the two headline robots
  primary_counters  issue-slot ceiling of the primary kernel (SQ_INSTS_VALU / SALU per env-step)
  api_step_*        the per-launch step() API: device tensors, one HIP graph, NumPy over PCIe, the three fused wrappers -- never `value`
  api_benchmark_vector_step   the metric by the reference's own protocol (utils/performance.py:57-103): NumPy API, host-side sampling in the loop
  shared_rng        rng="shared" (the reference's NumPy CartPoleVectorEnv semantics): step() and rollout() throughput of the compatibility mode
  opt_in            fast_math / Newton-solver configurations next to the default (reference-faithful) ones
  cpu_reference     Gymnasium's own AsyncVectorEnv / SyncVectorEnv / NumPy CartPoleVectorEnv where `import gymnasium` works (GYM_REFERENCE or an
                    installed package); elsewhere (the GPU box) the AsyncVectorEnv ARCHITECTURE restated (oracle/async_baseline.py, pinned on the
                    real one by tests/test_async_baseline.py) timed in this run, with the build container's real numbers as cpu_reference_recorded

The sidecar file is rewritten after every section (a run that is cut short leaves what it had); the LAST stdout line is the headline dict that
bench.py puts into its line as `secondary`.  No optional measurement STARTS once the run is older than --budget seconds."""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bench import HBM_PEAK_GBS, MJ_COOP, Config, _rocprof_counters, child_args, cpu_baseline, usable_cpus  # noqa: E402

STEP_BYTES = {"CartPole-v1": 108, "Pendulum-v1": 68, "Acrobot-v1": 116, "MountainCar-v0": 68, "MountainCarContinuous-v0": 64}
# (env, num_envs per GPU, vector steps per launch) of the secondary lines: BASELINE.json configs[2..4] + the north_star's Ant @65536
SECONDARY = [("Pendulum-v1", 65536, 128), ("Acrobot-v1", 65536, 128), ("MountainCarContinuous-v0", 65536, 128),
             ("Ant-v5", 32768, 4), ("Ant-v5", 65536, 4), ("Humanoid-v5", 32768, 4)]
# the ToyText kinds (bit-exact integer kernels), measured last and only while the run is young
TOYTEXT = [("FrozenLake-v1", 65536, 128), ("Taxi-v4", 65536, 128), ("Blackjack-v1", 65536, 128), ("MountainCar-v0", 65536, 128)]  # (+ the fifth classic id)
# The other contact regime of the two headline robots: the random policy with `terminate_when_unhealthy` ends a Humanoid episode after ~22 steps, so
# the batch above is mostly robots still upright; with termination off and a warm-up of GROUND_WARM launches every robot lies on the ground (many
# contacts, the PGS sweeps dominate).  A learner that keeps the robot alive lives between the two lines.
# what the counters say bounds the tabular rollouts (profiles/r06_{frozenlake,taxi,blackjack}_rollout*.txt): not memory
TOYTEXT_BOUND = {e: ("one wavefront per SIMD: ~380 (FrozenLake) / ~780 (Blackjack) instructions per env-step issued one per ~5 cycles (two 128-bit PCG64 steps -- the env's "
                     "transition draw and the policy's --, the categorical search, five stores) plus dependent LDS table lookups: half the wave cycles issue, the other "
                     "half wait; HBM traffic is the algorithmic bytes.  4x the sub-environments (4 wavefronts per SIMD) runs 2.6x (FrozenLake) / 2.0x (Blackjack) faster")
                 for e in ("FrozenLake-v1", "Taxi-v4", "Blackjack-v1")}
GROUND_WARM = 40
SECONDARY_GROUND = [("Ant-v5", 32768, 4), ("Humanoid-v5", 32768, 4)]
F64_PEAK_TFLOPS = 78.6  # MI355X vector fp64 (MI355X_MICROARCH.md): 256 CUs x 4 SIMDs x 16 lanes x 2 flops x 2.4 GHz
SIMDS, CLOCK_HZ = 1024, 2.4e9
SQ_COUNTERS = ["SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"]
ISSUE_COUNTERS = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU"]
FLOP_COUNTERS = ["SQ_INSTS_VALU", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_MFMA_MOPS_F64"]
# the reference itself, measured in the build container (no gymnasium on the GPU box)
CPU_REFERENCE_RECORDED = {
    "hardware": "8 vCPU Intel Xeon @ 2.10 GHz (build container), Python 3.10.12, NumPy 2.2.6, reference gymnasium v1.4.0",
    "how": "gymnasium.utils.performance.benchmark_vector_step, 2-3 s runs (BASELINE.md section 2)", "unit": "env-steps/s",
    "AsyncVectorEnv CartPole-v1": {"num_envs=4": 10.0e3, "num_envs=8": 14.6e3, "num_envs=16": 16.8e3, "cores": 8},
    "SyncVectorEnv CartPole-v1": {"num_envs=4": 51e3, "num_envs=64": 74e3, "num_envs=1024": 84e3, "cores": 1},
    "NumPy CartPoleVectorEnv (vector_entry_point)": {"num_envs=1024": 7.3e6, "num_envs=65536": 13.7e6, "cores": 1},
    "single env gym.make": {"CartPole-v1": 82e3, "MountainCar-v0": 70e3, "MountainCarContinuous-v0": 35e3, "Pendulum-v1": 20e3, "Acrobot-v1": 16e3, "cores": 1},
    "MuJoCo ids": "unavailable: `mujoco` is not installed in the build container either",
}


def dominant_kernel_of(env_id, inner):
    """bench.Config.dominant_kernel without constructing the env."""
    if env_id in bench.ROLLOUT_BYTES:
        chunk = bench.DUO_CHUNK.get(env_id)
        return "rollout_duo_kernel" if (chunk and inner % chunk == 0 and os.environ.get("MI355ENV_ROLLOUT_DUO", "1")[:1] != "0") else "rollout_kernel"
    if env_id in MJ_COOP:
        return "mj_physics_kernel"
    return bench.tabular_kernel(env_id) if env_id.split("-")[0] in ("FrozenLake", "FrozenLake8x8", "Taxi", "Blackjack", "CliffWalking") else "mj_rollout_kernel"


# ---- counters -------------------------------------------------------------------------------------------------------------------------
def issue_counters(c: Config, kernel_s, timeout_s=150):
    """The OTHER ceiling of a classic rollout kernel: at one wavefront per SIMD the instruction issue rate bounds it before HBM does."""
    ic = _rocprof_counters(child_args(c.env_id, c.N, c.inner, c.env_kwargs), ISSUE_COUNTERS, c.dominant_kernel(), timeout_s)
    if not ic or ic.get("SQ_WAVES", (0, 0))[0] <= 0:
        return None
    waves, groups = ic["SQ_WAVES"][0], c.N / 64.0  # a group = 64 sub-environments: ONE wavefront in the one-role kernels, an env + an aux wavefront in rollout_duo_kernel
    valu, salu = ic["SQ_INSTS_VALU"][0] / groups / c.inner, ic["SQ_INSTS_SALU"][0] / groups / c.inner
    per_group = waves / groups
    slots = valu + salu if per_group < 1.5 else valu  # with two wavefronts per SIMD scalar instructions issue beside the partner's vector instructions
    ceiling = SIMDS * 64 * CLOCK_HZ / (4.0 * slots)
    lane_steps = c.N * c.inner / kernel_s
    return {"valu_per_env_step": valu, "salu_per_env_step": salu, "wavefronts_per_64_envs": per_group, "ceiling_env_steps_per_s": ceiling,
            "achieved_lane_steps_per_s": lane_steps, "frac_of_issue_ceiling": lane_steps / ceiling,
            "hbm_ceiling_env_steps_per_s": HBM_PEAK_GBS * 1e9 / (c.algorithmic_bytes_per_launch() / (c.N * c.inner)),
            "assumptions": f"{SIMDS} SIMDs x 64 lanes, one instruction per 4 cycles at {CLOCK_HZ / 1e9:.1f} GHz",
            "source": "rocprofv3 --pmc " + " ".join(ISSUE_COUNTERS) + " on a child invocation in this run"}


def coop_counters(c: Config, kernel_s, env_steps_per_s, warm=1, timeout_s=150, measured=None):
    """Cooperative MuJoCo physics kernel: share of wave cycles that issue VALU work, and the fp64 flops it EXECUTES against the vector fp64 peak
    (2 per FMA, 1 per MUL / ADD, x 64 lanes per wave-level instruction -- masked lanes are counted: the rate the vector units are driven at).
    `measured`: {"SQ": ..., "FLOP": ...} from bench.live_counters_batch (one child process for several configurations); else two passes of its own."""
    out, args, kernel = {}, child_args(c.env_id, c.N, c.inner, c.env_kwargs, warm), c.dominant_kernel()
    sq = (measured or {}).get("SQ") or (None if measured is not None else _rocprof_counters(args, SQ_COUNTERS, kernel, timeout_s))
    if sq and sq.get("SQ_WAVE_CYCLES", (0, 0))[0] > 0:
        wc = sq["SQ_WAVE_CYCLES"][0]
        out["sq"] = {k[3:].lower() + "_frac": sq[k][0] / wc for k in SQ_COUNTERS[1:] if k in sq}
    fc = (measured or {}).get("FLOP") or (None if measured is not None else _rocprof_counters(args, FLOP_COUNTERS, kernel, timeout_s))
    if fc and fc.get("SQ_INSTS_VALU", (0, 0))[0] > 0:
        g = lambda k: fc.get(k, (0.0, 0))[0]  # noqa: E731
        per_dispatch = 64.0 * (2.0 * g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_ADD_F64"))
        stepping = (env_steps_per_s * kernel_s / c.inner) if env_steps_per_s else float(c.N)
        rate = per_dispatch / (kernel_s / c.inner)  # dispatches of the physics kernel run back to back: one per vector step
        out["flops"] = {"fma_f64": g("SQ_INSTS_VALU_FMA_F64"), "mul_f64": g("SQ_INSTS_VALU_MUL_F64"), "add_f64": g("SQ_INSTS_VALU_ADD_F64"),
                        "trans_f64": g("SQ_INSTS_VALU_TRANS_F64"), "mfma_mops_f64": g("SQ_INSTS_VALU_MFMA_MOPS_F64"), "valu": g("SQ_INSTS_VALU"),
                        "flops_per_env_step": per_dispatch / max(stepping, 1.0)}
        out.update({"achieved_tflops_f64": rate / 1e12, "peak_tflops_f64": F64_PEAK_TFLOPS, "frac_of_f64_peak": rate / 1e12 / F64_PEAK_TFLOPS})
    return out


# ---- the reference's own vectorisers ---------------------------------------------------------------------------------------------------------
def cpu_reference(budget_s=4.0):
    """Gymnasium's own vectorisers on this host's cores, if the package is importable (utils/performance.py:57-103); else the port."""
    ref = os.environ.get("GYM_REFERENCE", "/root/reference")
    if os.path.isdir(os.path.join(ref, "gymnasium")) and ref not in sys.path:
        sys.path.append(ref)
    try:
        import gymnasium as gym
        from gymnasium.utils.performance import benchmark_vector_step
    except Exception:
        return cpu_reference_port(budget_s)
    cores = os.cpu_count() or 1
    out = {"cores": cores, "unit": "env-steps/s", "gymnasium": gym.__version__, "how": f"benchmark_vector_step, target_duration={budget_s} s"}
    for label, kw in ((f"AsyncVectorEnv CartPole-v1 num_envs={cores}", dict(num_envs=cores, vectorization_mode="async")),
                      ("SyncVectorEnv CartPole-v1 num_envs=1024", dict(num_envs=1024, vectorization_mode="sync")),
                      ("NumPy CartPoleVectorEnv num_envs=65536", dict(num_envs=65536, vectorization_mode="vector_entry_point"))):
        try:
            env = gym.make_vec("CartPole-v1", **kw)
            out[label] = benchmark_vector_step(env, target_duration=budget_s, seed=0)
            env.close()
        except Exception as e:  # a missing optional dependency must not cost the GPU numbers
            out[label] = f"failed: {type(e).__name__}: {e}"
    return out


def cpu_reference_port(budget_s=4.0):
    """Where gymnasium itself is not importable (the GPU box): the reference's AsyncVectorEnv ARCHITECTURE restated (oracle/async_baseline.py:
    one process per sub-environment, pipes, shared-memory observations, the scalar CartPole in Python) and timed by the same counting rule as
    benchmark_vector_step -- in THIS run, on THIS host's cores.  It carries less per-step overhead than the real thing: an upper bound of it."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "async_baseline.py")):
        return None
    usable, why = usable_cpus()
    out = {"kind": "port", "what": "oracle/async_baseline.py: AsyncVectorEnv's architecture (vector/async_vector_env.py) around a Python CartPole-v1, "
                                   "NOT gymnasium itself (not installed on this host)", "unit": "env-steps/s", "usable_cpus": usable,
           "usable_cpus_source": why, "host_cpu_count": os.cpu_count(), "how": f"benchmark_vector_step's loop and counting rule, target_duration={budget_s} s"}
    for n in sorted({usable, 4 * usable}):  # the reference's own convention (num_envs = cores) and an over-subscribed one
        key = f"AsyncVectorEnv-port CartPole-v1 num_envs={n}"
        try:  # in a fresh interpreter: its worker processes are forked from a process without a HIP context
            p = subprocess.run([sys.executable, "-m", "oracle.async_baseline", str(n), str(budget_s)], cwd=ROOT, capture_output=True, text=True, timeout=budget_s * 3 + 120)
            out[key] = float(p.stdout.strip().splitlines()[-1])
        except Exception as e:
            out[key] = f"failed: {type(e).__name__}: {e}"
    return out


# ---- the per-launch step() API ---------------------------------------------------------------------------------------------------------------
def api_legs(env_id, N, local_rank=0):
    import torch

    import gymnasium_amd
    from gymnasium_amd.gym_api import error

    out = {}
    dev = torch.device("cuda", local_rank)
    env = gymnasium_amd.make_vec(env_id, num_envs=N, device=local_rank, output="torch", copy=False)
    env.reset(seed=0)
    eng = env._engine
    a_dev = torch.randint(0, 2, (N,), device=dev) if env._discrete else (torch.rand((N, eng.act_dim), device=dev) * 0.8 - 0.4)
    for _ in range(20):
        env.step(a_dev)
    torch.cuda.synchronize()
    reps = 300
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        env.step(a_dev)
    e1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    step_kernel_s = e0.elapsed_time(e1) * 1e-3 / reps
    out["api_step_device"] = {"value": N * reps / dt, "unit": "vector-env lanes/s (incl. autoreset lanes)", "us_per_step_wall": dt / reps * 1e6,
                              "us_per_step_gpu": step_kernel_s * 1e6,
                              "roofline_frac": (STEP_BYTES[env_id] * N / step_kernel_s / 1e9 / HBM_PEAK_GBS) if env_id in STEP_BYTES else None}
    # the same steps as ONE HIP graph (HipVectorEnv.capture_steps): what is left when the host is out of the loop
    G_STEPS, g_reps = 32, 30
    try:
        graphed = env.capture_steps(actions=a_dev, steps=G_STEPS)
        for _ in range(3):
            graphed.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(g_reps):
            graphed.replay()
        e1.record()
        torch.cuda.synchronize()
        dtg = time.perf_counter() - t0
        out["api_step_graph"] = {"value": N * G_STEPS * g_reps / dtg, "unit": "vector-env lanes/s (incl. autoreset lanes)", "steps_per_graph": G_STEPS,
                                 "us_per_step_wall": dtg / (G_STEPS * g_reps) * 1e6, "us_per_step_gpu": e0.elapsed_time(e1) * 1e3 / (G_STEPS * g_reps),
                                 "roofline_frac": (STEP_BYTES[env_id] * N / (dtg / (G_STEPS * g_reps)) / 1e9 / HBM_PEAK_GBS) if env_id in STEP_BYTES else None}
        del graphed
    except error.Error as e:  # envs that assemble infos on the host (ToyText) or carry fused wrappers cannot be captured: say so, do not abort
        out["api_step_graph"] = {"skipped": str(e)[:200]}
    env.close()
    env_np = gymnasium_amd.make_vec(env_id, num_envs=N, device=local_rank, copy=False)
    env_np.reset(seed=0)
    env_np.action_space.seed(0)
    actions = [env_np.action_space.sample() for _ in range(8)]
    t0 = time.perf_counter()
    for _ in range(40):
        env_np.action_space.sample()
    sample_us = (time.perf_counter() - t0) / 40 * 1e6
    reps = 100

    def timed_steps(use_pinned):
        for k in range(5):
            env_np.step(actions[k % 8])
        t0 = time.perf_counter()
        for k in range(reps):
            env_np.step(env_np.action_buffer if use_pinned else actions[k % 8])  # pinned: the caller's policy writes straight into the upload array
        return (time.perf_counter() - t0) / reps

    dt_page = timed_steps(False)
    env_np.action_buffer[...] = actions[0]
    dt_pin = timed_steps(True)
    out["api_step_numpy"] = {"value": N / dt_page, "unit": "vector-env lanes/s (NumPy in / NumPy out over PCIe, host action sampling excluded)",
                             "us_per_step_wall": dt_page * 1e6, "us_per_step_wall_actions_in_pinned_buffer": dt_pin * 1e6, "host_action_space_sample_us": sample_us}
    env_np.close()
    if env_id in STEP_BYTES:  # the reference's stateful wrappers on top, device tensors in and out
        from gymnasium_amd import wrappers as gw

        wrapped = {}
        for mode in ("fused", "standalone"):
            env_w = gymnasium_amd.make_vec(env_id, num_envs=N, device=local_rank, output="torch", copy=False)
            if mode == "standalone":
                env_w.FUSES_WRAPPERS = False
            w = gw.ClipReward(gw.NormalizeReward(gw.NormalizeObservation(env_w)), -5.0, 5.0)
            w.reset(seed=0)
            for _ in range(10):
                w.step(a_dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                w.step(a_dev)
            torch.cuda.synchronize()
            wrapped[mode] = (time.perf_counter() - t0) / 100 * 1e6
            w.close()
        out["api_step_wrapped"] = {"wrappers": "ClipReward(NormalizeReward(NormalizeObservation(env)))", "us_per_step_wall_fused": wrapped["fused"],
                                   "us_per_step_wall_standalone_passes": wrapped["standalone"], "launches_per_step_fused": 2 if N <= 262144 else 3,
                                   "launches_per_step_standalone": 11}
    return out


def api_faithful_leg(env_id, N, seconds=2.0, local_rank=0):
    """the metric as the reference's own `benchmark_vector_step` measures it (utils/performance.py:57-103) -- NumPy batches through the
    public API, the policy `action_space.sample()` drawn on the HOST inside the timed loop, NEXT_STEP reset steps not counted, wall clock.  This is what a
    user's unchanged Gymnasium script gets; bench.py's `value` is the fused, device-resident rollout.  Since round 6 `action_space.sample()` of a
    HipVectorEnv is served by the engine (one device launch + one pinned copy per block of batches, vector/device_policy.py) instead of NumPy on the host."""
    import gymnasium_amd

    env = gymnasium_amd.make_vec(env_id, num_envs=N, device=local_rank, copy=False)
    env.action_space.seed(0)
    env.reset(seed=0)
    env.step(env.action_space.sample())
    env.reset(seed=0)
    counted, prev_done, t0 = 0, np.zeros(N, dtype=np.bool_), time.time()
    while True:
        _, _, te, tr, _ = env.step(env.action_space.sample())
        counted += N - int(np.count_nonzero(prev_done))
        prev_done = np.logical_or(te, tr)
        t1 = time.time()
        if t1 - t0 > seconds:
            break
    env.close()
    return {"value": counted / (t1 - t0), "unit": "env-steps/s", "how": f"benchmark_vector_step's protocol, target_duration={seconds} s, NumPy in / out, action_space.sample() NumPy batches drawn by the engine"}


def api_faithful_device_leg(env_id, N, seconds=2.0, local_rank=0):
    """The same protocol with the policy on the device (round 6): `output="torch", sample_output="torch"` -- `action_space.sample()` hands out device
    tensors the engine drew ahead from the space's own stream (mi_action_sample; bit-equal to the NumPy sampler, tests/test_gpu_device_policy.py), `step()`
    returns device tensors, and the loop `env.step(env.action_space.sample())` of utils/performance.py:82-97 enqueues nothing but step kernels.  The count is
    the engine's own `env_steps` (NEXT_STEP reset steps not counted, like the reference's `num_envs - count_nonzero(previous_done)`), read once after the
    loop instead of from the flags on the host after every step -- the one change to the protocol, and what keeps the host out of the loop.  Also: the same
    loop as `env.step(None)` (the step kernel draws the batch itself: one launch per step, mi_step with actions == NULL) and as one HIP graph of 32 such steps."""
    import torch

    import gymnasium_amd

    out = {}

    def protocol(step_fn, label, per_call=1):
        env = gymnasium_amd.make_vec(env_id, num_envs=N, device=local_rank, output="torch", sample_output="torch", copy=False)
        env.action_space.seed(0)
        env.reset(seed=0)
        env.step(env.action_space.sample())
        env.reset(seed=0)
        fn = step_fn(env)
        for _ in range(3):
            fn()
        env.synchronize()
        env.reset_statistics()
        calls, t0 = 0, time.time()
        while True:
            fn()
            calls += 1
            if (calls & 63) == 0 and time.time() - t0 > seconds:  # (the clock is read every 64 calls: time.time() costs what a step costs)
                break
        env.synchronize()
        t1 = time.time()
        st = env.statistics()
        out[label] = {"value": st["env_steps"] / (t1 - t0), "unit": "env-steps/s", "us_per_step_wall": (t1 - t0) / (calls * per_call) * 1e6,
                      "vector_steps": calls * per_call, "counted_env_steps": st["env_steps"], "reset_steps_not_counted": st["reset_steps"]}
        env.close()

    protocol(lambda env: (lambda: env.step(env.action_space.sample())), "step_of_a_device_sample")
    protocol(lambda env: (lambda: env.step(None)), "step_none")
    try:
        protocol(lambda env: env.capture_steps(policy="random", steps=32).replay, "graph_of_32_sampled_steps", per_call=32)
    except Exception as e:
        out["graph_of_32_sampled_steps"] = {"skipped": f"{type(e).__name__}: {e}"[:200]}
    out["how"] = ("benchmark_vector_step's loop (utils/performance.py:82-97) with output='torch', sample_output='torch'; env-steps counted by the engine's own "
                  f"counter after the loop; target_duration={seconds} s")
    out["value"] = out["step_of_a_device_sample"]["value"]
    return out


def shared_rng_leg(N=65536, local_rank=0):
    """rng="shared" (the reference's NumPy CartPoleVectorEnv semantics, MI_CFG_SHARED_RNG): a compatibility mode, two launches per step -- what it costs next to
    the per-sub-environment streams, and next to the class it replaces (13.7 M env-steps/s for the NumPy class at 65 536 on one core)."""
    import torch

    import gymnasium_amd

    env = gymnasium_amd.make_vec("CartPole-v1", num_envs=N, rng="shared", device=local_rank, output="torch", copy=False)
    env.reset(seed=0)
    env.action_space.seed(0)
    a = torch.randint(0, 2, (N,), device=torch.device("cuda", local_rank))
    for _ in range(20):
        env.step(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    env.reset_statistics()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(300):
        env.step(a)
    e1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = env.statistics()["env_steps"]
    out = {"step_value": steps / dt, "unit": "env-steps/s", "us_per_step_wall": dt / 300 * 1e6, "us_per_step_gpu": e0.elapsed_time(e1) * 1e3 / 300, "launches_per_step": 2}
    env.rollout(128)
    torch.cuda.synchronize()
    env.reset_statistics()
    t0 = time.perf_counter()
    for _ in range(4):
        env.rollout(128)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out.update({"rollout_value": env.statistics()["env_steps"] / dt, "rollout": "4 x rollout(128): 128 x [policy sample, scan, step] launches each"})
    env.close()
    return out


def steady(c, seconds=0.7):
    """~`seconds` of back-to-back launches after a tenth of that as warm-up: (env-steps/s, launches, avg kernel s, elapsed)"""
    import torch

    for _ in range(2):
        c.launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        c.launch()
    torch.cuda.synchronize()
    k = int(min(20000, max(3, round(seconds / max((time.perf_counter() - t0) / 3, 1e-6)))))
    for _ in range(max(1, k // 10)):
        c.launch()
    el, ks, st = c.timed(k, torch.cuda.synchronize)
    return st["env_steps"] / el, k, ks, el


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--primary", default=None, help="JSON file with bench.py's primary result (copied into the sidecar)")
    ap.add_argument("--pmc", choices=["auto", "full", "off"], default="auto")
    ap.add_argument("--budget", type=float, default=210.0)
    ap.add_argument("--started", type=float, default=None, help="time.time() at which the parent command started (budget reference)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-api", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle comparison of the MuJoCo lines (bench.mujoco_window_check)")
    ap.add_argument("--only", default=None, help="comma-separated env ids: restrict the secondary lines")
    ap.add_argument("--api-only", action="store_true", help="only the per-launch step() API legs (for a kernel trace of step_kernel)")
    ap.add_argument("--policy-only", action="store_true", help="only the benchmark_vector_step protocol legs (NumPy and device policy)")
    args = ap.parse_args()
    t_ref = args.started if args.started else time.time()
    young = lambda: args.pmc == "full" or (time.time() - t_ref) < args.budget  # noqa: E731
    import torch

    torch.cuda.set_device(0)
    full = {"primary": json.load(open(args.primary)) if args.primary and os.path.exists(args.primary) else None, "secondary": []}
    head = {}

    def flush():
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out + ".tmp", "w") as f:
            json.dump(full, f, indent=1, default=float)
        os.replace(args.out + ".tmp", args.out)

    only = set(args.only.split(",")) if args.only else None
    # Live HBM traffic for EVERY line of this file (no recorded round-3 numbers): the FETCH_SIZE / WRITE_SIZE passes of all
    # configurations run up front in a handful of child processes (bench.live_traffic_batch), a line then finds its bytes under its key.
    traffic, coop_measured = {}, {}
    if args.pmc != "off" and not args.policy_only and not args.api_only:
        reqs = []
        for env_id, n2, inner2 in SECONDARY + TOYTEXT:
            if only and env_id not in only:
                continue
            reqs.append({"env_id": env_id, "N": n2, "inner": inner2, "env_kwargs": None, "warm": 1, "kernel": dominant_kernel_of(env_id, inner2), "key": f"{env_id}:{n2}",
                         "extra": env_id in MJ_COOP})
        for env_id, n2, inner2 in SECONDARY_GROUND:
            if only and env_id not in only:
                continue
            reqs.append({"env_id": env_id, "N": n2, "inner": inner2, "env_kwargs": {"terminate_when_unhealthy": False}, "warm": GROUND_WARM,
                         "kernel": dominant_kernel_of(env_id, inner2), "key": f"{env_id}:{n2}:ground", "extra": True})
        t_tr = time.time()
        traffic, coop_measured = bench.live_counters_batch(reqs, {"SQ": SQ_COUNTERS, "FLOP": FLOP_COUNTERS})
        full["traffic_passes"] = {"seconds": round(time.time() - t_tr, 1), "configurations": len(reqs), "measured": sum(1 for v in traffic.values() if v[0] is not None)}
    if args.policy_only:
        full["api_benchmark_vector_step"] = api_faithful_leg("CartPole-v1", 65536)
        full["api_benchmark_vector_step_device_policy"] = api_faithful_device_leg("CartPole-v1", 65536)
        flush()
        print(json.dumps({"numpy": full["api_benchmark_vector_step"]["value"], **{k: v.get("value", v) for k, v in full["api_benchmark_vector_step_device_policy"].items() if isinstance(v, dict)}}))
        return
    if args.api_only:
        full.update(api_legs("CartPole-v1", 65536))
        flush()
        print(json.dumps({"step_api_us_gpu": round(full["api_step_device"]["us_per_step_gpu"], 2)}))
        return
    for env_id, n2, inner2 in SECONDARY:
        if only and env_id not in only:
            continue
        c2 = Config(env_id, n2, inner2, 0, 0)
        # the known-answer launch of this configuration (as bench.py's first timed launch: reset(seed=0), policy stream seeded 0): its digest is what
        # tests/golden/bench_digest_configs2.json holds FROM THE REFERENCE for the classic kinds (tests/test_gpu_bench_contract.py compares)
        c2.launch(c2.first)
        sha = bench.trajectory_digest(c2.host_trajectory())
        v2, k2, ks2, el2 = steady(c2)
        line = {"env": env_id, "num_envs": n2, "vector_steps_per_launch": inner2, "launches": k2, "value": v2, "seconds": el2, "unit": "env-steps/s",
                "ms_per_launch": el2 / k2 * 1e3, "dtype": "f64", "output_sha256": sha, "roofline": c2.roofline(ks2, traffic=traffic.get(f"{env_id}:{n2}"))}
        head[f"{env_id}@{n2}"] = float(f"{v2:.4g}")
        if env_id not in MJ_COOP:
            head.setdefault("hbm_frac", {})[env_id] = round(line["roofline"]["frac"], 4)
        full["secondary"].append(line)
        flush()
        if env_id in MJ_COOP and args.pmc != "off":
            line["roofline"].update(coop_counters(c2, ks2, v2, measured=coop_measured.get(f"{env_id}:{n2}", {})))
        if env_id in MJ_COOP and not args.no_verify:  # >= 1 024 robots x 40 steps against the oracle, from where the timed launches left them
            try:
                line["verified"] = bench.mujoco_window_check(c2)
            except Exception as e:
                line["verified"] = {"ok": None, "error": f"{type(e).__name__}: {e}"[:200]}
            head.setdefault("verified", {})[f"{env_id}@{n2}"] = line["verified"].get("ok")
        c2.close()
        opt = {"fast_math": True} if env_id in STEP_BYTES else ({"solver": "Newton"} if env_id in ("Humanoid-v5", "HumanoidStandup-v5") else None)
        if opt and young():  # the opt-in, faster configuration next to the default (reference-faithful) one
            c3 = Config(env_id, n2, inner2, 0, 0, opt)
            line["opt_in"] = {"env_kwargs": opt, "value": steady(c3)[0], "unit": "env-steps/s"}
            c3.close()
            # (north_star asks for tolerance parity on the continuous dynamics; the default above is bit-exact; this is the tolerance-parity configuration)
            head.setdefault("opt_in", {})[f"{env_id} {next(iter(opt))}={next(iter(opt.values()))}"] = float(f"{line['opt_in']['value']:.4g}")
        if not args.no_cpu_baseline and young():  # MuJoCo: a bounded sample (the oracle's per-env cost does not depend on the batch size)
            line["cpu_baseline"] = cpu_baseline(env_id, min(n2, 64 * usable_cpus()[0]) if env_id in MJ_COOP else n2, budget_s=3.0)
        flush()
    # the on-the-ground regime of the two headline robots: termination off, GROUND_WARM launches before anything is timed or profiled
    for env_id, n2, inner2 in SECONDARY_GROUND:
        if only and env_id not in only:
            continue
        kw = {"terminate_when_unhealthy": False}
        c2 = Config(env_id, n2, inner2, 0, 0, kw)
        for _ in range(GROUND_WARM):
            c2.launch()
        v2, k2, ks2, el2 = steady(c2)
        line = {"env": env_id, "num_envs": n2, "vector_steps_per_launch": inner2, "launches": k2, "value": v2, "seconds": el2, "unit": "env-steps/s",
                "ms_per_launch": el2 / k2 * 1e3, "dtype": "f64", "env_kwargs": kw,
                "regime": f"robots on the ground: terminate_when_unhealthy=False, {GROUND_WARM} launches ({GROUND_WARM * inner2} vector steps) of warm-up before the timed region",
                "roofline": c2.roofline(ks2, traffic=traffic.get(f"{env_id}:{n2}:ground"))}
        head[f"{env_id}@{n2} on the ground"] = float(f"{v2:.4g}")
        if args.pmc != "off":
            line["roofline"].update(coop_counters(c2, ks2, v2, warm=GROUND_WARM, measured=coop_measured.get(f"{env_id}:{n2}:ground", {})))
        if not args.no_verify:  # the regime that matters: robots on the ground, many contacts, PGS at its sweep cap
            try:
                line["verified"] = bench.mujoco_window_check(c2)
            except Exception as e:
                line["verified"] = {"ok": None, "error": f"{type(e).__name__}: {e}"[:200]}
            head.setdefault("verified", {})[f"{env_id}@{n2} on the ground"] = line["verified"].get("ok")
        c2.close()
        full["secondary"].append(line)
        flush()
    prim = (full["primary"] or {}).get("config", {})
    env_id, N, inner = prim.get("env", "CartPole-v1"), prim.get("num_envs_per_gpu", 65536), prim.get("vector_steps_per_launch", 128)
    if not args.no_api and not only:
        try:
            full.update(api_legs(env_id, N))
            head["step_api_us_gpu"] = round(full["api_step_device"]["us_per_step_gpu"], 2)
            if "us_per_step_gpu" in full.get("api_step_graph", {}):
                head["step_graph_us_gpu"] = round(full["api_step_graph"]["us_per_step_gpu"], 2)
        except Exception as e:
            full["api_error"] = f"{type(e).__name__}: {e}"
        flush()
    if not args.no_api and not only:  # (not subject to the budget: SURVEY 8(d)(i) asks for this number first)
        try:
            full["api_benchmark_vector_step"] = api_faithful_leg(env_id, N)
            head["api_faithful_numpy"] = float(f"{full['api_benchmark_vector_step']['value']:.4g}")
        except Exception as e:
            full["api_benchmark_vector_step"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        try:
            full["api_benchmark_vector_step_device_policy"] = api_faithful_device_leg(env_id, N)
            head["api_faithful"] = float(f"{full['api_benchmark_vector_step_device_policy']['value']:.4g}")
            head["api_step_none"] = float(f"{full['api_benchmark_vector_step_device_policy']['step_none']['value']:.4g}")
        except Exception as e:
            full["api_benchmark_vector_step_device_policy"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        flush()
    if env_id == "CartPole-v1" and not args.no_api and not only and young():
        try:
            full["shared_rng"] = shared_rng_leg(N)
            head["shared_rng_step"] = float(f"{full['shared_rng']['step_value']:.4g}")
        except Exception as e:
            full["shared_rng"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        flush()
    if env_id in STEP_BYTES and not prim.get("env_kwargs") and not only and young():
        c_opt = Config(env_id, N, inner, 0, 0, {"fast_math": True})
        v_o, _, ks_o, _ = steady(c_opt)
        full["opt_in"] = {"env_kwargs": {"fast_math": True}, "value": v_o, "unit": "env-steps/s",
                          "note": "device sin / cos and x * x instead of the bit-exact libm restatements (tolerance parity, tests/test_gpu_parity.py)"}
        c_opt.close()
        flush()
    if args.pmc != "off" and env_id in STEP_BYTES and not only and young():
        c1 = Config(env_id, N, inner, 0, 0, prim.get("env_kwargs") or None)
        _, _, ks1, _ = steady(c1, 0.3)
        full["primary_counters"] = {"issue_bound": issue_counters(c1, ks1)}
        c1.close()
        if full["primary_counters"]["issue_bound"]:
            head["issue_frac"] = round(full["primary_counters"]["issue_bound"]["frac_of_issue_ceiling"], 3)
        flush()
    for env_id2, n2, inner2 in TOYTEXT:
        if only or not young():
            break
        try:
            c2 = Config(env_id2, n2, inner2, 0, 0)
            c2.launch(c2.first)
            sha = bench.trajectory_digest(c2.host_trajectory())
            v2, k2, ks2, el2 = steady(c2, 0.4)
            full["secondary"].append({"env": env_id2, "num_envs": n2, "vector_steps_per_launch": inner2, "launches": k2, "value": v2, "seconds": el2, "unit": "env-steps/s",
                                      "ms_per_launch": el2 / k2 * 1e3, "dtype": "i64", "output_sha256": sha, "roofline": c2.roofline(ks2, traffic=traffic.get(f"{env_id2}:{n2}")),
                                      "bound_note": TOYTEXT_BOUND.get(env_id2)})
            head[f"{env_id2}@{n2}"] = float(f"{v2:.4g}")
            c2.close()
        except Exception as e:
            full["secondary"].append({"env": env_id2, "error": f"{type(e).__name__}: {e}"[:300]})
        flush()
    if not args.no_cpu_baseline and not only and young():
        ref = cpu_reference()
        full["cpu_reference"] = ref
        if ref is None or ref.get("kind") == "port":
            full["cpu_reference_recorded"] = CPU_REFERENCE_RECORDED
        if ref:
            best = [v for k, v in ref.items() if k.startswith("AsyncVectorEnv") and isinstance(v, float)]
            if best:
                head["cpu_async_vector_env" + ("_port" if ref.get("kind") == "port" else "")] = float(f"{max(best):.4g}")
        flush()
    head["seconds"] = round(time.time() - t_ref, 1)
    full["headline"] = head
    flush()
    print(json.dumps(head, allow_nan=False))


if __name__ == "__main__":
    main()
