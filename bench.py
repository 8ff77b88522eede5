#!/usr/bin/env python3
"""bench.py -- env-steps/s of the MI355X vector-environment engine (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--env CartPole-v1] [--num-envs 65536] [--inner 128]

Primary workload (BASELINE.json configs[1]): CartPole-v1, num_envs = 65536 PER GPU, random policy.  One bench "step" is one
launch of the hot path over the whole batch: `rollout(inner)` = `inner` lockstep vector steps of all sub-environments with the
random policy `action_space.sample()` evaluated on device (bit-identical to the host policy) and the full trajectory (actions,
observations, rewards, terminated, truncated) written to HBM.  Inputs are resident in HBM when the timed region starts; nothing
crosses PCIe inside it.

value = env-steps/s counted like the reference's benchmark_vector_step (gymnasium/utils/performance.py:88-90: NEXT_STEP autoreset
steps are not counted), whole job over all ranks.  The same JSON line also carries (rank 0, --gpus 1):

  sustained_value   the same launch repeated for >= --sustained seconds (clock / thermal steady state).  Without --steps the timed region
                    itself is ~1 s (K chosen from a 5-launch pilot) and sustained_value repeats `value`
  secondary         BASELINE.json configs[2..4]: Pendulum / Acrobot / MountainCarContinuous @65536, Ant-v5 @32768 and @65536,
                    Humanoid-v5 @32768 (per GPU), each with its own roofline and cpu_baseline
  roofline          dominant kernel: HBM-bound classic kernels as algorithmic bytes / launch time vs 8 TB/s with `traffic` from
                    rocprofv3 PMC passes run BY THIS COMMAND on a short child invocation (FETCH_SIZE x2 + WRITE_SIZE, separate
                    passes); the cooperative MuJoCo kernels are VALU / latency bound: `bound: "valu"`, frac = SQ_ACTIVE_INST_VALU /
                    SQ_WAVE_CYCLES of the same kind of pass, plus the scratch-inclusive traffic ratio
  cpu_baseline      the C oracle (kind "port") on the host's cores: the batch sharded over one single-threaded process per core (count stated), bounded sample
  cpu_reference     Gymnasium's own AsyncVectorEnv (num_envs = os.cpu_count()) / SyncVectorEnv / NumPy CartPoleVectorEnv timed in
                    this run when `import gymnasium` works (GYM_REFERENCE or an installed package); the GPU box has neither, so there
                    the AsyncVectorEnv ARCHITECTURE restated (oracle/async_baseline.py, pinned on the real one by tests/test_async_baseline.py)
                    is timed in this run on this host's cores (kind "port"), and the real gymnasium's numbers from the build container
                    ride along as cpu_reference_recorded (hardware stated)
  api_step_device / api_step_numpy   the per-launch step() API -- never `value`

N > 1: one process per GPU (torchrun), each rank owns its own num_envs sub-environments (global indices rank*num_envs ...; no
data-path collective), one RCCL all-reduce of {env_steps, episodes, return_sum} at the end.  BASELINE.json configs[4]
(Humanoid-v5, 262144 envs over 8 GPUs) is `torchrun --nproc-per-node 8 bench.py --gpus 8 --env Humanoid-v5 --num-envs 32768 --inner 4`.
"""
import argparse
import glob
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per env-step (DESIGN.md "Kernels"): fused rollout = what one env-step must write (+ amortised state)
ROLLOUT_BYTES = {"CartPole-v1": 34, "Pendulum-v1": 26, "Acrobot-v1": 42, "MountainCar-v0": 26, "MountainCarContinuous-v0": 22}
STATE_BYTES = {"CartPole-v1": 96, "Pendulum-v1": 64, "Acrobot-v1": 96, "MountainCar-v0": 64, "MountainCarContinuous-v0": 64}
STEP_BYTES = {"CartPole-v1": 108, "Pendulum-v1": 68, "Acrobot-v1": 116, "MountainCar-v0": 68, "MountainCarContinuous-v0": 64}
# MuJoCo family on the cooperative kernel: a rollout launch = `inner` x [mj_sample_kernel, mj_physics_kernel, mj_step_kernel]; the
# physics kernel is > 98 % of the time (profiles/r01_z_ant_coop_physics.txt)
MJ_COOP = ("Ant-v5", "Humanoid-v5", "HumanoidStandup-v5", "HalfCheetah-v5")
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
# (env, num_envs per GPU, vector steps per launch, launches) of the secondary lines: BASELINE.json configs[2..4] + the north_star's Ant @65536
SECONDARY = [("Pendulum-v1", 65536, 128, 20), ("Acrobot-v1", 65536, 128, 10), ("MountainCarContinuous-v0", 65536, 128, 20),
             ("Ant-v5", 32768, 4, 6), ("Ant-v5", 65536, 4, 4), ("Humanoid-v5", 32768, 4, 3)]
# The other contact regime of the two headline robots: the random policy with `terminate_when_unhealthy` ends a Humanoid
# episode after ~22 steps, so the batch above is mostly robots still upright; with termination off and a warm-up of GROUND_WARM launches every
# robot lies on the ground (many contacts, the PGS sweeps dominate).  A learner that keeps the robot alive lives between the two lines.
GROUND_WARM = 40
SECONDARY_GROUND = [("Ant-v5", 32768, 4), ("Humanoid-v5", 32768, 4)]
F64_PEAK_TFLOPS = 78.6  # MI355X vector fp64 (MI355X_MICROARCH.md): 256 CUs x 4 SIMDs x 16 lanes x 2 flops x 2.4 GHz
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


# ---- CPU legs -----------------------------------------------------------------------------------------------------------------
def _oracle_rollouts(env_id, num_envs, offset, budget_s, start_at=None):
    """One process' share of the CPU baseline: `num_envs` sub-environments (global indices from `offset`) of the C oracle stepping the
    same fused random-policy rollout for ~budget_s seconds.  Returns (env_steps, seconds, vector_steps)."""
    import gymnasium_amd
    from gymnasium_amd import _native
    from oracle import oracle

    env = gymnasium_amd.make_vec(env_id, num_envs=num_envs, env_index_offset=offset, _engine_factory=oracle.engine_factory)
    env.reset(seed=0)
    env.action_space.seed(0)
    eng = env._engine
    eng.action_seed(_native.pcg_words(env.action_space.np_random))
    T = 1 if env_id in MJ_COOP else 4
    obs = np.zeros((T, num_envs) if (eng.obs_dtype is np.int64 and eng.obs_dim == 1) else (T, num_envs, eng.obs_dim), eng.obs_dtype)
    rew, te, tr = np.zeros((T, num_envs)), np.zeros((T, num_envs), np.bool_), np.zeros((T, num_envs), np.bool_)
    acts = np.zeros((T, num_envs) if eng.act_dtype is np.int64 else (T, num_envs, eng.act_dim), dtype=eng.act_dtype)
    eng.rollout(T, None, acts, obs, rew, te, tr)  # warm-up (page faults, first-touch)
    if start_at is not None:  # all workers of a multi-process sample start together
        while time.time() < start_at:
            time.sleep(0.001)
    eng.reset_stats()
    reps = 0
    t0 = time.perf_counter()
    while True:
        eng.rollout(T, None, acts, obs, rew, te, tr)
        reps += 1
        if time.perf_counter() - t0 >= budget_s:
            break
    dt = time.perf_counter() - t0
    steps = eng.stats()["env_steps"]
    env.close()
    return steps, dt, reps * T


def usable_cpus():
    """(logical CPUs this process may actually use, why): os.cpu_count() capped by the scheduler affinity and by the cgroup CPU quota.  The GPU box
    reports 256 logical CPUs (2 x EPYC 9575F, SMT) but runs the job in a cgroup with cpu.max = 16 CPUs' worth of time: more worker processes
    than that only time-slice (measured, scripts/r04/cpu_workers_probe.py: 32 workers 552 M env-steps/s, 64: 505 M, 256: 241 M)."""
    n, why = os.cpu_count() or 1, "os.cpu_count()"
    try:
        a = len(os.sched_getaffinity(0))
        if a < n:
            n, why = a, "sched_getaffinity"
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            q = max(1, -(-int(quota) // int(period)))
            if q < n:
                n, why = q, f"cgroup cpu.max = {quota} {period}"
    except Exception:
        pass
    return n, why


def cpu_baseline(env_id, num_envs, budget_s=12.0, workers=None):
    """The CPU oracle (C restatement of the reference's env + SyncVectorEnv semantics) on the same workload ON THE HOST'S CORES: the batch
    is sharded over `workers` processes (default: one per CPU this job may use -- usable_cpus(): the cgroup quota counts, not the 256 logical CPUs
    the GPU box reports; rounds 1-3 used 64 there, which over-subscribed a 16-CPU quota -- MI355ENV_CPU_WORKERS overrides; each process runs the single-threaded C
    rollout on its contiguous block of sub-environments, like one rank of the GPU job), all started together and run for ~budget_s.
    value = env-steps of all processes / the longest process' time.  kind="port": the Python reference is not present on the GPU box."""
    cores = os.cpu_count() or 1
    usable, why = usable_cpus()
    if workers is None:
        workers = int(os.environ.get("MI355ENV_CPU_WORKERS", usable))
    workers = max(1, min(workers, num_envs))
    if workers == 1:
        steps, dt, vsteps = _oracle_rollouts(env_id, num_envs, 0, budget_s)
        per = [(steps, dt, vsteps)]
    else:
        base, rem = divmod(num_envs, workers)
        start_at = time.time() + 3.0 + 0.02 * workers  # interpreter start + imports + warm-up of every worker (a late one just starts late: every worker times its own window)
        procs, off = [], 0
        env = dict(os.environ, OMP_NUM_THREADS="1")
        for w in range(workers):
            n = base + (1 if w < rem else 0)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", env_id, str(n), str(off), str(budget_s), str(start_at)],
                                          stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, cwd=ROOT, text=True))
            off += n
        per = []
        for p in procs:
            out, _ = p.communicate(timeout=budget_s + 180)
            if p.returncode == 0 and out.strip():
                per.append(tuple(json.loads(out.strip().splitlines()[-1])))
        if len(per) != workers:  # a worker died: fall back to what one process measures
            steps, dt, vsteps = _oracle_rollouts(env_id, num_envs, 0, budget_s)
            per, workers = [(steps, dt, vsteps)], 1
    steps, dt = sum(x[0] for x in per), max(x[1] for x in per)
    return {"value": steps / dt, "unit": "env-steps/s", "cores": workers, "host_cpu_count": cores, "usable_cpus": usable, "usable_cpus_source": why, "kind": "port",
            "per_core_value": steps / dt / workers,
            "sample": f"{env_id} num_envs={num_envs} sharded over {workers} process(es) of the C oracle (one single-threaded rollout loop per core, "
                      f"same random policy, same outputs materialised), {steps} env-steps in {dt:.1f} s"}


def cpu_reference(budget_s=4.0):
    """Gymnasium's own vectorisers on this host's cores, if the package is importable (utils/performance.py:57-103).  Returns None
    where it is not (the GPU box): the caller then carries the numbers recorded in the build container."""
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
    one process per sub-environment, pipes, shared-memory observations, the scalar CartPole in Python; pinned on the real AsyncVectorEnv by
    tests/test_async_baseline.py) and timed by the same counting rule as benchmark_vector_step -- in THIS run, on THIS host's cores.  It carries
    less per-step overhead than the real thing (no PassiveEnvChecker / OrderEnforcing layers, no info-dict assembly): an upper bound of it."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "async_baseline.py")):
        return None
    usable, why = usable_cpus()
    out = {"kind": "port", "what": "oracle/async_baseline.py: AsyncVectorEnv's architecture (vector/async_vector_env.py) around a Python CartPole-v1, "
                                   "NOT gymnasium itself (not installed on this host)", "unit": "env-steps/s", "usable_cpus": usable,
           "usable_cpus_source": why, "host_cpu_count": os.cpu_count(), "how": f"benchmark_vector_step's loop and counting rule, target_duration={budget_s} s"}
    for n in sorted({usable, 4 * usable}):  # the reference's own convention (num_envs = cores) and an over-subscribed one
        key = f"AsyncVectorEnv-port CartPole-v1 num_envs={n}"
        try:  # in a fresh interpreter: its worker processes are forked from a process without a HIP context or RCCL threads
            p = subprocess.run([sys.executable, "-m", "oracle.async_baseline", str(n), str(budget_s)], cwd=ROOT, capture_output=True, text=True, timeout=budget_s * 3 + 120)
            out[key] = float(p.stdout.strip().splitlines()[-1])
        except Exception as e:
            out[key] = f"failed: {type(e).__name__}: {e}"
    return out


# ---- live PMC passes (rocprofv3 on a short child invocation of this file) --------------------------------------------------------
def _rocprof_counters(args, counters, kernel_like, timeout_s):
    """Run `rocprofv3 --pmc <counters> -- python bench.py --child <args>` and return {counter: (avg value per dispatch, dispatches)} of
    the kernels whose name contains `kernel_like`; None if rocprofv3 is absent or the pass fails."""
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    tmp = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [exe, "--pmc", *counters, "--kernel-trace", "-d", tmp, "--", sys.executable, os.path.abspath(__file__), "--child", *args]
    try:
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
        dbs = sorted(glob.glob(tmp + "/**/*.db", recursive=True))
        if not dbs:
            return None
        db = sqlite3.connect(dbs[-1])
        rows = db.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like ? group by counter_name",
                          (f"%{kernel_like}%",)).fetchall()
        return {c: (float(v), int(n)) for c, v, n in rows} or None
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


SQ_COUNTERS = ["SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"]


ISSUE_COUNTERS = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU"]
FLOP_COUNTERS = ["SQ_INSTS_VALU", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_MFMA_MOPS_F64"]
SIMDS, CLOCK_HZ = 1024, 2.4e9  # MI355X: 256 CUs x 4 SIMDs; peak engine clock (MI355X_MICROARCH.md)


DUO_CHUNK = {"CartPole-v1": 4, "MountainCar-v0": 8, "MountainCarContinuous-v0": 8}  # envs_classic.h DUO_ROLLOUT / DUO_CHUNK


def live_counters(env_id, N, inner, kernel, want_traffic, want_sq, timeout_s=150, env_kwargs=None, want_issue=False, want_flops=False, warm=1):
    """HBM traffic per dispatch of `kernel` (WRITE_SIZE + 2 FETCH_SIZE, in KiB on gfx950; MI355X_MICROARCH.md HBM section: separate
    passes, FETCH_SIZE doubled) and, for the VALU-bound kernels, the SQ activity counters as shares of SQ_WAVE_CYCLES."""
    args = ["--env", env_id, "--num-envs", str(N), "--inner", str(inner), "--steps", "3", "--warmup", str(warm), "--env-kwargs", json.dumps(env_kwargs or {})]
    out = {}
    if want_flops:
        # EXECUTED fp64 vector instructions of the kernel, by kind (wave-level: one count per 64-lane instruction, whatever the EXEC mask)
        fc = _rocprof_counters(args, FLOP_COUNTERS, kernel, timeout_s)
        if fc and fc.get("SQ_INSTS_VALU", (0, 0))[0] > 0:
            g = lambda k: fc.get(k, (0.0, 0))[0]  # noqa: E731
            out["flops"] = {"fma_f64": g("SQ_INSTS_VALU_FMA_F64"), "mul_f64": g("SQ_INSTS_VALU_MUL_F64"), "add_f64": g("SQ_INSTS_VALU_ADD_F64"),
                            "trans_f64": g("SQ_INSTS_VALU_TRANS_F64"), "mfma_mops_f64": g("SQ_INSTS_VALU_MFMA_MOPS_F64"), "valu": g("SQ_INSTS_VALU"),
                            "dispatches": fc["SQ_INSTS_VALU"][1], "source": "rocprofv3 --pmc " + " ".join(FLOP_COUNTERS) + " on a child invocation in this run"}
    if want_traffic:
        f = _rocprof_counters(args, ["FETCH_SIZE"], kernel, timeout_s)
        w = _rocprof_counters(args, ["WRITE_SIZE"], kernel, timeout_s) if f else None
        if f and w and "FETCH_SIZE" in f and "WRITE_SIZE" in w:
            out["traffic"] = 1024.0 * (w["WRITE_SIZE"][0] + 2.0 * f["FETCH_SIZE"][0])
            out["traffic_source"] = (f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, FETCH_SIZE doubled) on a child invocation in this run, "
                                     f"{f['FETCH_SIZE'][1]} dispatches of {kernel}")
    if want_issue:
        ic = _rocprof_counters(args, ISSUE_COUNTERS, kernel, timeout_s)
        if ic and ic.get("SQ_WAVES", (0, 0))[0] > 0:
            waves, groups = ic["SQ_WAVES"][0], N / 64.0  # a group = 64 sub-environments: ONE wavefront in the one-role kernels, an env + an aux wavefront in rollout_duo_kernel
            valu, salu = ic["SQ_INSTS_VALU"][0] / groups / inner, ic["SQ_INSTS_SALU"][0] / groups / inner
            per_group = waves / groups
            # a SIMD issues at most one 64-lane VALU instruction per 4 cycles.  With ONE wavefront per SIMD scalar instructions are not hidden behind
            # another wavefront's vector work and take issue slots too; with two (the two-role kernel) they issue beside the partner's VALU work.
            slots = valu + salu if per_group < 1.5 else valu
            out["issue"] = {"valu_per_env_step": valu, "salu_per_env_step": salu, "wavefronts_per_64_envs": per_group,
                            "ceiling_env_steps_per_s": SIMDS * 64 * CLOCK_HZ / (4.0 * slots),
                            "assumptions": (f"{SIMDS} SIMDs x 64 lanes, one instruction per 4 cycles at {CLOCK_HZ / 1e9:.1f} GHz; issue slots per env-step = "
                                            + ("VALU + SALU (one wavefront per SIMD)" if per_group < 1.5 else "VALU (two wavefronts per SIMD: scalar instructions issue beside the partner's vector instructions)")),
                            "source": "rocprofv3 --pmc " + " ".join(ISSUE_COUNTERS) + " on a child invocation in this run"}
    if want_sq:
        sq = _rocprof_counters(args, SQ_COUNTERS, kernel, timeout_s)
        if sq and sq.get("SQ_WAVE_CYCLES", (0, 0))[0] > 0:
            wc = sq["SQ_WAVE_CYCLES"][0]
            out["sq"] = {k[3:].lower() + "_frac": sq[k][0] / wc for k in SQ_COUNTERS[1:] if k in sq}
            out["sq_source"] = "rocprofv3 --pmc " + " ".join(SQ_COUNTERS) + " on a child invocation in this run"
    return out


def recorded_traffic(env_id, N, inner):
    """(bytes per launch, provenance) from profiles/pmc_traffic.json -- PMC passes of an earlier run of the same command; the file's
    `_recorded_at` names the commit / round whose kernels were profiled (a later kernel change does not update it by itself)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(path))
        return rec.get(f"{env_id}:{N}:{inner}"), rec.get("_recorded_at", "unstamped")
    except Exception:
        return None, None


# ---- one configuration on this rank's GPU --------------------------------------------------------------------------------------
class Config:
    def __init__(self, env_id, N, inner, local_rank, rank, env_kwargs=None):
        import torch

        import gymnasium_amd
        from gymnasium_amd import _native

        self.torch, self.env_id, self.N, self.inner, self.env_kwargs = torch, env_id, N, inner, env_kwargs
        dev = torch.device("cuda", local_rank)
        env = gymnasium_amd.make_vec(env_id, num_envs=N, device=local_rank, output="torch", env_index_offset=rank * N, **(env_kwargs or {}))
        env.reset(seed=0)
        env.action_space.seed(rank)
        eng = env._engine
        self.env, self.eng = env, eng
        # preallocated trajectory buffers, reused every launch (a real collector would hand them to the learner)
        act_dtype = torch.int64 if env._discrete else torch.float32
        obs_dtype = {np.float64: torch.float64, np.float32: torch.float32, np.int64: torch.int64}[eng.obs_dtype]
        self.acts = torch.empty((inner, N) if env._discrete else (inner, N, eng.act_dim), dtype=act_dtype, device=dev)
        self.obs = torch.empty((inner, N) if (eng.obs_dtype is np.int64 and eng.obs_dim == 1) else (inner, N, eng.obs_dim), dtype=obs_dtype, device=dev)
        self.rew = torch.empty((inner, N), dtype=torch.float64, device=dev)
        self.te = torch.empty((inner, N), dtype=torch.bool, device=dev)
        self.tr = torch.empty((inner, N), dtype=torch.bool, device=dev)
        env._bind_stream()
        eng.action_seed(_native.pcg_words(env.action_space.np_random))

    def launch(self):
        self.eng.rollout(self.inner, None, self.acts.data_ptr(), self.obs.data_ptr(), self.rew.data_ptr(), self.te.data_ptr(), self.tr.data_ptr())

    def timed(self, K, sync):
        """K launches between two HIP events on the engine's stream (env._bind_stream() = torch's current stream).  One event on either
        side: an event after every launch would put a marker packet between the kernels (+9 us per 96 us launch, measured)."""
        t = self.torch
        ev0, ev1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
        sync()
        self.eng.reset_stats()
        sync()
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(K):
            self.launch()
        ev1.record()
        sync()
        elapsed = time.perf_counter() - t0
        return elapsed, ev0.elapsed_time(ev1) * 1e-3 / K, self.env.statistics()

    def algorithmic_bytes_per_launch(self):
        eng, env = self.eng, self.env
        if self.env_id in ROLLOUT_BYTES:
            per_step, per_launch = ROLLOUT_BYTES[self.env_id], STATE_BYTES[self.env_id]
        else:  # float32 / int64 action row + obs row + reward + 2 flags per env-step
            per_step = (8 if env._discrete else 4 * eng.act_dim) + (8 if eng.obs_dtype is not np.float32 else 4) * eng.obs_dim + 8 + 2
            per_launch = 2 * (8 * eng.state_dim + 4 + 8 + 4)
            if self.env_id in MJ_COOP:  # every vector step is its own set of launches: the state row is read and written per step
                per_step, per_launch = per_step + per_launch, 0
        return (per_step * self.inner + per_launch) * self.N

    def dominant_kernel(self):
        if self.env_id in ROLLOUT_BYTES:
            # the collector's configuration of CartPole / MountainCar / MountainCarContinuous runs the two-role kernel (engine.hip rollout_duo_kernel)
            # when its chunk divides the number of fused steps; MI355ENV_ROLLOUT_DUO=0 is the A/B switch back to rollout_kernel
            chunk = DUO_CHUNK.get(self.env_id)
            if chunk and self.inner % chunk == 0 and os.environ.get("MI355ENV_ROLLOUT_DUO", "1")[:1] != "0":
                return "rollout_duo_kernel"
            return "rollout_kernel"
        if self.eng.obs_dtype is np.int64:
            return "tab_rollout_kernel"
        return "mj_physics_kernel" if self.env_id in MJ_COOP else "mj_rollout_kernel"

    def roofline(self, kernel_s, pmc=(), warm=1, env_steps_per_s=None):
        """kernel_s = average duration of one rollout launch; pmc = which live counter passes to run ("traffic", "sq").  HBM-bound kernels: algorithmic bytes per launch / kernel_s against
        8 TB/s.  The cooperative MuJoCo kernels are VALU / latency bound (one wavefront per SIMD): frac is the
        measured share of wave cycles that issue VALU work, and the HBM side is reported as a traffic ratio."""
        algo = self.algorithmic_bytes_per_launch()
        achieved = algo / kernel_s / 1e9
        kernel = self.dominant_kernel()
        coop = self.env_id in MJ_COOP
        per = self.inner if coop else 1  # dispatches of the dominant kernel per rollout launch
        live = live_counters(self.env_id, self.N, self.inner, kernel, "traffic" in pmc, "sq" in pmc and coop, env_kwargs=self.env_kwargs,
                             want_issue="issue" in pmc and not coop, want_flops="flops" in pmc and coop, warm=max(1, warm)) if pmc else {}
        traffic, src = live.get("traffic"), live.get("traffic_source")
        if traffic is not None:
            traffic *= per
        else:
            traffic, stamp = recorded_traffic(self.env_id, self.N, self.inner)
            src = None if traffic is None else f"profiles/pmc_traffic.json, recorded at {stamp} by a rocprofv3 run of the same command (no live pass in this run)"
        base = {"kernel": kernel, "traffic": traffic, "traffic_source": src, "algorithmic_bytes_per_launch": algo, "avg_kernel_ms": kernel_s * 1e3,
                "traffic_over_algorithmic": (traffic / algo) if traffic else None}
        if coop:
            sq = live.get("sq")
            out = {"bound": "valu", "achieved": (sq or {}).get("active_inst_valu_frac"), "peak": 1.0, "unit": "share of wave cycles issuing VALU instructions (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES), one wavefront per SIMD",
                   "frac": (sq or {}).get("active_inst_valu_frac"), "sq": sq, "sq_source": live.get("sq_source"), "hbm_frac": achieved / HBM_PEAK_GBS,
                   "avg_vector_step_ms": kernel_s * 1e3 / self.inner, **base}
            fl = live.get("flops")
            if fl:
                # A fraction of a PEAK next to the utilisation proxy above: fp64 flops the physics kernel EXECUTES per env-step (2 per FMA, 1 per
                # MUL / ADD, x 64 lanes per wave-level instruction -- lanes switched off by the EXEC mask or carrying no body / dof are counted, so
                # this is the rate the vector units are driven at, an upper bound of the useful rate) x env-steps/s, against the chip's vector fp64 peak.
                per_dispatch = 64.0 * (2.0 * fl["fma_f64"] + fl["mul_f64"] + fl["add_f64"])
                stepping = (env_steps_per_s * kernel_s / self.inner) if env_steps_per_s else float(self.N)  # sub-environments that take a real step per vector step
                fl["flops_per_env_step"] = per_dispatch / max(stepping, 1.0)
                fl["f64_instruction_share_of_valu"] = (fl["fma_f64"] + fl["mul_f64"] + fl["add_f64"] + fl["trans_f64"]) / fl["valu"]
                rate = per_dispatch / (kernel_s / self.inner)  # dispatches of the physics kernel run back to back: one per vector step
                out["flops"] = fl
                out["achieved_tflops_f64"] = rate / 1e12
                out["peak_tflops_f64"] = F64_PEAK_TFLOPS
                out["frac_of_f64_peak"] = rate / 1e12 / F64_PEAK_TFLOPS
            return out
        out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, **base}
        if live.get("issue"):
            # The OTHER ceiling of this kernel: at one wavefront per SIMD the instruction issue rate bounds it before HBM does.
            iss = dict(live["issue"])
            steps_per_s = self.N * self.inner / kernel_s  # lanes stepped per second by this kernel (autoreset lanes included: they execute too)
            iss["achieved_lane_steps_per_s"] = steps_per_s
            iss["frac_of_issue_ceiling"] = steps_per_s / iss["ceiling_env_steps_per_s"]
            iss["hbm_ceiling_env_steps_per_s"] = HBM_PEAK_GBS * 1e9 / (algo / (self.N * self.inner))
            out["issue_bound"] = iss
        return out

    def close(self):
        self.env.close()


T_START = time.perf_counter()


def main(argv=None, harness=None):
    """`harness` is None for every measurement.  tests/bench_dryrun.py (test infrastructure, not reachable from this script's command line)
    passes an object that swaps the Config class and the process-group backend, so that the N > 1 control flow can be executed without GPUs."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed launches K (default: as many as fill ~1 s, so that clocks and thermals are in steady state)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed launches W before the timed region (default: K / 10, at least 5)")
    ap.add_argument("--env", default="CartPole-v1")
    ap.add_argument("--num-envs", type=int, default=65536, help="sub-environments PER GPU")
    ap.add_argument("--inner", type=int, default=128, help="vector steps fused into one launch (one bench step)")
    ap.add_argument("--spinup", type=float, default=0.5, help="seconds of untimed launches before the warm-up, so that the timed region runs at sustained clocks (0 = off)")
    ap.add_argument("--sustained", type=float, default=1.5, help="seconds of back-to-back launches for sustained_value (0 = skip)")
    ap.add_argument("--pmc", choices=["auto", "full", "off"], default="auto",
                    help="live rocprofv3 counter passes on a child invocation: auto = primary traffic, then SQ activity of the MuJoCo secondaries and live "
                         "traffic of every secondary while the run is younger than --pmc-budget seconds (recorded values, stamped with the profile's "
                         "commit, after that); full = everything live; off = recorded values only")
    ap.add_argument("--pmc-budget", type=float, default=150.0, help="auto: no further live traffic passes for the secondaries once the run is this old (s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-api", action="store_true", help="skip the per-launch step() API measurements")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary configurations (BASELINE.json configs[2..4])")
    ap.add_argument("--env-kwargs", default="{}", help='JSON constructor kwargs of the env, e.g. \'{"solver": "Newton"}\' (Humanoid: opt-in solver)')
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)  # profiled child: launches only, prints nothing
    ap.add_argument("--cpu-budget", type=float, default=12.0, help=argparse.SUPPRESS)
    ap.add_argument("--pilot-seconds", type=float, default=1.0, help=argparse.SUPPRESS)
    args = ap.parse_args(argv)

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (already exported on the GPU boxes: the host driver only supports dmabuf IPC, which RCCL needs across ranks)
    import torch

    gpu = harness is None
    backend = "nccl" if gpu else harness.backend
    make_config = Config if gpu else harness.config_cls
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
    if gpu:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank) if gpu else torch.device("cpu")
    N, K, W, inner = args.num_envs, args.steps, args.warmup, args.inner

    def sync_local():
        if gpu:
            torch.cuda.synchronize()

    def sync_all():
        if world > 1:
            dist.barrier()
        sync_local()

    env_kwargs = json.loads(args.env_kwargs)
    cfg = make_config(args.env, N, inner, local_rank, rank, env_kwargs)
    if K is None:  # ~1 s of timed launches: a 50-launch burst is 5 ms, over before the clocks have ramped (round 1: the driver's sampler saw 0 % busy)
        for _ in range(3):
            cfg.launch()
        sync_local()
        t0 = time.perf_counter()
        for _ in range(5):
            cfg.launch()
        sync_local()
        pilot = torch.tensor([(time.perf_counter() - t0) / 5], device=dev)
        if world > 1:
            dist.all_reduce(pilot, op=dist.ReduceOp.MAX)  # every rank must time the same K
        K = int(min(20000, max(20, round(args.pilot_seconds / max(float(pilot.item()), 1e-6)))))
    if W is None:
        W = max(5, K // 10)
    # Clock spin-up (untimed, BEFORE the W warm-up launches, reported as `clock_spinup`): a GPU that has been idle starts in a low power state and
    # takes a few hundred milliseconds of load to reach its sustained clocks -- round 2's 20-launch runs (2 ms) measured 99 us per launch where
    # the steady state is 90.  With explicit --steps the timed region can be that short, so the device is brought to steady state first; with the
    # default (~1 s) timed region the spin-up is the same load a few hundred milliseconds earlier.  --spinup 0 switches it off.
    spin_launches = 0
    if args.spinup > 0 and not args.child:
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < args.spinup:
            for _ in range(8):
                cfg.launch()
            spin_launches += 8
            sync_local()
    for _ in range(W):
        cfg.launch()
    if args.child:
        for _ in range(K):
            cfg.launch()
        sync_local()
        cfg.close()
        return
    elapsed, kernel_s, st = cfg.timed(K, sync_all)
    from gymnasium_amd import distributed as gd

    red = gd.reduce_statistics(st, elapsed_s=elapsed, device=dev)  # the only collective: a few dozen bytes over RCCL/xGMI
    # What the collective itself proves about the job (n_gpus below is NOT read from the environment): gymnasium_amd/distributed.py census()
    cen = gd.census(rank, local_rank, device=dev)
    ranks_seen, devices = cen["ranks"], cen["devices"]
    elapsed = red["elapsed_s"]
    env_steps, episodes, return_sum = float(red["env_steps"]), float(red["episodes"]), float(red["return_sum"])
    single = rank == 0 and world == 1 and gpu
    pmc_primary = ("traffic", "sq", "issue", "flops") if (single and args.pmc != "off") else ()

    result = None
    if rank == 0:
        result = {
            "metric": "env-steps/sec at num_envs=65536 (1/2/4/8 MI355X) vs CPU AsyncVectorEnv",
            "value": env_steps / elapsed, "unit": "env-steps/s", "n_gpus": ranks_seen, "steps": K, "warmup": W,
            "rccl_ranks": ranks_seen, "world_size_env": world, "devices": devices, "distinct_devices": cen["distinct_devices"],
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic", "clock_spinup": {"seconds": args.spinup, "launches": spin_launches, "note": "untimed, before the warmup launches"},
            **({} if gpu else {"engine": harness.label}),
            "config": {"workload": f"{args.env} num_envs={N} per GPU, random policy (on-device action_space.sample()), "
                                   f"NEXT_STEP autoreset, TimeLimit, fused rollout of {inner} vector steps per launch, "
                                   "trajectory (actions, obs, rewards, terminated, truncated) written to HBM",
                       "env": args.env, "env_kwargs": env_kwargs, "num_envs_per_gpu": N, "vector_steps_per_launch": inner,
                       "parallelism": f"env-sharded x{world} (no data-path collective)"},
            "episodes": episodes, "mean_episode_return": (return_sum / episodes) if episodes else None,
        }

    # ---- sustained: the same launch back to back for >= args.sustained seconds (all ranks, same barrier discipline) ----------------
    if args.sustained > 0 and elapsed >= 0.8:  # (elapsed is the all-reduced maximum: every rank takes the same branch)
        if rank == 0:
            result["sustained_value"] = result["value"]
            result["sustained"] = {"launches": K, "seconds": elapsed, "avg_kernel_ms": kernel_s * 1e3, "note": "the timed region itself is the sustained measurement"}
    elif args.sustained > 0:
        Ks = max(K, int(args.sustained / max(kernel_s, 1e-7)) + 1)
        el_s, k_s, st_s = cfg.timed(Ks, sync_all)
        red_s = gd.reduce_statistics(st_s, elapsed_s=el_s, device=dev)
        if rank == 0:
            result["sustained_value"] = float(red_s["env_steps"]) / red_s["elapsed_s"]
            result["sustained"] = {"launches": Ks, "seconds": red_s["elapsed_s"], "avg_kernel_ms": k_s * 1e3}
    if rank == 0:
        result["roofline"] = cfg.roofline(kernel_s, pmc_primary, env_steps_per_s=result["value"])

    # ---- the per-launch step() API (rank 0, one GPU): device tensors and the NumPy path ----------------------------------------------
    if single and not args.no_api:
        import gymnasium_amd

        env, eng = cfg.env, cfg.eng
        a_dev = torch.randint(0, 2, (N,), device=dev) if env._discrete else (torch.rand((N, eng.act_dim), device=dev) * 0.8 - 0.4)
        env.copy = False
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
        result["api_step_device"] = {"value": N * reps / dt, "unit": "vector-env lanes/s (incl. autoreset lanes)",
                                     "us_per_step_wall": dt / reps * 1e6, "us_per_step_gpu": step_kernel_s * 1e6,
                                     "roofline_frac": (STEP_BYTES[args.env] * N / step_kernel_s / 1e9 / HBM_PEAK_GBS) if args.env in STEP_BYTES else None}
        # the same steps as ONE HIP graph (HipVectorEnv.capture_steps): what is left when the host is out of the loop
        G_STEPS, g_reps = 32, 30
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
        result["api_step_graph"] = {"value": N * G_STEPS * g_reps / dtg, "unit": "vector-env lanes/s (incl. autoreset lanes)", "steps_per_graph": G_STEPS,
                                    "us_per_step_wall": dtg / (G_STEPS * g_reps) * 1e6, "us_per_step_gpu": e0.elapsed_time(e1) * 1e3 / (G_STEPS * g_reps),
                                    "roofline_frac": (STEP_BYTES[args.env] * N / (dtg / (G_STEPS * g_reps)) / 1e9 / HBM_PEAK_GBS) if args.env in STEP_BYTES else None,
                                    "what": f"{G_STEPS} step() calls captured into one hipGraph, replayed {g_reps} times"}
        del graphed
        env_np = gymnasium_amd.make_vec(args.env, num_envs=N, device=local_rank, copy=False)
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
                if use_pinned:  # the caller's policy writes straight into the pinned upload array
                    env_np.step(env_np.action_buffer)
                else:
                    env_np.step(actions[k % 8])
            return (time.perf_counter() - t0) / reps

        dt_page = timed_steps(False)
        env_np.action_buffer[...] = actions[0]
        dt_pin = timed_steps(True)
        result["api_step_numpy"] = {"value": N / dt_page, "unit": "vector-env lanes/s (NumPy in / NumPy out over PCIe: the step kernel reads the actions from and writes its outputs to one page-locked block, no copy-engine hand-off; "
                                                                 "host action sampling excluded)",
                                    "us_per_step_wall": dt_page * 1e6, "us_per_step_wall_actions_in_pinned_buffer": dt_pin * 1e6,
                                    "host_action_space_sample_us": sample_us}
        env_np.close()
        # the reference's stateful wrappers on top (ClipReward(NormalizeReward(NormalizeObservation(env)))), device tensors in and out
        if args.env in STEP_BYTES:
            from gymnasium_amd import wrappers as gw

            wrapped = {}
            for mode in ("fused", "standalone"):
                env_w = gymnasium_amd.make_vec(args.env, num_envs=N, device=local_rank, output="torch", copy=False)
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
            result["api_step_wrapped"] = {"wrappers": "ClipReward(NormalizeReward(NormalizeObservation(env)))", "us_per_step_wall_fused": wrapped["fused"],
                                          "us_per_step_wall_standalone_passes": wrapped["standalone"], "launches_per_step_fused": 2 if N <= 262144 else 3,
                                          "launches_per_step_standalone": 11, "evidence": "profiles/r02_wrappers_fused.txt"}
    cfg.close()
    if single and args.env in STEP_BYTES and not env_kwargs:  # the opt-in configuration next to the default (bit-exact libm) one
        c_opt = Config(args.env, N, inner, local_rank, 0, {"fast_math": True})
        for _ in range(2):
            c_opt.launch()
        el_o, _, st_o = c_opt.timed(K, sync_local)
        result["opt_in"] = {"env_kwargs": {"fast_math": True}, "value": st_o["env_steps"] / el_o, "unit": "env-steps/s",
                            "note": "device sin / cos and x * x instead of the bit-exact libm restatements (tolerance parity, tests/test_gpu_parity.py)"}
        c_opt.close()

    # ---- secondary configurations (rank 0, one GPU) -------------------------------------------------------------------------------------
    if single and not args.no_secondary and args.env == "CartPole-v1":
        result["secondary"] = []
        def steady(c, seconds=0.7):
            """~`seconds` of back-to-back launches after a tenth of that as warm-up: (env-steps/s, launches, avg kernel s)"""
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
            el, ks, st = c.timed(k, sync_local)
            return st["env_steps"] / el, k, ks, el

        for env_id, n2, inner2, k2 in SECONDARY:
            c2 = Config(env_id, n2, inner2, local_rank, 0)
            v2, k2, ks2, el2 = steady(c2)
            line = {"env": env_id, "num_envs": n2, "vector_steps_per_launch": inner2, "launches": k2, "value": v2, "sustained_value": v2, "seconds": el2,
                    "unit": "env-steps/s", "ms_per_launch": el2 / k2 * 1e3, "dtype": "f64"}
            # full: everything live.  auto: the SQ-activity pass for the VALU-bound MuJoCo kernels, and live FETCH / WRITE passes too while
            # the run is young enough (each pass is a child process of ~5-10 s); after that the recorded profile's value, stamped as such
            young = (time.perf_counter() - T_START) < args.pmc_budget
            want = ("traffic", "sq", "flops") if args.pmc == "full" else ((("traffic",) if young else ()) + (("sq", "flops") if env_id in MJ_COOP else ()) if args.pmc == "auto" else ())
            line["roofline"] = c2.roofline(ks2, want, env_steps_per_s=v2)
            c2.close()
            opt = {"fast_math": True} if env_id in STEP_BYTES else ({"solver": "Newton"} if env_id in ("Humanoid-v5", "HumanoidStandup-v5") else None)
            if opt:  # the opt-in, faster configuration next to the default (reference-faithful) one
                c3 = Config(env_id, n2, inner2, local_rank, 0, opt)
                line["opt_in"] = {"env_kwargs": opt, "value": steady(c3)[0], "unit": "env-steps/s"}
                c3.close()
            if not args.no_cpu_baseline:  # MuJoCo: a 512-env sample (the oracle's per-env cost does not depend on the batch size)
                line["cpu_baseline"] = cpu_baseline(env_id, min(n2, 64 * usable_cpus()[0]) if env_id in MJ_COOP else n2, budget_s=3.0)
            result["secondary"].append(line)

        # the on-the-ground regime of the two headline robots: termination off, GROUND_WARM launches before anything is timed or profiled
        for env_id, n2, inner2 in SECONDARY_GROUND:
            kw = {"terminate_when_unhealthy": False}
            c2 = Config(env_id, n2, inner2, local_rank, 0, kw)
            for _ in range(GROUND_WARM):
                c2.launch()
            v2, k2, ks2, el2 = steady(c2)
            line = {"env": env_id, "num_envs": n2, "vector_steps_per_launch": inner2, "launches": k2, "value": v2, "sustained_value": v2, "seconds": el2,
                    "unit": "env-steps/s", "ms_per_launch": el2 / k2 * 1e3, "dtype": "f64", "env_kwargs": kw,
                    "regime": f"robots on the ground: terminate_when_unhealthy=False, {GROUND_WARM} launches ({GROUND_WARM * inner2} vector steps) of warm-up before the timed region"}
            want = ("sq", "flops") if args.pmc in ("auto", "full") else ()
            line["roofline"] = c2.roofline(ks2, want, warm=GROUND_WARM, env_steps_per_s=v2)
            c2.close()
            result["secondary"].append(line)

    # ---- CPU legs: the oracle port on rank 0 (every N), the reference's own vectorisers where importable -------------------------------
    if rank == 0:
        # the whole batch of one GPU on every host core (MuJoCo: a bounded sample of 64 sub-environments per core -- the oracle's per-env cost
        # does not depend on the batch size)
        n_cpu = min(N, 64 * usable_cpus()[0]) if args.env in MJ_COOP else N
        result["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(args.env, n_cpu, budget_s=args.cpu_budget)
        if not args.no_cpu_baseline and world == 1 and gpu:
            ref = cpu_reference()
            result["cpu_reference"] = ref  # gymnasium's own vectorisers where importable, else the AsyncVectorEnv port (kind "port"), timed in this run
            if ref is None or ref.get("kind") == "port":
                result["cpu_reference_recorded"] = CPU_REFERENCE_RECORDED  # the real gymnasium's numbers from the build container, hardware stated
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":  # one process of cpu_baseline's multi-core sample
        _, _, w_env, w_n, w_off, w_budget, w_start = sys.argv
        print(json.dumps(_oracle_rollouts(w_env, int(w_n), int(w_off), float(w_budget), float(w_start))))
    else:
        main()
