#!/usr/bin/env python3
"""bench.py -- env-steps/s of the MI355X vector-environment engine (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--env CartPole-v1] [--num-envs 65536] [--inner 128]

Primary workload (BASELINE.json configs[1]): CartPole-v1, num_envs = 65536 PER GPU, random policy.  One bench "step" is one launch of the
hot path over the whole batch: `mi_rollout(inner)` = `inner` lockstep vector steps of all sub-environments with the random policy
`action_space.sample()` evaluated on device and the full trajectory (actions, observations, rewards, terminated, truncated) written to
HBM.  Inputs are resident in HBM when the timed region starts; nothing crosses PCIe inside it.

Order of events on every rank: W untimed warm-up launches (after an untimed clock spin-up) -> the sub-environments are RE-ARMED
(`reset(seed=0)`, policy stream seeded with the rank: a few microseconds, outside the timed region) -> barrier + synchronise -> EXACTLY K
timed launches between two HIP events -> synchronise + barrier.  Re-arming makes the first timed launch a known-answer computation: its
trajectory lands in its own buffers, `output_sha256` is the digest of those bytes, and the `verified` object says how they compare with
the CPU oracle run on the same seeds in this run (classic control / ToyText: every byte of all sub-environments; MuJoCo kinds: a strided
subset within 1e-8).  tests/test_bench_digest.py computes the same digest on the CPU for configs[1]'s exact shape.

value = env-steps/s counted like the reference's benchmark_vector_step (gymnasium/utils/performance.py:88-90: NEXT_STEP autoreset steps
are not counted), whole job over all ranks.  The ONE stdout line is kept under 4 KB (round 4's 20 KB line could not be parsed by the
driver); everything else -- BASELINE.json configs[2..4] with their own rooflines, the per-launch step() API, the opt-in configurations, the
reference's own vectorisers -- is measured by scripts/bench_extras.py (a child process of this command at --gpus 1) and written to the
sidecar file the line names in `full`; the line carries only their headline values in `secondary`.

N > 1: one process per GPU (torchrun), each rank owns its own num_envs sub-environments (global indices rank*num_envs ...; no data-path
collective), one RCCL all-reduce of {env_steps, episodes, return_sum} at the end.  BASELINE.json configs[4] (Humanoid-v5, 262144 envs
over 8 GPUs) is `torchrun --nproc-per-node 8 bench.py --gpus 8 --env Humanoid-v5 --num-envs 32768 --inner 4`.
"""
import argparse
import copy
import glob
import hashlib
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

METRIC = "env-steps/sec at num_envs=65536 (1/2/4/8 MI355X) vs CPU AsyncVectorEnv"
LINE_LIMIT = 4096  # bytes of the one stdout line (tests/test_gpu_bench_contract.py)
# algorithmic bytes per env-step (DESIGN.md "Kernels"): fused rollout = what one env-step must write (+ the state row once per launch)
ROLLOUT_BYTES = {"CartPole-v1": 34, "Pendulum-v1": 26, "Acrobot-v1": 42, "MountainCar-v0": 26, "MountainCarContinuous-v0": 22}
STATE_BYTES = {"CartPole-v1": 96, "Pendulum-v1": 64, "Acrobot-v1": 96, "MountainCar-v0": 64, "MountainCarContinuous-v0": 64}
# MuJoCo family on the cooperative kernel: a rollout launch = `inner` x [mj_sample_kernel, mj_physics_kernel, mj_step_kernel]
MJ_COOP = ("Ant-v5", "Humanoid-v5", "HumanoidStandup-v5", "HalfCheetah-v5", "Walker2d-v5")
MJ_IDS = MJ_COOP + ("Hopper-v5", "InvertedPendulum-v5", "InvertedDoublePendulum-v5", "Reacher-v5", "Swimmer-v5", "Pusher-v5")
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
DUO_CHUNK = {"CartPole-v1": 8, "Pendulum-v1": 8, "MountainCar-v0": 8, "MountainCarContinuous-v0": 8}  # envs_classic.h DUO_ROLLOUT / DUO_CHUNK


# ---- CPU legs: the oracle as the timed baseline and as the checker of the first timed launch ----------------------------------------
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
    is sharded over `workers` processes (default: one per CPU this job may use -- usable_cpus(): the cgroup quota counts, not the 256 logical
    CPUs the GPU box reports; MI355ENV_CPU_WORKERS overrides; each process runs the single-threaded C rollout on its contiguous block of
    sub-environments, like one rank of the GPU job), all started together and run for ~budget_s.
    value = env-steps of all processes / the longest process' time.  kind="port": the Python reference is not present on the GPU box."""
    cores = os.cpu_count() or 1
    usable, why = usable_cpus()
    if workers is None:
        workers = int(os.environ.get("MI355ENV_CPU_WORKERS", usable))
    workers = max(1, min(workers, num_envs))
    if workers == 1:
        per = [_oracle_rollouts(env_id, num_envs, 0, budget_s)]
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
            per, workers = [_oracle_rollouts(env_id, num_envs, 0, budget_s)], 1
    steps, dt = sum(x[0] for x in per), max(x[1] for x in per)
    return {"value": steps / dt, "unit": "env-steps/s", "cores": workers, "kind": "port", "host_cpu_count": cores, "usable_cpus": usable,
            "usable_cpus_source": why, "per_core_value": steps / dt / workers,
            "sample": f"{env_id} num_envs={num_envs} over {workers} single-threaded C-oracle process(es), same random policy, same outputs written, "
                      f"{steps} env-steps in {dt:.1f} s"}


def trajectory_digest(traj) -> str:
    """sha256 over the bytes of (actions, observations, rewards, terminated, truncated), time-major, in that order."""
    h = hashlib.sha256()
    for a in traj:
        h.update(np.ascontiguousarray(a).view(np.uint8).reshape(-1).data)
    return h.hexdigest()


def oracle_trajectory(env_id, N, T, offset=0, policy_seed=0, env_kwargs=None, actions=None, env_indices=None):
    """The reference computation of one bench launch on the CPU checker: sub-environments with global indices `offset + env_indices` (default
    all N) are reset with seed 0 (env g <- seed 0 + g, sync_vector_env.py:207-208) and stepped T times, either under the random policy
    `action_space.seed(policy_seed); action_space.sample()` or teacher-forced with `actions`.  Returns (actions, obs, rewards, terminated, truncated)."""
    import gymnasium_amd
    from gymnasium_amd import _native
    from oracle import oracle

    if env_indices is None:
        n, kw, seed = N, {"env_index_offset": offset}, 0
    else:
        n, kw, seed = len(env_indices), {}, [int(offset + g) for g in env_indices]
    env = gymnasium_amd.make_vec(env_id, num_envs=n, _engine_factory=oracle.engine_factory, **kw, **(env_kwargs or {}))
    env.reset(seed=seed)
    eng = env._engine
    obs = np.zeros((T, n) if (eng.obs_dtype is np.int64 and eng.obs_dim == 1) else (T, n, eng.obs_dim), eng.obs_dtype)
    rew, te, tr = np.zeros((T, n)), np.zeros((T, n), np.bool_), np.zeros((T, n), np.bool_)
    if actions is None:
        env.action_space.seed(policy_seed)
        eng.action_seed(_native.pcg_words(env.action_space.np_random))
        acts = np.zeros((T, n) if eng.act_dtype is np.int64 else (T, n, eng.act_dim), dtype=eng.act_dtype)
        eng.rollout(T, None, acts, obs, rew, te, tr)
    else:
        acts = np.ascontiguousarray(actions)
        eng.rollout(T, acts, None, obs, rew, te, tr)
    env.close()
    return acts, obs, rew, te, tr


def oracle_check(cfg, traj):
    """`traj` = the first timed launch of this rank, on the host.  (1) its actions are the host policy's: `action_space.seed(rank)` + T x
    `sample()` of the batched space (spaces/multi_discrete.py:176-178, spaces/box.py:463-465); (2) the oracle, reset with the same seeds and
    teacher-forced with those actions, produces the same observations / rewards / flags: every sub-environment, bit for bit, for the
    bit-exact kinds (classic control, ToyText); <= 1 024 strided sub-environments within 1e-8 for the MuJoCo kinds, whose
    later steps mujoco_window_check covers."""
    acts, obs, rew, te, tr = traj
    T, N = cfg.inner, cfg.N
    if (cfg.env_kwargs or {}).get("fast_math"):  # tolerance parity by design; whole-launch equality does not apply (chaotic kinds diverge within a launch)
        return {"ok": None, "skipped": "fast_math=True: within 1 ulp per libm call of the reference, asserted with re-synchronisation by tests/test_gpu_parity.py"}
    sp = copy.deepcopy(cfg.env.action_space)
    sp.seed(cfg.rank)
    host_acts = np.stack([sp.sample() for _ in range(T)]).reshape(acts.shape)
    policy_ok = bool(np.array_equal(host_acts.astype(acts.dtype), acts))
    exact = cfg.env_id not in MJ_IDS
    stride = 1 if exact else max(1, N // 1024)
    idx = None if stride == 1 else np.arange(0, N, stride)
    sel = (lambda a: a) if idx is None else (lambda a: np.ascontiguousarray(a[:, idx]))
    _, o2, r2, te2, tr2 = oracle_trajectory(cfg.env_id, N, T, offset=cfg.rank * N, env_kwargs=cfg.env_kwargs, actions=sel(acts), env_indices=idx)
    flags_ok = bool(np.array_equal(sel(te), te2) and np.array_equal(sel(tr), tr2))
    if exact:
        vals_ok = bool(np.array_equal(sel(obs), o2) and np.array_equal(sel(rew), r2))
        worst = 0.0 if vals_ok else float(max(np.nanmax(np.abs(sel(obs).astype(np.float64) - o2)), np.nanmax(np.abs(sel(rew) - r2))))
    else:
        worst = float(max(np.max(np.abs(sel(obs) - o2)), np.max(np.abs(sel(rew) - r2))))
        vals_ok = worst <= 1e-8
    return {"ok": policy_ok and flags_ok and vals_ok, "against": "oracle/ (C restatement), same seeds, in this run", "envs": N if idx is None else len(idx),
            "steps": T, "compare": "array_equal" if exact else "atol 1e-8", "max_abs_diff": worst, "policy_equals_host_sample": policy_ok}


def mujoco_window_check(cfg, robots=1024, windows=10, threads=None):
    """The MuJoCo kinds beyond their first launch: `windows` further launches of `cfg.inner` vector steps from WHEREVER the
    sub-environments are now -- after the timed region, i.e. late in their episodes, or lying on the ground when the configuration switched termination
    off and warmed up -- each compared on `robots` strided sub-environments with the oracle started from the engine's own state, generator words and
    TimeLimit counters at the window's start and teacher-forced with the engine's actions: observations and rewards within 1e-8, flags equal.  (The
    oracle is re-synchronised per window because articulated bodies in contact amplify a 1e-12 difference past any tolerance within tens of steps; inside a
    window of `inner` steps the agreement is what DESIGN.md section 4 states.)  The oracle's robots are stepped by `threads` host threads (ctypes drops the GIL)."""
    import concurrent.futures as cf

    import gymnasium_amd
    from oracle import oracle

    env, eng, N, T = cfg.env, cfg.eng, cfg.N, cfg.inner
    idx = np.arange(0, N, max(1, N // robots))[:robots]
    threads = threads or max(1, min(usable_cpus()[0], len(idx) // 16 or 1))
    slices = [s for s in np.array_split(np.arange(len(idx)), threads) if len(s)]
    kw = {k: v for k, v in (cfg.env_kwargs or {}).items()}
    checkers = [gymnasium_amd.make_vec(cfg.env_id, num_envs=len(s), _engine_factory=oracle.engine_factory, **kw) for s in slices]
    for c in checkers:
        c.reset(seed=0)
    worst, flags_ok, finished = 0.0, True, 0
    bufs = cfg.alloc_trajectory()
    for _ in range(windows):
        state, elapsed, flags = env.get_state()
        words = env.get_rng_state()
        cfg.launch(bufs)
        acts, obs, rew, te, tr = tuple(b.cpu().numpy() for b in bufs[0])

        def one(k):
            sl, c = idx[slices[k]], checkers[k]
            c._engine.seed(np.ascontiguousarray(words[sl]), None)
            c.set_state(state[sl], elapsed[sl], flags[sl])
            o2, r2 = np.zeros((T, len(sl), eng.obs_dim)), np.zeros((T, len(sl)))
            te2, tr2 = np.zeros((T, len(sl)), np.bool_), np.zeros((T, len(sl)), np.bool_)
            c._engine.rollout(T, np.ascontiguousarray(acts[:, sl]), None, o2, r2, te2, tr2)
            d = max(float(np.max(np.abs(obs[:, sl] - o2))), float(np.max(np.abs(rew[:, sl] - r2))))
            return d, bool(np.array_equal(te[:, sl], te2) and np.array_equal(tr[:, sl], tr2)), int(np.count_nonzero(te2 | tr2))

        with cf.ThreadPoolExecutor(len(slices)) as pool:
            for d, ok, fin in pool.map(one, range(len(slices))):
                worst, flags_ok, finished = max(worst, d), flags_ok and ok, finished + fin
    for c in checkers:
        c.close()
    return {"ok": bool(flags_ok and worst <= 1e-8), "envs": int(len(idx)), "steps": int(windows * T), "window_steps": int(T), "max_abs_diff": worst,
            "episodes_finished_inside": finished, "compare": "atol 1e-8 per window, oracle re-synchronised to the engine's state at every window start",
            "host_threads": len(slices)}


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


def tabular_kernel(env_id, env_kwargs=None):
    """Which rollout kernel mi_rollout launches for a ToyText id with the on-device policy in NEXT_STEP mode (engine.hip): the branch-free kernels for
    Blackjack and for one plain table of <= 3 outcomes per (state, action) that fits into LDS in its packed form; MI355ENV_TAB_LEAN=0 is the A/B switch."""
    kw = env_kwargs or {}
    if os.environ.get("MI355ENV_TAB_LEAN", "1")[:1] == "0":
        return "tab_rollout_kernel"
    if env_id.startswith("Blackjack"):
        return "bj_rollout_lean_kernel"
    if kw.get("fickle_passenger") or kw.get("is_rainy") or ("map_name" in kw and kw["map_name"] is None) or isinstance(kw.get("desc"), (list, tuple)) and kw["desc"] and not isinstance(kw["desc"][0], str):
        return "tab_rollout_kernel"  # Taxi's fickle rule, 3 x 3 000 outcomes (192 KB packed), one table per sub-environment
    return "tab_rollout_lean_kernel"


def child_args(env_id, N, inner, env_kwargs=None, warm=1):
    return ["--env", env_id, "--num-envs", str(N), "--inner", str(inner), "--steps", "3", "--warmup", str(warm), "--env-kwargs", json.dumps(env_kwargs or {})]


def live_traffic(env_id, N, inner, kernel, env_kwargs=None, warm=1, timeout_s=150):
    """HBM bytes per dispatch of `kernel`: WRITE_SIZE + 2 x FETCH_SIZE, in KiB on gfx950, from two separate rocprofv3 --pmc passes
    (MI355X_MICROARCH.md, HBM section).  (bytes, provenance) or (None, None)."""
    args = child_args(env_id, N, inner, env_kwargs, warm)
    f = _rocprof_counters(args, ["FETCH_SIZE"], kernel, timeout_s)
    w = _rocprof_counters(args, ["WRITE_SIZE"], kernel, timeout_s) if f else None
    if f and w and "FETCH_SIZE" in f and "WRITE_SIZE" in w:
        return 1024.0 * (w["WRITE_SIZE"][0] + 2.0 * f["FETCH_SIZE"][0]), f"live rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes in this run ({f['FETCH_SIZE'][1]} dispatches)"
    return None, None


def kernel_tag(env_id, kernel):
    """A fragment of the dominant kernel's demangled name that tells this env's instantiation from every other env's (one rocprofv3 pass can then
    serve several configurations: live_traffic_batch)."""
    classic = {"CartPole-v1": "CartPoleT<", "Pendulum-v1": "PendulumT<", "Acrobot-v1": "AcrobotT<", "MountainCar-v0": "MountainCarT<",
               "MountainCarContinuous-v0": "MountainCarContinuousT<"}
    if env_id in classic:
        return f"{kernel}<mi::{classic[env_id]}"
    if kernel in ("mj_physics_kernel", "mj_rollout_kernel"):
        return f"{kernel}<mjx::MjEnv<mjx::{env_id.split('-')[0]}Model,"
    return kernel  # the tabular kernels are one instantiation for every table: such configurations get a pass of their own


def live_counters_batch(configs, extra_sets=None, timeout_s=240):
    """rocprofv3 counter passes for several configurations with a handful of child processes instead of several per configuration: a child runs the
    configurations of a group one after the other (`--child-list`), each one's dispatches are told apart by kernel_tag, and every counter SET is one pass
    over the group (FETCH_SIZE and WRITE_SIZE each on their own, as MI355X_MICROARCH.md prescribes for the HBM bytes; `extra_sets`: name -> counters, e.g.
    the SQ activity and the fp64 instruction mix of the cooperative MuJoCo kernels).  configs: dicts with env_id, N, inner, env_kwargs, warm, kernel, key.
    Returns ({key: (HBM bytes per dispatch, provenance)}, {key: {set name: {counter: (avg per dispatch, dispatches)}}}).  Configurations whose tag another
    member of a group shares (the tabular kernels; one robot at two sizes or in two regimes) go into further groups."""
    groups = []
    for c in configs:
        tag = kernel_tag(c["env_id"], c["kernel"])
        for g in groups:
            if all(tag not in t and t not in tag for t in g["tags"]):
                g["tags"].append(tag), g["members"].append((c, tag))
                break
        else:
            groups.append({"tags": [tag], "members": [(c, tag)]})
    traffic, extra = {}, {}
    for g in groups:
        spec = json.dumps([[c["env_id"], c["N"], c["inner"], c.get("env_kwargs") or {}, c.get("warm", 1)] for c, _ in g["members"]])
        tags = [tag for _, tag in g["members"]]
        res = {counter: _rocprof_counters_multi(["--child-list", spec], [counter], tags, timeout_s) for counter in ("FETCH_SIZE", "WRITE_SIZE")}
        wanted = {name: ctrs for name, ctrs in (extra_sets or {}).items() if any(c.get("extra") for c, _ in g["members"])}
        more = {name: _rocprof_counters_multi(["--child-list", spec], ctrs, tags, timeout_s) for name, ctrs in wanted.items()}
        for c, tag in g["members"]:
            f, w = (res["FETCH_SIZE"] or {}).get(tag), (res["WRITE_SIZE"] or {}).get(tag)
            if f and w and "FETCH_SIZE" in f and "WRITE_SIZE" in w:
                traffic[c["key"]] = (1024.0 * (w["WRITE_SIZE"][0] + 2.0 * f["FETCH_SIZE"][0]),
                                     f"live rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes in this run ({f['FETCH_SIZE'][1]} dispatches; one child process for {len(g['members'])} configurations)")
            else:
                traffic[c["key"]] = (None, None)
            if c.get("extra"):
                extra[c["key"]] = {name: (r or {}).get(tag) for name, r in more.items()}
    return traffic, extra


def live_traffic_batch(configs, timeout_s=240):
    return live_counters_batch(configs, None, timeout_s)[0]


def _rocprof_counters_multi(args, counters, kernel_likes, timeout_s):
    """_rocprof_counters for several kernel-name fragments out of ONE profiled run: {fragment: {counter: (avg per dispatch, dispatches)}}."""
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
        out = {}
        for like in kernel_likes:
            rows = db.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like ? group by counter_name",
                              (f"%{like}%",)).fetchall()
            out[like] = {c: (float(v), int(n)) for c, v, n in rows} or None
        return out
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def recorded_traffic(env_id, N, inner):
    """(bytes per launch, provenance) from profiles/pmc_traffic.json -- PMC passes of an earlier run of the same command."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        v = rec.get(f"{env_id}:{N}:{inner}")
        return v, (None if v is None else f"profiles/pmc_traffic.json ({rec.get('_recorded_at', 'unstamped')}); no live pass in this run")
    except Exception:
        return None, None


# ---- one configuration on this rank's GPU --------------------------------------------------------------------------------------
class Config:
    """One (env id, num_envs, fused steps) configuration on this rank's GPU: the env, two sets of trajectory buffers (`first`: written by
    the first timed launch only and read back for the digest; `rest`: reused by every other launch, as a collector hands them to the
    learner), the launch, and the timed region.  tests/bench_dryrun.py subclasses the four device-specific methods."""

    def __init__(self, env_id, N, inner, local_rank, rank, env_kwargs=None):
        self.env_id, self.N, self.inner, self.local_rank, self.rank, self.env_kwargs = env_id, N, inner, local_rank, rank, env_kwargs
        self.env = self.make_env()
        self.eng = self.env._engine
        self.first, self.rest = self.alloc_trajectory(), self.alloc_trajectory()
        self.rearm()

    # -- device-specific ----------------------------------------------------------------------------------
    def make_env(self):
        import gymnasium_amd

        return gymnasium_amd.make_vec(self.env_id, num_envs=self.N, device=self.local_rank, output="torch", env_index_offset=self.rank * self.N,
                                      **(self.env_kwargs or {}))

    def alloc_trajectory(self):
        import torch

        env, eng, T, N = self.env, self.eng, self.inner, self.N
        dev = torch.device("cuda", self.local_rank)
        obs_dtype = {np.float64: torch.float64, np.float32: torch.float32, np.int64: torch.int64}[eng.obs_dtype]
        bufs = (torch.empty((T, N) if env._discrete else (T, N, eng.act_dim), dtype=torch.int64 if env._discrete else torch.float32, device=dev),
                torch.empty((T, N) if (eng.obs_dtype is np.int64 and eng.obs_dim == 1) else (T, N, eng.obs_dim), dtype=obs_dtype, device=dev),
                torch.empty((T, N), dtype=torch.float64, device=dev), torch.empty((T, N), dtype=torch.bool, device=dev),
                torch.empty((T, N), dtype=torch.bool, device=dev))
        return bufs, tuple(b.data_ptr() for b in bufs)

    def host_trajectory(self):
        return tuple(b.cpu().numpy() for b in self.first[0])

    def timed(self, K, sync):
        """K launches between two HIP events on the engine's stream (env._bind_stream() = torch's current stream).  One event on either
        side: an event after every launch would put a marker packet between the kernels (+9 us per 96 us launch, measured).
        Returns (wall seconds, average seconds per launch by the events, statistics of the K launches)."""
        import torch

        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sync()
        self.eng.reset_stats()
        sync()
        t0 = time.perf_counter()
        ev0.record()
        self.launch(self.first)
        for _ in range(K - 1):
            self.launch()
        ev1.record()
        sync()
        elapsed = time.perf_counter() - t0
        return elapsed, ev0.elapsed_time(ev1) * 1e-3 / K, self.env.statistics()

    # -- common --------------------------------------------------------------------------------------------
    def rearm(self):
        """Bring the sub-environments and the policy stream to the known start: reset(seed=0) (env g <- seed g, global index) and
        `action_space.seed(rank)` handed to the engine.  The launch that follows is reproducible by anyone (oracle_trajectory)."""
        from gymnasium_amd import _native

        self.env.reset(seed=0)
        self.env.action_space.seed(self.rank)
        self.env._bind_stream()
        self.eng.action_seed(_native.pcg_words(self.env.action_space.np_random))

    def launch(self, bufs=None):
        p = (bufs or self.rest)[1]
        self.eng.rollout(self.inner, None, p[0], p[1], p[2], p[3], p[4])

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
            return tabular_kernel(self.env_id, self.env_kwargs)
        return "mj_physics_kernel" if self.env_id in MJ_COOP else "mj_rollout_kernel"

    def traffic_request(self, key, warm=1):
        """This configuration as live_traffic_batch wants it."""
        return {"env_id": self.env_id, "N": self.N, "inner": self.inner, "env_kwargs": self.env_kwargs, "warm": warm, "kernel": self.dominant_kernel(), "key": key}

    def roofline(self, kernel_s, live=False, warm=1, traffic=None):
        """The contract's roofline object for the dominant kernel.  achieved = algorithmic bytes per launch / the average launch duration the
        HIP events measured; traffic = HBM bytes per launch from the PMC counters (live passes on a child invocation when `live`, else the
        recorded profile, else null).  The cooperative MuJoCo kernels are VALU / latency bound: `bound` says so, the
        HBM numbers are still reported, and their issue / flop counters live in the sidecar (scripts/bench_extras.py)."""
        algo = self.algorithmic_bytes_per_launch()
        achieved = algo / kernel_s / 1e9
        kernel = self.dominant_kernel()
        per = self.inner if self.env_id in MJ_COOP else 1  # dispatches of the dominant kernel per rollout launch
        if traffic is not None:  # (bytes per dispatch, provenance) measured by the caller: live_traffic_batch
            traffic, src = traffic
        else:
            traffic, src = live_traffic(self.env_id, self.N, self.inner, kernel, self.env_kwargs, warm) if live else (None, None)
        stale = False
        if traffic is not None:
            traffic *= per
        else:
            traffic, src = recorded_traffic(self.env_id, self.N, self.inner)
            stale = traffic is not None
        return {"bound": "valu" if self.env_id in MJ_COOP else "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src, **({"stale": True} if stale else {}), "algorithmic_bytes_per_launch": algo,
                "avg_kernel_ms": kernel_s * 1e3, "traffic_over_algorithmic": (traffic / algo) if traffic else None}

    def close(self):
        self.env.close()


T_START = time.perf_counter()


def fit_line(result, limit=LINE_LIMIT):
    """Serialise strictly (no NaN / Infinity) and keep the line under `limit` bytes by dropping optional detail, least important first."""
    for drop in (None, "secondary", "devices", "sustained", "clock_spinup"):
        if drop is not None:
            result.pop(drop, None)
        line = json.dumps(result, allow_nan=False, separators=(",", ":"))
        if len(line) < limit:
            return line
    raise AssertionError(f"bench line is {len(line)} bytes even without its optional parts")


def run_extras(args, full_path, primary):
    """BASELINE.json configs[2..4], the step() API, the opt-in configurations and the reference's own vectorisers: scripts/bench_extras.py in a
    child process (its own HIP context; a failure there cannot take the primary line with it).  Returns the headline dict for `secondary`."""
    script = os.path.join(ROOT, "scripts", "bench_extras.py")
    if not os.path.exists(script):
        return {"skipped": "scripts/bench_extras.py not present"}
    os.makedirs(os.path.dirname(full_path), exist_ok=True)
    tmp = full_path + ".primary.json"
    with open(tmp, "w") as f:
        json.dump(primary, f, allow_nan=False)
    cmd = [sys.executable, script, "--out", full_path, "--primary", tmp, "--pmc", args.pmc, "--budget", str(args.extras_budget),
           "--started", str(time.time() - (time.perf_counter() - T_START))]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    if args.no_api:
        cmd.append("--no-api")
    try:
        p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=args.extras_budget + 240)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not lines:
            return {"failed": f"exit {p.returncode}: {p.stderr.strip().splitlines()[-1][:200] if p.stderr.strip() else 'no output'}"}
        return json.loads(lines[-1])
    except subprocess.TimeoutExpired:
        return {"failed": f"timed out after {args.extras_budget + 240:.0f} s"}
    except Exception as e:  # the primary line must survive anything that happens here
        return {"failed": f"{type(e).__name__}: {e}"[:200]}
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


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
    ap.add_argument("--pmc", choices=["auto", "full", "off"], default="auto", help="live rocprofv3 counter passes on child invocations (off = recorded values only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle comparison of the first timed launch (the digest is still reported)")
    ap.add_argument("--no-api", action="store_true", help="extras: skip the per-launch step() API measurements")
    ap.add_argument("--no-secondary", "--no-extras", dest="no_secondary", action="store_true", help="skip scripts/bench_extras.py (BASELINE.json configs[2..4], API, opt-in lines)")
    ap.add_argument("--full-out", default=os.path.join(ROOT, "gpurun_out", "bench_full.json"), help="sidecar file with everything the line leaves out")
    ap.add_argument("--cpu-baseline-at-any-n", action="store_true", help="time the CPU leg on rank 0 also when WORLD_SIZE > 1 (default: at N = 1 only)")
    ap.add_argument("--extras-budget", type=float, default=210.0, help="extras: seconds after which no further optional measurement is started")
    ap.add_argument("--env-kwargs", default="{}", help='JSON constructor kwargs of the env, e.g. \'{"solver": "Newton"}\' (Humanoid: opt-in solver)')
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)  # profiled child: launches only, prints nothing
    ap.add_argument("--child-list", default=None, help=argparse.SUPPRESS)  # ... for several configurations one after the other: JSON [[env, N, inner, kwargs, warm], ...]
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

    if args.child and args.child_list:  # the profiled child of live_traffic_batch: each configuration's warm-up and three launches, nothing printed
        for c_env, c_n, c_inner, c_kw, c_warm in json.loads(args.child_list):
            c = make_config(c_env, int(c_n), int(c_inner), local_rank, rank, c_kw or None)
            for _ in range(int(c_warm) + 3):
                c.launch()
            sync_local()
            c.close()
        return 0
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
    # the steady state is 90.  --spinup 0 switches it off.
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
        return 0
    cfg.rearm()  # known start of the timed region: its first launch is the known-answer rollout (output_sha256 / verified)
    elapsed, kernel_s, st = cfg.timed(K, sync_all)
    from gymnasium_amd import distributed as gd

    red = gd.reduce_statistics(st, elapsed_s=elapsed, device=dev)  # the only collective: a few dozen bytes over RCCL/xGMI
    elapsed = red["elapsed_s"]
    env_steps, episodes, return_sum = float(red["env_steps"]), float(red["episodes"]), float(red["return_sum"])

    # ---- the first timed launch: digest on every rank, oracle comparison (the checker; never the thing measured) ----------------------
    traj = cfg.host_trajectory()
    digest = trajectory_digest(traj)
    verified = None
    if not args.no_verify:
        try:
            verified = oracle_check(cfg, traj)
        except Exception as e:
            verified = {"ok": None, "error": f"{type(e).__name__}: {e}"[:200]}
    del traj
    if verified is not None and verified.get("ok") is not None and args.env in MJ_IDS and gpu:
        try:  # ... and ten more launches from where the timed region left the robots (contacts, resets, TimeLimit), window by window
            later = mujoco_window_check(cfg)
            verified.update({"later_windows": later, "ok": bool(verified["ok"] and later["ok"]), "steps": verified["steps"] + later["steps"],
                             "max_abs_diff": max(verified["max_abs_diff"], later["max_abs_diff"])})
        except Exception as e:
            verified["later_windows"] = {"ok": None, "error": f"{type(e).__name__}: {e}"[:200]}
    # ---- sustained: the same launch back to back for >= args.sustained seconds (all ranks, same barrier discipline) ----------------
    sustained = None
    if args.sustained > 0 and elapsed >= 0.8:  # (elapsed is the all-reduced maximum: every rank takes the same branch)
        sustained = {"value": env_steps / elapsed, "launches": K, "seconds": elapsed, "avg_kernel_ms": kernel_s * 1e3, "note": "the timed region itself"}
    elif args.sustained > 0:
        Ks = max(K, int(args.sustained / max(kernel_s, 1e-7)) + 1)
        el_s, k_s, st_s = cfg.timed(Ks, sync_all)
        red_s = gd.reduce_statistics(st_s, elapsed_s=el_s, device=dev)
        sustained = {"value": float(red_s["env_steps"]) / red_s["elapsed_s"], "launches": Ks, "seconds": red_s["elapsed_s"], "avg_kernel_ms": k_s * 1e3}
    # What the collective itself proves about the job (n_gpus below is NOT read from the environment): gymnasium_amd/distributed.py census()
    cen = gd.census(rank, local_rank, device=dev, extra={"output_sha256": digest[:16], "verified": None if verified is None else verified["ok"]})
    ranks_seen = cen["ranks"]
    single = rank == 0 and world == 1 and gpu

    result = None
    if rank == 0:
        all_ok = [d.get("verified") for d in cen["devices"]]
        result = {
            "metric": METRIC, "value": env_steps / elapsed, "unit": "env-steps/s", "n_gpus": ranks_seen, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            **({} if gpu else {"engine": harness.label}),
            "config": {"workload": f"{args.env} num_envs={N} per GPU, random policy (on-device action_space.sample()), NEXT_STEP autoreset, TimeLimit, "
                                   f"fused rollout of {inner} vector steps per launch, trajectory (actions, obs, rewards, terminated, truncated) written to HBM",
                       "env": args.env, "env_kwargs": env_kwargs, "num_envs_per_gpu": N, "vector_steps_per_launch": inner,
                       "parallelism": f"env-sharded x{world} (no data-path collective)"},
            "roofline": cfg.roofline(kernel_s, live=single and args.pmc != "off"),
            "cpu_baseline": None,
            "output_sha256": digest, "verified": verified, "verified_all_ranks": (None if args.no_verify else (False if any(v is False for v in all_ok) else (True if all(v is True for v in all_ok) else None))),
            "rccl_ranks": ranks_seen, "world_size_env": world, "distinct_devices": cen["distinct_devices"],
            "devices": [{k: d.get(k) for k in ("rank", "uuid", "output_sha256", "verified")} for d in cen["devices"]],
            "episodes": episodes, "mean_episode_return": (return_sum / episodes) if episodes else None,
            "sustained_value": sustained["value"] if sustained else None, "sustained": sustained,
            "clock_spinup": {"seconds": args.spinup, "launches": spin_launches, "note": "untimed, before the warmup launches"},
        }
    cfg.close()

    # ---- CPU leg: the oracle port on rank 0, at N = 1 only (the contract; --cpu-baseline-at-any-n for the dry runs of the N > 1 flow) ------------------
    if rank == 0 and not args.no_cpu_baseline and (world == 1 or args.cpu_baseline_at_any_n):
        # the whole batch of one GPU on every host core (MuJoCo: a bounded sample of 64 sub-environments per core -- the oracle's per-env cost
        # does not depend on the batch size)
        n_cpu = min(N, 64 * usable_cpus()[0]) if args.env in MJ_COOP else N
        result["cpu_baseline"] = cpu_baseline(args.env, n_cpu, budget_s=args.cpu_budget)
    # ---- everything else (rank 0, one GPU): a child process, a sidecar file, headline values in the line ------------------------------------
    if single and not args.no_secondary:
        result["secondary"] = run_extras(args, args.full_out, result)
        result["full"] = os.path.relpath(args.full_out, ROOT)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(fit_line(result), flush=True)
        if (verified is not None and verified.get("ok") is False) or result.get("verified_all_ranks") is False:
            print("bench.py: the first timed launch of " + ("this rank" if (verified or {}).get("ok") is False else "another rank") +
                  " does NOT equal the oracle's trajectory -- the number above is not a measurement", file=sys.stderr)
            return 1
    return 0


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":  # one process of cpu_baseline's multi-core sample
        _, _, w_env, w_n, w_off, w_budget, w_start = sys.argv
        print(json.dumps(_oracle_rollouts(w_env, int(w_n), int(w_off), float(w_budget), float(w_start))))
    else:
        sys.exit(main())
