#!/usr/bin/env python3
"""bench.py -- env-steps/s of the MI355X vector-environment engine (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--env CartPole-v1] [--num-envs 65536] [--inner 128]

Workload (BASELINE.json configs[1]): CartPole-v1, num_envs = 65536 PER GPU, random policy.  One bench "step" is one
launch of the hot path over the whole batch: `rollout(inner)` = `inner` lockstep vector steps of all 65536
sub-environments with the random policy `action_space.sample()` evaluated on device (bit-identical to the host
policy) and the full trajectory (actions, observations, rewards, terminated, truncated) written to HBM.  Inputs are
resident in HBM when the timed region starts; nothing crosses PCIe inside it.

value = env-steps/s counted like the reference's benchmark_vector_step (gymnasium/utils/performance.py:88-90: NEXT_STEP
autoreset steps are not counted), whole job over all ranks.  Also reported: the per-launch step() API with device
tensors and the NumPy API (PCIe-inclusive) -- never as `value`.

N > 1: one process per GPU (torchrun), each rank owns its own 65536 sub-environments (global indices
rank*65536 ...; no data-path collective), one RCCL all-reduce of {env_steps, episodes, return_sum} at the end.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per env-step (DESIGN.md "Kernels"): fused rollout = what one env-step must write (+ amortised state)
ROLLOUT_BYTES = {"CartPole-v1": 34, "Pendulum-v1": 26, "Acrobot-v1": 42, "MountainCar-v0": 26, "MountainCarContinuous-v0": 22}
STATE_BYTES = {"CartPole-v1": 96, "Pendulum-v1": 64, "Acrobot-v1": 96, "MountainCar-v0": 64, "MountainCarContinuous-v0": 64}
STEP_BYTES = {"CartPole-v1": 108, "Pendulum-v1": 68, "Acrobot-v1": 116, "MountainCar-v0": 68, "MountainCarContinuous-v0": 64}
# MuJoCo family: which kernel dominates a rollout launch (Ant / Humanoid: `inner` x [mj_sample_kernel, mj_physics_kernel, mj_step_kernel];
# the cooperative physics kernel is > 99 % of the time -- profiles/r01_k_ant_coop.txt)
MJ_KERNEL = {"Ant-v5": "mj_physics_kernel", "Humanoid-v5": "mj_physics_kernel", "HumanoidStandup-v5": "mj_physics_kernel", "HalfCheetah-v5": "mj_physics_kernel"}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def cpu_baseline(env_id, num_envs, budget_s=12.0):
    """The CPU oracle (C restatement of the reference's CartPole + SyncVectorEnv semantics, 1 thread) on the same workload,
    bounded to ~budget_s of CPU time.  kind="port": the Python reference itself is not present on the GPU box."""
    import gymnasium_amd
    from gymnasium_amd import _native
    from oracle import oracle

    env = gymnasium_amd.make_vec(env_id, num_envs=num_envs, _engine_factory=oracle.engine_factory)
    env.reset(seed=0)
    env.action_space.seed(0)
    eng = env._engine
    eng.action_seed(_native.pcg_words(env.action_space.np_random))
    T = 4
    obs = np.zeros((T, num_envs) if (eng.obs_dtype is np.int64 and eng.obs_dim == 1) else (T, num_envs, eng.obs_dim), eng.obs_dtype)
    rew, te, tr = np.zeros((T, num_envs)), np.zeros((T, num_envs), np.bool_), np.zeros((T, num_envs), np.bool_)
    acts = np.zeros((T, num_envs) if eng.act_dtype is np.int64 else (T, num_envs, eng.act_dim), dtype=eng.act_dtype)
    t0 = time.perf_counter()
    eng.rollout(T, None, acts, obs, rew, te, tr)
    per_step = (time.perf_counter() - t0) / T
    reps = max(1, int(budget_s / (per_step * T)))
    eng.reset_stats()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.rollout(T, None, acts, obs, rew, te, tr)
    dt = time.perf_counter() - t0
    steps = eng.stats()["env_steps"]
    env.close()
    return {"value": steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{env_id} num_envs={num_envs}, {reps * T} vector steps ({steps} env-steps, {dt:.1f} s) of the C oracle's "
                      "rollout (same random policy, same outputs materialised), 1 thread"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--env", default="CartPole-v1")
    ap.add_argument("--num-envs", type=int, default=65536, help="sub-environments PER GPU")
    ap.add_argument("--inner", type=int, default=128, help="vector steps fused into one launch (one bench step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-api", action="store_true", help="skip the secondary per-launch step() API measurements")
    args = ap.parse_args()

    import torch

    import gymnasium_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    N, K, W, inner = args.num_envs, args.steps, args.warmup, args.inner

    env = gymnasium_amd.make_vec(args.env, num_envs=N, device=local_rank, output="torch", env_index_offset=rank * N)
    env.reset(seed=0)
    env.action_space.seed(rank)
    eng = env._engine

    # preallocated trajectory buffers, reused every step (a real collector would hand them to the learner)
    from gymnasium_amd import _native

    act_dtype = torch.int64 if env._discrete else torch.float32
    obs_dtype = {np.float64: torch.float64, np.float32: torch.float32, np.int64: torch.int64}[eng.obs_dtype]
    acts = torch.empty((inner, N) if env._discrete else (inner, N, eng.act_dim), dtype=act_dtype, device=dev)
    obs = torch.empty((inner, N) if (eng.obs_dtype is np.int64 and eng.obs_dim == 1) else (inner, N, eng.obs_dim), dtype=obs_dtype, device=dev)
    rew = torch.empty((inner, N), dtype=torch.float64, device=dev)
    te = torch.empty((inner, N), dtype=torch.bool, device=dev)
    tr = torch.empty((inner, N), dtype=torch.bool, device=dev)
    env._bind_stream()
    eng.action_seed(_native.pcg_words(env.action_space.np_random))

    def one_step():
        eng.rollout(inner, None, acts.data_ptr(), obs.data_ptr(), rew.data_ptr(), te.data_ptr(), tr.data_ptr())

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(W):
        one_step()
    sync_all()
    eng.reset_stats()
    # One HIP event on either side of the K launches, on the stream they are queued on (env._bind_stream() = torch's current stream):
    # the average launch duration is their distance / K.  (An event after every launch would put a marker packet between the kernels:
    # measured +9 us per 96 us launch, which rocprofv3's kernel trace does not see.)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sync_all()
    t0 = time.perf_counter()
    ev0.record()
    for k in range(K):
        one_step()
    ev1.record()
    sync_all()
    elapsed = time.perf_counter() - t0
    st = env.statistics()
    kernel_ms = [ev0.elapsed_time(ev1) / K]
    from gymnasium_amd import distributed as gd

    red = gd.reduce_statistics(st, elapsed_s=elapsed, device=dev)  # the only collective: a few dozen bytes over RCCL/xGMI
    elapsed = red["elapsed_s"]
    env_steps, episodes, return_sum = float(red["env_steps"]), float(red["episodes"]), float(red["return_sum"])

    result = None
    if rank == 0:
        value = env_steps / elapsed
        avg_kernel_s = float(np.mean(kernel_ms)) * 1e-3
        if args.env in ROLLOUT_BYTES:
            rollout_b, state_b = ROLLOUT_BYTES[args.env], STATE_BYTES[args.env]
        else:  # MuJoCo family: float32 action row + float64 obs row + reward + 2 flags per env-step; state row R+W per launch
            rollout_b = (8 if env._discrete else 4 * eng.act_dim) + 8 * eng.obs_dim + 8 + 2
            state_b = 2 * (8 * eng.state_dim + 4 + 8 + 4)
        bytes_per_launch = (rollout_b * inner + state_b) * N
        achieved = bytes_per_launch / avg_kernel_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(f"{args.env}:{N}:{inner}")
            except Exception:
                traffic = None
        result = {
            "metric": "env-steps/sec at num_envs=65536 (1/2/4/8 MI355X) vs CPU AsyncVectorEnv",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.env} num_envs={N} per GPU, random policy (on-device action_space.sample()), "
                                   f"NEXT_STEP autoreset, TimeLimit, fused rollout of {inner} vector steps per launch, "
                                   "trajectory (actions, obs, rewards, terminated, truncated) written to HBM",
                       "env": args.env, "num_envs_per_gpu": N, "vector_steps_per_launch": inner,
                       "parallelism": f"env-sharded x{world} (no data-path collective)"},
            "episodes": episodes, "mean_episode_return": (return_sum / episodes) if episodes else None,
            "roofline": {"bound": "hbm", "kernel": "rollout_kernel" if args.env in ROLLOUT_BYTES else ("tab_rollout_kernel" if eng.obs_dtype is np.int64 else MJ_KERNEL.get(args.env, "mj_rollout_kernel")), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": bytes_per_launch, "avg_kernel_ms": avg_kernel_s * 1e3},
        }

    # ---- secondary numbers (rank 0, N=1 only): per-launch step() API ---------------------------------------
    if rank == 0 and world == 1 and not args.no_api:
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
        env_np = gymnasium_amd.make_vec(args.env, num_envs=N, device=local_rank, copy=False)
        env_np.reset(seed=0)
        env_np.action_space.seed(0)
        for _ in range(5):
            env_np.step(env_np.action_space.sample())
        t0 = time.perf_counter()
        reps = 100
        for _ in range(reps):
            env_np.step(env_np.action_space.sample())
        dt = time.perf_counter() - t0
        result["api_step_numpy"] = {"value": N * reps / dt, "unit": "vector-env lanes/s (NumPy in/out over PCIe, host action sampling)",
                                    "us_per_step_wall": dt / reps * 1e6}
        env_np.close()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.env, N)
    elif rank == 0:
        result["cpu_baseline"] = None

    env.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
