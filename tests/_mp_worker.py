"""World-size-N worker for tests/test_multiprocess.py (launched with torch.distributed.run, gloo backend, CPU).

Each rank owns a contiguous shard of the global batch (gymnasium_amd.distributed.shard_range), backed by the CPU oracle
through the product's host class (the GPU is absent in this container); rank 0 additionally runs the whole batch in one
process and checks that (a) the gathered shard trajectories equal the single-process ones bit for bit and (b) the
all-reduced statistics equal the single-process totals.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist

    import gymnasium_amd
    from gymnasium_amd import distributed as gd
    from oracle import oracle

    env_id, total, T, out_path = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo")
    rank, _, world = gd.rank_info()
    lo, hi = gd.shard_range(total, rank, world)
    env = gymnasium_amd.make_vec(env_id, num_envs=hi - lo, env_index_offset=lo, _engine_factory=oracle.engine_factory)
    obs, _ = env.reset(seed=123)
    # the global random policy: every rank draws the same global action batch and keeps its slice
    full = gymnasium_amd.make_vec(env_id, num_envs=total, _engine_factory=oracle.engine_factory) if rank == 0 else None
    policy = gymnasium_amd.make_vec(env_id, num_envs=total, _engine_factory=oracle.engine_factory).action_space
    policy.seed(5)
    traj = [obs.copy()]
    ref = []
    if rank == 0:
        o, _ = full.reset(seed=123)
        ref.append(o.copy())
    for _ in range(T):
        a = policy.sample()
        o, r, te, tr, _ = env.step(a[lo:hi])
        traj.append(o.copy())
        if rank == 0:
            o2, _, _, _, _ = full.step(a)
            ref.append(o2.copy())
    mine = torch.from_numpy(np.stack(traj))  # [T+1, n_local, obs_dim]
    sizes = [gd.shard_range(total, r, world) for r in range(world)]
    width = max(h - l for l, h in sizes)  # gloo's all_gather wants equal shapes: pad the short shards
    padded = torch.zeros((T + 1, width, mine.shape[2]), dtype=mine.dtype)
    padded[:, : hi - lo] = mine
    gathered = [torch.empty_like(padded) for _ in sizes]
    dist.all_gather(gathered, padded)
    gathered = [g[:, : h - l] for g, (l, h) in zip(gathered, sizes)]
    red = gd.reduce_statistics(env.statistics(), elapsed_s=1.0 + rank)
    if rank == 0:
        whole = torch.cat(gathered, dim=1).numpy()
        ok_traj = bool(np.array_equal(whole, np.stack(ref)))
        st = full.statistics()
        ok_stats = all(red[k] == st[k] for k in ("env_steps", "reset_steps", "episodes", "length_sum")) and \
            abs(red["return_sum"] - st["return_sum"]) <= 1e-9 * max(1.0, abs(st["return_sum"]))
        json.dump({"ok_traj": ok_traj, "ok_stats": ok_stats, "world": world, "elapsed_max": red["elapsed_s"],
                   "env_steps": red["env_steps"], "episodes": red["episodes"]}, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
