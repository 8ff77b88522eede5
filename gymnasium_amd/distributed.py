"""Multi-GPU layout of the vector-env path: sub-environments shard across ranks, nothing else does.

The reference has no distributed backend at all (AsyncVectorEnv is one OS process per env over pipes,
vector/async_vector_env.py:252-277).  Sub-environments never interact (vector/sync_vector_env.py:277-323 touches only
index i), so the N>1 path is: one process per GPU, rank r owns a contiguous block of global env indices, env g keeps
seed ``seed + g`` whatever the world size, and the ONLY collective is the final metric reduction (a few dozen bytes
over RCCL/xGMI; backend "nccl" on ROCm, "gloo" in the CPU tests).
"""
from __future__ import annotations

import os


def rank_info():
    """(rank, local_rank, world_size) from the torchrun environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(total_envs: int, rank: int, world: int):
    """Contiguous block [lo, hi) of global env indices owned by ``rank``.  Blocks differ by at most one
    env when ``world`` does not divide ``total_envs``."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of size {world}")
    base, rem = divmod(int(total_envs), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


STAT_KEYS = ("env_steps", "reset_steps", "episodes", "length_sum", "return_sum")


def reduce_statistics(stats: dict, elapsed_s: float | None = None, device=None) -> dict:
    """Sum the per-rank ``VectorEnv.statistics()`` over all ranks (one all_reduce) and take the MAX of ``elapsed_s``.

    Works on whatever backend the default process group uses; a no-op without an initialised group."""
    import torch
    import torch.distributed as dist

    out = {k: stats[k] for k in STAT_KEYS}
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        if elapsed_s is not None:
            out["elapsed_s"] = float(elapsed_s)
        return out
    # integers are exact in float64 up to 2^53 -- env-step counts of any realistic run
    t = torch.tensor([float(stats[k]) for k in STAT_KEYS], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    for k, v in zip(STAT_KEYS, t.tolist()):
        out[k] = v if k == "return_sum" else int(round(v))
    if elapsed_s is not None:
        e = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
        dist.all_reduce(e, op=dist.ReduceOp.MAX)
        out["elapsed_s"] = float(e[0])
    return out


def census(rank: int, local_rank: int, device=None, force_collective: bool = False, extra: dict | None = None) -> dict:
    """What the collective itself proves about the job: ``ranks`` = an all-reduce of ones (the ranks that really took part -- not WORLD_SIZE
    read from the environment) and ``devices`` = every rank's own device identity (name, UUID, PCI bus id), gathered: N distinct UUIDs = N
    different GPUs.  ``extra`` rides along in this rank's entry (bench.py: the digest of its first timed launch and the oracle's verdict on it).
    Without a process group (or at world size 1, unless ``force_collective``) nothing is communicated."""
    import torch
    import torch.distributed as dist

    on_gpu = device is not None and getattr(device, "type", "cpu") == "cuda"
    if on_gpu:
        props = torch.cuda.get_device_properties(local_rank)
        me = {"rank": rank, "local_rank": local_rank, "name": props.name, "uuid": str(getattr(props, "uuid", "")), "pci_bus_id": getattr(props, "pci_bus_id", None)}
    else:
        me = {"rank": rank, "local_rank": local_rank, "name": "cpu (dry run)", "uuid": f"cpu-{rank}"}
    if extra:
        me.update(extra)
    ranks, devices = 1, [me]
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_collective):
        ones = torch.ones(1, dtype=torch.int64, device=device)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks = int(ones.item())
        gathered = [None] * dist.get_world_size()
        dist.all_gather_object(gathered, me)
        devices = gathered
    return {"ranks": ranks, "devices": devices, "distinct_devices": len({d["uuid"] for d in devices})}
