"""TEST INFRASTRUCTURE: bench.py's control flow on the CPU checker, for tests/test_bench_multirank.py.

Launched exactly like the driver launches bench.py (`python -m torch.distributed.run --nproc-per-node N tests/bench_dryrun.py --gpus N ...`), this
imports bench.main and hands it a harness that (a) replaces bench.Config by a subclass whose engine is the oracle behind the same host class
(NumPy buffers, synchronous launches: the wall clock is the kernel clock) and (b) selects the gloo process group.  What executes is everything of
the N > 1 path that no GPU-less box could otherwise run: the pilot's MAX all-reduce, the barrier brackets, both `sustained` branches, the
statistics all-reduce, the rank / device census, rank 0's CPU legs and the single JSON line.  The line it prints is NOT a measurement and says so
(`engine`).  bench.py itself has no flag that reaches this: a product script must not be able to print the metric line from the checker."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class OracleConfig(bench.Config):
    def __init__(self, env_id, N, inner, local_rank, rank, env_kwargs=None):
        import torch

        import gymnasium_amd
        from gymnasium_amd import _native
        from oracle import oracle

        self.torch, self.env_id, self.N, self.inner, self.env_kwargs = torch, env_id, N, inner, env_kwargs
        env = gymnasium_amd.make_vec(env_id, num_envs=N, env_index_offset=rank * N, _engine_factory=oracle.engine_factory, **(env_kwargs or {}))
        env.reset(seed=0)
        env.action_space.seed(rank)
        eng = env._engine
        self.env, self.eng = env, eng
        self.acts = np.zeros((inner, N) if env._discrete else (inner, N, eng.act_dim), dtype=eng.act_dtype)
        self.obs = np.zeros((inner, N) if (eng.obs_dtype is np.int64 and eng.obs_dim == 1) else (inner, N, eng.obs_dim), eng.obs_dtype)
        self.rew, self.te, self.tr = np.zeros((inner, N)), np.zeros((inner, N), np.bool_), np.zeros((inner, N), np.bool_)
        eng.action_seed(_native.pcg_words(env.action_space.np_random))

    def launch(self):
        self.eng.rollout(self.inner, None, self.acts, self.obs, self.rew, self.te, self.tr)

    def timed(self, K, sync):
        sync()
        self.eng.reset_stats()
        sync()
        t0 = time.perf_counter()
        for _ in range(K):
            self.launch()
        sync()
        elapsed = time.perf_counter() - t0
        return elapsed, elapsed / K, self.env.statistics()


class Harness:
    backend = "gloo"
    config_cls = OracleConfig
    label = "oracle (CPU checker: a dry run of the control flow, NOT a measurement)"


if __name__ == "__main__":
    bench.main(harness=Harness())
