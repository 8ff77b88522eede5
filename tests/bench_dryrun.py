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
    """bench.Config with its four device-specific methods on the CPU checker: NumPy buffers, synchronous launches (the wall clock is the kernel clock)."""

    def make_env(self):
        import gymnasium_amd
        from oracle import oracle

        return gymnasium_amd.make_vec(self.env_id, num_envs=self.N, env_index_offset=self.rank * self.N, _engine_factory=oracle.engine_factory,
                                      **(self.env_kwargs or {}))

    def alloc_trajectory(self):
        env, eng, T, N = self.env, self.eng, self.inner, self.N
        return (np.zeros((T, N) if env._discrete else (T, N, eng.act_dim), dtype=eng.act_dtype),
                np.zeros((T, N) if (eng.obs_dtype is np.int64 and eng.obs_dim == 1) else (T, N, eng.obs_dim), eng.obs_dtype),
                np.zeros((T, N)), np.zeros((T, N), np.bool_), np.zeros((T, N), np.bool_)), None

    def host_trajectory(self):
        return tuple(b.copy() for b in self.first[0])

    def launch(self, bufs=None):
        b = (bufs or self.rest)[0]
        self.eng.rollout(self.inner, None, b[0], b[1], b[2], b[3], b[4])

    def timed(self, K, sync):
        sync()
        self.eng.reset_stats()
        sync()
        t0 = time.perf_counter()
        self.launch(self.first)
        for _ in range(K - 1):
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
