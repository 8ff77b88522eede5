"""How fast can this box run DEPENDENT kernels back to back?  A 65 536-element in-place add (nothing to compute) 1) launched eagerly from Python,
2) as 32 nodes of one HIP graph -- the floor under the per-launch step() API (bench.py api_step_device / api_step_graph)."""
import time

import torch

x = torch.zeros(65536, device="cuda")
for _ in range(50):
    x.add_(1.0)
torch.cuda.synchronize()
reps = 2000
t0 = time.perf_counter()
for _ in range(reps):
    x.add_(1.0)
torch.cuda.synchronize()
print("eager  in-place add, 65 536 floats: %.2f us per kernel" % ((time.perf_counter() - t0) / reps * 1e6))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(32):
        x.add_(1.0)
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    g.replay()
torch.cuda.synchronize()
print("graph  in-place add, 65 536 floats: %.2f us per kernel (32 nodes per graph)" % ((time.perf_counter() - t0) / 3200 * 1e6))
