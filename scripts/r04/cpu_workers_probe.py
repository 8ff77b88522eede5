"""How many worker processes give the C oracle its best aggregate rate on this host?  (bench.py cpu_baseline; the GPU box reports 256 logical CPUs)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except Exception as e:
        print(f, "unreadable")
os.system("lscpu | egrep 'Model name|Socket|Core|Thread|^CPU\\(s\\)' ")
for w in (32, 64, 96, 128, 192, 256):
    if w <= (os.cpu_count() or 1):
        r = bench.cpu_baseline("CartPole-v1", 65536, budget_s=2.0, workers=w)
        print(w, "workers: %.1f M env-steps/s, per worker %.2f M" % (r["value"] / 1e6, r["per_core_value"] / 1e6), flush=True)
