"""Diagnostic: host CPU share of the GPU box and the C oracle's thread scaling."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try: print(f, open(f).read().strip())
    except OSError as e: print(f, "absent")
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
import srbd_oracle as orc, c_oracle
x0, xr, ft, ct = orc.synthetic_batch(4096, 10, seed=1000, schedule="single")
p = orc.SrbdParams()
for nt in (1, 8, 16, 32, 64, 128, 256):
    B = 512 if nt == 1 else 4096
    c_oracle.solve_batch(p, x0[:B], xr[:B], ft[:B], ct[:B], nthreads=nt)
    t = time.perf_counter(); reps = 3
    for _ in range(reps): c_oracle.solve_batch(p, x0[:B], xr[:B], ft[:B], ct[:B], nthreads=nt)
    dt = (time.perf_counter() - t) / reps
    print(nt, "threads:", round(B / dt), "QP/s")
