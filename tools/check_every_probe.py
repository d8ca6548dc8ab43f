"""Batch-1 latency of the staged C-ABI call over srbdqp_config.check_every (the termination test is made every
check_every iterations; the default is 5)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np, torch
import srbd_oracle as orc
from g1_locomotion_amd import BatchMPC
x0, xr, ft, ct = orc.synthetic_batch(64, 10, seed=99, schedule="single")
for ce in (2, 3, 4, 5, 6, 8, 10):
    with BatchMPC(horizon=10, check_every=ce) as eng:
        st = eng.stage()
        ts, its = [], []
        for i in range(2100):
            b = i % 64
            st["x0"][0] = x0[b]; st["x_ref"][0] = xr[b]; st["foot"][0] = ft[b]; st["contact"][0] = ct[b]
            t = time.perf_counter()
            eng.solve_staged(1, want_x=True)
            ts.append(time.perf_counter() - t); its.append(int(st["iters"][0]))
        ts = np.array(ts[100:]) * 1e6
        print("check_every %2d  p50 %.2f us  mean %.2f us  p99 %.1f  mean iters %.1f" % (ce, np.percentile(ts, 50), ts.mean(), np.percentile(ts, 99), np.mean(its[100:])))
