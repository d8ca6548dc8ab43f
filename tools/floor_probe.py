import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np, torch
import srbd_oracle as orc
from g1_locomotion_amd import BatchMPC, _lib
x0, xr, ft, ct = orc.synthetic_batch(64, 10, seed=99, schedule="single")
for name, kw, zero in (("flight_spin", {}, True), ("flight_sync", {"flags": _lib.FLAG_NO_SPIN}, True), ("maxiter5", {"max_iter": 5}, False), ("normal", {}, False)):
    eng = BatchMPC(horizon=10, **kw)
    st = eng.stage()
    ts = []
    for i in range(2100):
        b = i % 64
        st["x0"][0] = x0[b]; st["x_ref"][0] = xr[b]; st["foot"][0] = ft[b]; st["contact"][0] = 0 if zero else ct[b]
        t = time.perf_counter()
        eng.solve_staged(1, want_x=True)
        ts.append(time.perf_counter() - t)
    ts = np.array(ts[100:]) * 1e6
    print(name, "p50 %.2f min %.2f p99 %.2f" % (np.percentile(ts, 50), ts.min(), np.percentile(ts, 99)))
    eng.close()
