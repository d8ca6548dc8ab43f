#!/usr/bin/env python3
"""Fuzz of the one-wave kernel's rho restart in place against the C restatement (oracle/srbd_oracle.c) run the same way: seeds x horizons x (period, count).
    python tools/restart_fuzz.py [QPs per case = 8192]
Per case: share of QPs with the same status / the same iteration count, the largest iteration difference, the largest force difference over the QPs with the same
count, solved share with and without the restart.  (Test infrastructure: the oracle is the checker here.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
import torch  # noqa: F401  (HIP runtime first)
import srbd_oracle as orc
import c_oracle
from g1_locomotion_amd import BatchMPC

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rng = np.random.default_rng(11)
cases = []
for seed in (3, 17, 29, 41):
    for N in (10, 8, 4):
        cases.append((seed, N, 55, 2))
        cases.append((seed, N, int(rng.choice([30, 40, 45, 60, 75, 90])), int(rng.integers(1, 5))))
worst_it, worst_u, t0 = 0, 0.0, time.time()
print("# one-wave kernel, restart in place, against oracle/srbd_oracle.c (same rule): %d QPs per case, single-support gait" % B)
print("# seed  N  every x count | same status  same iters  max |d iters|  max |d u| N (same iters) | solved: kernel  oracle  fixed rho")
for seed, N, every, count in cases:
    x0, xr, ft, ct = orc.synthetic_batch(B, N, seed=seed, schedule="single")
    p = orc.params_for(N, rho_restart_iter=every, rho_restart_count=count)
    ref = c_oracle.solve_batch(p, x0, xr, ft, ct, nthreads=16)
    plain = c_oracle.solve_batch(orc.params_for(N), x0, xr, ft, ct, nthreads=16)
    with BatchMPC(horizon=N, max_contacts_per_step=2, rho_restart_iter=every, rho_restart_count=count) as eng:
        out = eng.solve(x0, xr, ft, ct)
        assert eng.kernel_name().startswith("wave_"), eng.kernel_name()
    same_st = out["status"] == ref["status"]
    dit = np.abs(out["iters"].astype(int) - ref["iters"].astype(int))
    same = dit == 0
    du = np.abs(out["u"] - ref["u"]).reshape(B, -1).max(1)
    worst_it = max(worst_it, int(dit.max())); worst_u = max(worst_u, float(du[same].max()))
    print("%5d %3d  %3d x %d      |   %.5f     %.5f      %3d        %.2e             |   %.5f   %.5f   %.5f"
          % (seed, N, every, count, same_st.mean(), same.mean(), dit.max(), du[same].max(), (out["status"] == 1).mean(), (ref["status"] == 1).mean(), (plain["status"] == 1).mean()))
print("# worst over %d cases: |d iters| %d, |d u| over the QPs with the same count %.2e N   (%.0f s)" % (len(cases), worst_it, worst_u, time.time() - t0))
