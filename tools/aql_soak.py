"""Soak of the staged one-QP call through the AQL queue + completion records: `calls` solves cycling over 9 different QPs (single / double / mixed support, a flight
phase and two that restart among them); every result must be bit-equal to the first result of its QP.   python tools/aql_soak.py [calls=300000]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch  # noqa: F401  (its HIP runtime first)
from g1_locomotion_amd import BatchMPC, synth

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
N = 10
qps = []
for sched, seed in (("single", 1), ("double", 2), ("mixed", 3), ("double", 4), ("single", 5), ("mixed", 6)):
    x0, xr, ft, ct = synth.synthetic_batch(1, N, seed, sched)
    qps.append((x0[0], xr[0], ft[0].reshape(N, 12), ct[0].reshape(N, 4)))
qps.append((qps[0][0], qps[0][1], qps[0][2], np.zeros((N, 4), np.uint8)))     # flight phase
with BatchMPC(horizon=N) as eng:                                              # two QPs that pass a restart mark (their further passes queue behind the first)
    for sched in ("double", "single"):
        x0, xr, ft, ct = synth.synthetic_batch(4096, N, 515, sched)
        r = eng.solve(x0, xr, ft, ct)
        j = int(np.argsort(-r["iters"])[3])
        assert r["iters"][j] > 55
        qps.append((x0[j], xr[j], ft[j].reshape(N, 12), ct[j].reshape(N, 4)))
with BatchMPC(horizon=N) as eng:
    st = eng.stage()
    first = [None] * len(qps)
    bad = 0
    t0 = time.perf_counter()
    for i in range(calls):
        q = (i * 5 + i // 11) % len(qps)
        x0, xr, ft, ct = qps[q]
        st["x0"][0] = x0; st["x_ref"][0] = xr; st["foot"][0] = ft; st["contact"][0] = ct
        eng.solve_staged(1, want_x=True)
        r = (st["u"][0].copy(), st["x"][0].copy(), int(st["status"][0]), int(st["iters"][0]))
        if first[q] is None:
            first[q] = r
        elif not (r[2] == first[q][2] and r[3] == first[q][3] and np.array_equal(r[0], first[q][0]) and np.array_equal(r[1], first[q][1])):
            bad += 1
            if bad < 5:
                print("MISMATCH at call", i, "QP", q, r[2], r[3], first[q][2], first[q][3], float(np.abs(r[0] - first[q][0]).max()))
    el = time.perf_counter() - t0
    print("launch path:", eng.batch1_launch_path(), " calls:", calls, " mismatches:", bad, " %.1f us per call incl. Python" % (el / calls * 1e6),
          " statuses / iterations of the 9 QPs:", [(f[2], f[3]) for f in first])
    sys.exit(1 if bad else 0)
