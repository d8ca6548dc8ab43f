"""Batch-1 latency of the staged C-ABI call per contact pattern and kernel (GPU box).

    python tools/latency_patterns.py [calls=2000]

The reference's own call feeds full double support on every step (g1_mujoco_sim/src/run_simulation.py:100-101,106); the
synthetic single-support gait is what configs[1] names.  For each (pattern, kernel): p50 / p99 of srbdqp_solve_staged_f64(B = 1)
with the inputs already staged, and the mean ADMM iteration count."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]


def main():
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    import torch  # noqa: F401
    from g1_locomotion_amd import BatchMPC, _lib, synth
    out = {}
    kernels = [("auto", _lib.KERNEL_AUTO), ("compact", _lib.KERNEL_COMPACT), ("wrench", _lib.KERNEL_WRENCH)]
    for sched in ("single", "double", "mixed"):
        x0, xr, ft, ct = synth.synthetic_batch(64, 10, seed=99, schedule=sched)
        for kname, kid in kernels:
            eng = BatchMPC(horizon=10, kernel=kid)
            st = eng.stage()
            ts, its, sts = [], [], []
            for i in range(calls + 100):
                b = i % 64
                st["x0"][0] = x0[b]; st["x_ref"][0] = xr[b]; st["foot"][0] = ft[b]; st["contact"][0] = ct[b]
                t = time.perf_counter()
                eng.solve_staged(1, want_x=True)
                ts.append(time.perf_counter() - t)
                its.append(int(st["iters"][0])); sts.append(int(st["status"][0]))
            ts = np.array(ts[100:]) * 1e6
            out[f"{sched}/{kname}"] = {"kernel": eng.kernel_name(), "p50_us": round(float(np.percentile(ts, 50)), 2),
                                       "p99_us": round(float(np.percentile(ts, 99)), 2), "min_us": round(float(ts.min()), 2),
                                       "mean_iters": float(np.mean(its[100:])), "solved": float(np.mean(np.array(sts[100:]) == 1))}
            print(sched, kname, out[f"{sched}/{kname}"], flush=True)
            eng.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
