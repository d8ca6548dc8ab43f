"""Robustness sweep of the engine's default routing: every horizon x schedule x precision on random batches (B = 2048, three
seeds): finite outputs, statuses in {SOLVED, MAX_ITER}, solved share, friction cone / normal-force bounds of the solved QPs.
    python tools/fuzz_general.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
import torch  # noqa: F401
from g1_locomotion_amd import BatchMPC, synth
bad = 0
for N in (4, 8, 10, 12, 16, 20, 24):
    for sched in ("single", "double", "mixed"):
        for f32 in (False, True):
            tot, solved, worst = 0, 0, 0.0
            names = set()
            for seed in (1, 2, 3):
                B = 2048
                x0, xr, ft, ct = synth.synthetic_batch(B, N, seed=1000 * N + seed, schedule=sched)
                if seed == 3:                      # some flight phases and one-foot point contacts
                    rng = np.random.default_rng(seed)
                    ct[rng.random(ct.shape[:2]) < 0.05] = 0
                    ct[rng.random(ct.shape) < 0.03] = 0
                with BatchMPC(horizon=N) as eng:
                    out = eng.solve(x0, xr, ft, ct, dtype=np.float32 if f32 else np.float64)
                    names.add(eng.kernel_name())
                u, st = out["u"].astype(np.float64), out["status"]
                ok = st == 1
                if not np.isfinite(u).all() or not np.isfinite(out["x"]).all() or not np.isin(st, (1, 2)).all():
                    bad += 1
                    print("  !! N=%d %s f32=%d seed=%d: finite %s statuses %s" % (N, sched, f32, seed, np.isfinite(u).all(), dict(zip(*np.unique(st, return_counts=True)))))
                f = u.reshape(B, N, 4, 3)[ok]
                c = ct[ok].astype(bool)
                fz = f[..., 2]
                # ADMM's own primal tolerance: eps_abs + eps_rel max(|Ax|, |z|) in the scaled variables = (1e-6 + 1e-6 * 10) * 100 N
                # at the 1000 N bound (fp32: eps floor 2e-6)
                tol = 2.5e-3 if f32 else 1.2e-3
                viol = max(float((np.abs(f[..., 0]) - 0.8 * fz).max()), float((np.abs(f[..., 1]) - 0.8 * fz).max()),
                           float((10.0 - fz[c]).max()) if c.any() else 0.0, float(np.abs(f[~c]).max()) if (~c).any() else 0.0)
                worst = max(worst, viol)
                if viol > tol:
                    bad += 1
                    print("  !! N=%d %s f32=%d seed=%d: constraint violation %.3e N" % (N, sched, f32, seed, viol))
                tot += B; solved += int(ok.sum())
            print("N=%2d %-6s %s %-22s solved %.4f  worst constraint violation %.2e N" % (N, sched, "f32" if f32 else "f64", ",".join(sorted(names)), solved / tot, worst), flush=True)
print("FAILURES:", bad)
