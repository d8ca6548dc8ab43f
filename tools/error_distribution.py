"""Distribution of |u_gpu - u_exact| over a seeded batch (max over a QP's forces, newtons): the solver-independent anchor of
the parity claim, reported instead of only asserted.  u_exact = ADMM at eps 1e-10 (oracle/srbd_oracle.c, 20000 iterations),
cross-checked on a sample against the active-set KKT solve of oracle.solve_reference().
    python tools/error_distribution.py [config=1|2] [B]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np, torch
import srbd_oracle as orc, c_oracle
from g1_locomotion_amd import BatchMPC
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N, sched, f32 = (10, "single", False) if cfg == 1 else (20, "double", True)
B = int(sys.argv[2]) if len(sys.argv) > 2 else (4096 if cfg == 1 else 1024)
x0, xr, ft, ct = orc.synthetic_batch(B, N, seed=1000 * cfg, schedule=sched)
xin = [v.astype(np.float32).astype(np.float64) for v in (x0, xr, ft)] if f32 else [x0, xr, ft]
with BatchMPC(horizon=N, max_contacts_per_step=2 if cfg == 1 else 4) as eng:
    out = eng.solve(x0, xr, ft, ct, dtype=np.float32 if f32 else np.float64)
    kn = eng.kernel_name()
p = orc.params_for(N)
tight = c_oracle.solve_batch(orc.SrbdParams(rho=p.rho, eps_abs=1e-10, eps_rel=1e-10, max_iter=20000, check_every=25), *xin, ct, nthreads=16)
err = np.abs(out["u"].astype(np.float64) - tight["u"]).reshape(B, -1).max(1)
ok = out["status"] == 1
chk = []
for b in np.random.default_rng(0).choice(B, 12, replace=False):
    xs, _ = orc.solve_reference(p, orc.build_qp(p, xin[0][b], xin[1][b], xin[2][b], ct[b]))
    chk.append(float(np.abs(tight["u"][b].reshape(-1) - xs * p.force_scale).max()))
q = lambda v, a: float(np.percentile(v, a)) if len(v) else None
res = {"config": cfg, "kernel": kn, "B": B, "horizon": N, "schedule": sched, "dtype": "f32" if f32 else "f64",
       "solved_frac": float(ok.mean()), "mean_iters": float(out["iters"].mean()),
       "err_N_solved": {"p50": q(err[ok], 50), "p90": q(err[ok], 90), "p99": q(err[ok], 99), "p99.9": q(err[ok], 99.9), "max": float(err[ok].max())},
       "err_N_at_cap": {"count": int((~ok).sum()), "p50": q(err[~ok], 50), "max": float(err[~ok].max()) if (~ok).any() else None},
       "tight_reference_vs_active_set_kkt_max_N": max(chk), "tight_reference_solved_frac": float((tight["status"] == 1).mean()),
       "max_force_N": float(np.abs(tight["u"]).max())}
print(json.dumps(res, indent=1))
