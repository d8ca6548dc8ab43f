"""Why the receding-horizon warm start does not pay on this ADMM (round-4 verdict, item 5).  CPU oracle only.

Closed loops as in bench.py's closed_loop_cold / closed_loop_warm: segments of 25 consecutive control steps of one robot, every state the previous plan's prediction,
the contact schedule shifted by one step ("disturbed": each segment starts from a synthetic mixed-gait state and x_ref_hor stays where it was; "steady": double support,
the reference's own pattern).  Table 1: the shipped algorithm (oracle.update, check every 5 iterations, rho restart on) from different starting points.
Table 2: the mechanism -- the same ADMM with the convergence test evaluated on x (OSQP's iterate, as shipped) or on x~ (the linear-system solution, which does not
carry the relaxation lag x_k+1 = alpha x~ + (1 - alpha) x_k), a check at every iteration, with the distance of the returned forces to the exact optimum.

    python tools/warm_start_sweep.py [segments=40] > profiles/r05_warm_start_sweep.txt
"""
import os, sys
from dataclasses import replace
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import srbd_oracle as orc
from g1_locomotion_amd import synth

segs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
N, L = 10, 25
p = orc.default_params(N)


def shift(v, per, pad="dup"):
    v = v.reshape(N, per)
    o = np.empty_like(v)
    o[:-1] = v[1:]
    o[-1] = v[-1] if pad == "dup" else 0.0
    return o.reshape(-1)


VARIANTS = [
    ("cold (x = 0, y = 0)", lambda u, y: None),
    ("x, y shifted, last step duplicated  = MPC(warm_start=True) of rounds 1-4", lambda u, y: (shift(u, 12), shift(y, 20))),
    ("x, y shifted, last step zero", lambda u, y: (shift(u, 12, "zero"), shift(y, 20, "zero"))),
    ("x shifted only (y = 0)", lambda u, y: (shift(u, 12), np.zeros_like(y))),
    ("y shifted only (x = 0)", lambda u, y: (np.zeros_like(u), shift(y, 20))),
    ("x, y of the previous solve, not shifted", lambda u, y: (u.copy(), y.copy())),
    ("x, y shifted, scaled by 0.5", lambda u, y: (0.5 * shift(u, 12), 0.5 * shift(y, 20))),
]


def loop(X0, XR, FT, CT, solve):
    """solve(x, xr, ft, ct, prev) -> (u_hat, y, iters, status, qp); returns per-solve records of the solves 2..L of every segment"""
    rec = []
    for s in range(len(X0)):
        x, prev = X0[s].copy(), None
        for j in range(L):
            ct = np.roll(CT[s], -j, axis=0)
            uh, y, it, st, qp, extra = solve(x, XR[s], FT[s], ct, prev)
            if prev is not None:
                rec.append((it, st) + extra)
            prev = (uh, y)
            x = orc.rollout(qp, x, uh, p.force_scale)[1].copy()
    return rec


def table1(name, data):
    print(f"\n## 1. {name}: the shipped algorithm (oracle.update: check every {p.check_every} iterations, restart {p.rho_restart_iter} x {p.rho_restart_count}); iterations of the solves 2..{L}")
    for label, fn in VARIANTS:
        def solve(x, xr, ft, ct, prev):
            warm = None if prev is None else fn(*prev)
            r = orc.update(p, x, xr, ft, ct, warm=warm)
            pe = ()
            if warm is not None:        # how far the start is from the optimum found, in the variables and through P (what the dual residual sees)
                red, vi, ri = orc.presolve(r["qp"], ct)
                e0 = warm[0][vi] - r["u_hat"][vi]
                pe = (float(np.abs(e0).max() / np.abs(r["u_hat"]).max()), float(np.abs(red["P"] @ e0).max() / np.abs(red["q"]).max()))
            return r["u_hat"], r["y"], r["iters"], r["status"], r["qp"], pe
        rec = loop(*data, solve)
        its = np.array([r[0] for r in rec])
        s = f"{label:78s} mean {its.mean():6.2f}  p50 {np.percentile(its, 50):4.0f}  p99 {np.percentile(its, 99):4.0f}  at the cap {sum(r[1] == orc.STATUS_MAX_ITER for r in rec):3d}"
        if len(rec[0]) > 2:
            s += f"   |x0 - x*| / |x*| p50 {np.median([r[2] for r in rec]):.3f}   |P (x0 - x*)| / |q| p50 {np.median([r[3] for r in rec]):6.2f}"
        print(s, flush=True)


def admm(pp, P, q, A, l, u, x_init, y_init, on_xt):
    """admm_solve with the test at every iteration, on (x, P x) as shipped or on (x~, P x~)"""
    n, m = P.shape[0], A.shape[0]
    rho = orc.rho_vector(pp, l, u)
    Kinv = np.linalg.inv(P + pp.sigma * np.eye(n) + (A.T * rho) @ A)
    x = np.zeros(n) if x_init is None else x_init.copy()
    y = np.zeros(m) if y_init is None else y_init.copy()
    z = np.clip(A @ x, l, u); Px = P @ x; qn = np.abs(q).max()
    for k in range(1, pp.max_iter + 1):
        xt = Kinv @ (pp.sigma * x - q + A.T @ (rho * z - y)); zt = A @ xt
        Pxt = pp.sigma * (x - xt) - q - A.T @ (rho * (zt - z) + y)
        x = pp.alpha * xt + (1 - pp.alpha) * x; Px = pp.alpha * Pxt + (1 - pp.alpha) * Px
        zh = pp.alpha * zt + (1 - pp.alpha) * z; zn = np.clip(zh + y / rho, l, u); y = y + rho * (zh - zn); z = zn
        xe, Pxe = (xt, Pxt) if on_xt else (x, Px)
        Ax, Aty = A @ xe, A.T @ y
        if (np.abs(Ax - z).max() <= pp.eps_abs + pp.eps_rel * max(np.abs(Ax).max(), np.abs(z).max())
                and np.abs(Pxe + q + Aty).max() <= pp.eps_abs + pp.eps_rel * max(np.abs(Pxe).max(), np.abs(Aty).max(), qn)):
            return xe, y, k, orc.STATUS_SOLVED
    return xe, y, pp.max_iter, orc.STATUS_MAX_ITER


def table2(name, data, nseg):
    print(f"\n## 2. {name}: test on x (shipped) or on x~, check at every iteration, no restart; {nseg} segments; distance of the returned forces to the exact (active-set) optimum [N]")
    pp = replace(p, rho_restart_iter=0)
    data = tuple(v[:nseg] for v in data)
    for label, warm, on_xt in (("cold, test on x", 0, 0), ("warm (x, y shifted), test on x", 1, 0), ("cold, test on x~", 0, 1), ("warm (x, y shifted), test on x~", 1, 1)):
        cnt = [0]
        def solve(x, xr, ft, ct, prev):
            qp = orc.build_qp(pp, x, xr, ft, ct); red, vi, ri = orc.presolve(qp, ct)
            xi, yi = (None, None) if not (warm and prev is not None) else (shift(prev[0], 12)[vi], shift(prev[1], 20)[ri])
            xs, y, k, st = admm(pp, red["P"], red["q"], red["A"], red["l"], red["u"], xi, yi, on_xt)
            cnt[0] += 1
            err = ()
            if cnt[0] % 5 == 0:
                xe, _ = orc.solve_reference(pp, qp)
                err = (float(np.abs(xs - xe[vi]).max() * pp.force_scale),)
            uh = np.zeros(12 * N); uh[vi] = xs; yy = np.zeros(20 * N); yy[ri] = y
            return uh, yy, k, st, qp, err
        rec = loop(*data, solve)
        its = np.array([r[0] for r in rec]); errs = [r[2] for r in rec if len(r) > 2]
        print(f"{label:40s} iterations mean {its.mean():6.2f}  p99 {np.percentile(its, 99):4.0f}   |u - u_exact| p50 {np.median(errs):.1e}  max {np.max(errs):.1e} N", flush=True)


if __name__ == "__main__":
    print(f"# tools/warm_start_sweep.py {segs}: N = {N}, eps {p.eps_abs:g}, rho {p.rho:g} x {p.rho_fz_scale:g}, alpha {p.alpha:g}, sigma {p.sigma:g}; {segs} segments x {L - 1} warm-startable solves")
    dist = synth.synthetic_batch(segs, N, seed=77, schedule="mixed")
    steady = synth.synthetic_batch(segs, N, seed=78, schedule="double")
    table1("disturbed (bench.py closed_loop_*: synthetic mixed-gait states)", dist)
    table1("steady (full double support, the reference's own pattern, run_simulation.py:100-101)", steady)
    table2("disturbed", dist, max(2, segs // 5))
    table2("steady", steady, max(2, segs // 5))
