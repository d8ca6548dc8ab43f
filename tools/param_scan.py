"""ADMM parameter scan on one device-resident batch: iterations, tail, accuracy vs the eps = 1e-10 solution, throughput.
    python tools/param_scan.py [N=10] [schedule=single] [B=4096] [f32=0]"""
import os, sys, time, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
from g1_locomotion_amd import BatchMPC, _lib, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
sched = sys.argv[2] if len(sys.argv) > 2 else "single"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
f32 = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
maxs = 2 if sched == "single" else 4
x0, xr, ft, ct = synth.synthetic_batch(B, N, seed=2026, schedule=sched)
dev = torch.device("cuda", 0)
def solve(f32=False, **kw):
    tdt = torch.float32 if f32 else torch.float64
    d = [torch.from_numpy(v).to(dev).to(tdt) if v.dtype == np.float64 else torch.from_numpy(v).to(dev) for v in (x0, xr, ft, ct)]
    u = torch.zeros((B, N, 12), dtype=tdt, device=dev); st = torch.zeros(B, dtype=torch.int32, device=dev); it = torch.zeros(B, dtype=torch.int32, device=dev)
    with BatchMPC(horizon=N, max_contacts_per_step=maxs, **kw) as eng:
        run = lambda: eng.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), u.data_ptr(), status=st.data_ptr(), iters=it.data_ptr(), f32=f32)
        run(); eng.synchronize()
        t = time.perf_counter()
        for _ in range(5): run()
        eng.synchronize()
        dt = (time.perf_counter() - t) / 5
    return u.cpu().numpy().astype(np.float64), st.cpu().numpy(), it.cpu().numpy(), dt
uex, stx, itx, _ = solve(eps_abs=1e-10, eps_rel=1e-10, max_iter=20000, rho_restart_iter=-1)
print(f"N={N} {sched} B={B} f32={f32}; exact: solved {np.mean(stx == 1):.4f} mean iters {itx.mean():.0f}")
print("  alpha  rho  chk | mean it  p50  p99  solved | err p50     p99     max (N)  | M QP/s")
for alpha, rho, chk in itertools.product((1.6, 1.7, 1.8), (0.7, 1.0, 1.4, 2.0), (5,)):
    u, st, it, dt = solve(f32=f32, alpha=alpha, rho=rho, check_every=chk, rho_restart_iter=-1)
    ok = (st == 1) & (stx == 1)
    e = np.abs(u - uex).reshape(B, -1).max(1)
    print(f"  {alpha:4.2f} {rho:5.2f} {chk:3d} | {it.mean():6.1f} {np.median(it):4.0f} {np.percentile(it, 99):4.0f} {np.mean(st == 1):.4f} | {np.median(e[ok]):.1e} {np.percentile(e[ok], 99):.1e} {e[ok].max():.1e} | {B / dt / 1e6:.2f}", flush=True)
