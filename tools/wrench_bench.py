"""Throughput / convergence of the general kernel on device-resident batches.
    python tools/wrench_bench.py [N=20] [schedule=double] [B=65536] [rho=3] [eps=1e-6] [max_iter=250]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np, torch
import srbd_oracle as orc
from g1_locomotion_amd import BatchMPC, _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sched = sys.argv[2] if len(sys.argv) > 2 else "double"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
rho = float(sys.argv[4]) if len(sys.argv) > 4 else 3.0
eps = float(sys.argv[5]) if len(sys.argv) > 5 else 1e-6
max_iter = int(sys.argv[6]) if len(sys.argv) > 6 else 250
restart = int(sys.argv[7]) if len(sys.argv) > 7 else 0
x0, xr, ft, ct = orc.synthetic_batch(B, N, seed=2026, schedule=sched)
dev = torch.device("cuda", 0)
res = {}
for name, f32 in (("f64", False), ("f32", True)):
    tdt = torch.float32 if f32 else torch.float64
    d = [torch.from_numpy(v).to(dev).to(tdt) if v.dtype == np.float64 else torch.from_numpy(v).to(dev) for v in (x0, xr, ft, ct)]
    u = torch.zeros((B, N, 12), dtype=tdt, device=dev)
    st = torch.zeros(B, dtype=torch.int32, device=dev)
    it = torch.zeros(B, dtype=torch.int32, device=dev)
    with BatchMPC(horizon=N, kernel=_lib.KERNEL_WRENCH, rho=rho, eps_abs=eps, eps_rel=eps, max_iter=max_iter, rho_restart_iter=restart) as eng:
        def run():
            eng.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), u.data_ptr(), status=st.data_ptr(), iters=it.data_ptr(), f32=f32)
        run(); eng.synchronize()
        t = time.perf_counter(); K = 3
        for _ in range(K): run()
        eng.synchronize()
        dt = (time.perf_counter() - t) / K
        kn = eng.kernel_name()
    itc, stc = it.cpu().numpy(), st.cpu().numpy()
    res[name] = (itc, stc, u.cpu().numpy().astype(np.float64))
    print(f"{name} {kn} N={N} {sched} B={B} rho={rho} eps={eps} restart={restart} max_iter={max_iter}: {B / dt / 1e6:.3f} M QP/s  {dt * 1e3:.2f} ms  mean iters {itc.mean():.1f} p50 {np.median(itc):.0f} p99 {np.percentile(itc, 99):.0f}"
          f" max {itc.max()}  solved {(stc == 1).mean():.5f}  status counts {dict(zip(*np.unique(stc, return_counts=True)))}", flush=True)
i64, s64, u64 = res["f64"]; i32, s32, u32 = res["f32"]
bad = np.where(s32 != 1)[0]
print("f32 unsolved:", len(bad), " of which f64 unsolved:", int((s64[bad] != 1).sum()), " f64 iters of those (first 20):", i64[bad][:20])
ok = (s32 == 1) & (s64 == 1)
du = np.abs(u64 - u32).reshape(B, -1).max(1)
print("max |u32 - u64| over QPs solved by both: %.4f N; p99 %.4f N; over f32-unsolved: %.4f N" % (du[ok].max(), np.percentile(du[ok], 99), du[bad].max() if len(bad) else 0.0))
