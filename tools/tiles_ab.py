"""fp32 calls of the general kernel: fp32 tiles (default for the QPs whose steps all have 0 or >= 3 stance contacts) against
fp64 tiles (SRBDQP_FLAG_F64_TILES) on one device-resident batch: throughput, convergence, and the force difference of
both to the fp64 solve.
    python tools/tiles_ab.py [N=20] [schedule=double] [B=65536]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
from g1_locomotion_amd import BatchMPC, _lib, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sched = sys.argv[2] if len(sys.argv) > 2 else "double"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
x0, xr, ft, ct = synth.synthetic_batch(B, N, seed=2026, schedule=sched)
dev = torch.device("cuda", 0)
res = {}
for name, f32, flags in (("f64", False, 0), ("f32/f64tiles", True, _lib.FLAG_F64_TILES), ("f32/f32tiles", True, _lib.FLAG_F32_TILES)):
    tdt = torch.float32 if f32 else torch.float64
    d = [torch.from_numpy(v).to(dev).to(tdt) if v.dtype == np.float64 else torch.from_numpy(v).to(dev) for v in (x0, xr, ft, ct)]
    u = torch.zeros((B, N, 12), dtype=tdt, device=dev)
    st = torch.zeros(B, dtype=torch.int32, device=dev)
    it = torch.zeros(B, dtype=torch.int32, device=dev)
    with BatchMPC(horizon=N, kernel=_lib.KERNEL_WRENCH, flags=flags) as eng:
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
    print(f"{name:13s} {kn} N={N} {sched} B={B}: {B / dt / 1e6:.3f} M QP/s  {dt * 1e3:.2f} ms  mean iters {itc.mean():.1f} max {itc.max()}"
          f"  solved {(stc == 1).mean():.5f}  status counts {dict(zip(*np.unique(stc, return_counts=True)))}", flush=True)
i64, s64, u64 = res["f64"]
for name in ("f32/f64tiles", "f32/f32tiles"):
    i32, s32, u32 = res[name]
    ok = (s32 == 1) & (s64 == 1)
    du = np.abs(u64 - u32).reshape(B, -1).max(1)
    print(f"{name}: |u - u_f64| over QPs solved by both: p50 {np.median(du[ok]):.2e} p99 {np.percentile(du[ok], 99):.2e} max {du[ok].max():.2e} N;"
          f" iteration count differs from f64 on {(i32[ok] != i64[ok]).mean():.4f} of them")
