"""Diagnostic: per-phase cycles of the general kernel from its in-kernel s_memtime stamps (shader cycles).
    python tools/wrench_stamps.py [N=20] [schedule=double] [B=65536] [f32=1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
from g1_locomotion_amd import BatchMPC, _lib, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sched = sys.argv[2] if len(sys.argv) > 2 else "double"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
f32 = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
names = ["load+tables", "E / V / Bd", "T assembly", "F (chol)", "W", "I", "fragments + x_q", "ADMM", "outputs + rollout"]
for b in (1, B):
    x0, xr, ft, ct = synth.synthetic_batch(b, N, 2026, sched)
    dev = torch.device("cuda", 0)
    tdt = torch.float32 if f32 else torch.float64
    d = [torch.from_numpy(v).to(dev).to(tdt) if v.dtype == np.float64 else torch.from_numpy(v).to(dev) for v in (x0, xr, ft, ct)]
    u = torch.empty((b, N, 12), dtype=tdt, device=dev)
    xo = torch.empty((b, N + 1, 13), dtype=tdt, device=dev)
    it = torch.empty(b, dtype=torch.int32, device=dev)
    st = torch.zeros((b, 16), dtype=torch.int64, device=dev)
    eng = BatchMPC(horizon=N, kernel=_lib.KERNEL_WRENCH, rho_restart_iter=-1)
    eng._lib.srbdqp_set_stamp_buffer(eng._h, C.c_void_p(st.data_ptr()))
    for _ in range(2):
        eng.solve_device(b, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), u.data_ptr(), x_out=xo.data_ptr(), iters=it.data_ptr(), f32=f32)
        eng.synchronize()
    s = st.cpu().numpy().astype(np.float64)
    its = it.cpu().numpy()
    dl = np.diff(s[:, :10], axis=1)
    print(f"B={b} kernel={eng.kernel_name()} {sched} mean iters {its.mean():.1f}")
    for i, nm in enumerate(names):
        print(f"  {nm:18s} mean {dl[:, i].mean():9.0f} cyc   median {np.median(dl[:, i]):9.0f}")
    if s[:, 10].any():   # inside "fragments + x_q": 6 -> 10 late V / Bd + half rows, 10 -> 11 x_q, 11 -> 12 G'(G x_q), 12 -> 7 refinement step
        sub = [s[:, 10] - s[:, 6], s[:, 11] - s[:, 10]] + ([s[:, 12] - s[:, 11], s[:, 7] - s[:, 12]] if s[:, 12].any() else [s[:, 7] - s[:, 11]])
        print("    of which late V / Bd + half rows, x_q" + (", G'(G x_q), refinement step" if s[:, 12].any() else ", rest") + ": " + " ".join(f"{v.mean():.0f}" for v in sub))
    tot = s[:, 9] - s[:, 0]
    print(f"  total              mean {tot.mean():9.0f} cyc   per ADMM iteration {np.mean(dl[:, 7] / np.maximum(its, 1)):.0f} cyc   set-up share {1 - dl[:, 7].mean() / tot.mean():.2f}")
    eng.close()
