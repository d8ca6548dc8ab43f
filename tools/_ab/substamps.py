import os, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/tools") else os.getcwd()
sys.path.insert(0, ROOT)
import ctypes as C, numpy as np, torch
from g1_locomotion_amd import BatchMPC, _lib, synth
N = int(sys.argv[1]); sched = sys.argv[2]; B = int(sys.argv[3]); f32 = bool(int(sys.argv[4]))
x0, xr, ft, ct = synth.synthetic_batch(B, N, 2026, sched)
dev = torch.device("cuda", 0); tdt = torch.float32 if f32 else torch.float64
d = [torch.from_numpy(v).to(dev).to(tdt) if v.dtype == np.float64 else torch.from_numpy(v).to(dev) for v in (x0, xr, ft, ct)]
u = torch.empty((B, N, 12), dtype=tdt, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev)
st = torch.zeros((B, 16), dtype=torch.int64, device=dev)
eng = BatchMPC(horizon=N, kernel=_lib.KERNEL_WRENCH, rho_restart_iter=-1)
eng._lib.srbdqp_set_stamp_buffer(eng._h, C.c_void_p(st.data_ptr()))
for _ in range(2):
    eng.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), u.data_ptr(), iters=it.data_ptr(), f32=f32); eng.synchronize()
s = st.cpu().numpy().astype(np.float64)
seq = [0, 10, 11, 12, 13, 14, 1]
names = ["global loads -> LDS", "sincos + bookkeeping", "CP prefix + J", "eh + T1/T2", "gt_tables(eh)", "MT + barrier"]
print(f"N={N} {sched} B={B} f32={f32} kernel={eng.kernel_name()}")
for i, nm in enumerate(names):
    dl = s[:, seq[i + 1]] - s[:, seq[i]]
    print(f"  {nm:22s} mean {dl.mean():8.0f} median {np.median(dl):8.0f}")
print("  total load+tables", (s[:, 1] - s[:, 0]).mean())
