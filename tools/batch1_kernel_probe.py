"""Kernel time (HIP events, library side) of ONE QP through the device API for the kernel variants.
    python tools/batch1_kernel_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
from g1_locomotion_amd import BatchMPC, _lib, synth
x0, xr, ft, ct = synth.synthetic_batch(64, 10, seed=99, schedule="single")
dev = torch.device("cuda", 0)
for name, kid, maxs in (("compact", _lib.KERNEL_COMPACT, 2), ("wave", _lib.KERNEL_WAVE, 2), ("wrench", _lib.KERNEL_WRENCH, 4), ("split", _lib.KERNEL_SPLIT, 2)):
    eng = BatchMPC(horizon=10, kernel=kid, max_contacts_per_step=maxs, timing=True)
    ks, its = [], []
    u = torch.zeros((1, 10, 12), dtype=torch.float64, device=dev); xo = torch.zeros((1, 11, 13), dtype=torch.float64, device=dev)
    it = torch.zeros(1, dtype=torch.int32, device=dev)
    for i in range(400):
        b = i % 64
        d = [torch.from_numpy(v[b:b + 1]).to(dev) for v in (x0, xr, ft, ct)]
        eng.solve_device(1, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), u.data_ptr(), x_out=xo.data_ptr(), iters=it.data_ptr())
        eng.synchronize()
        ks.append(eng.last_kernel_ms() * 1e3); its.append(int(it.item()))
    ks = np.array(ks[50:]); its = np.array(its[50:])
    A = np.vstack([np.ones_like(its), its]).T.astype(float)
    c = np.linalg.lstsq(A, ks, rcond=None)[0]
    print(f"{name:8s} kernel={eng.kernel_name():22s} p50 {np.percentile(ks, 50):6.1f} us  min {ks.min():6.1f}  mean iters {its.mean():.1f}  fit: {c[0]:.1f} us + {c[1]:.3f} us/iter")
    eng.close()
