"""Diagnostic: per-phase cycles of the 4-wave kernel on the STAGED batch-1 path (inputs / outputs in GPU-mapped host
memory), to set beside tools/phase_stamps.py (inputs / outputs in HBM)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import ctypes as C
import numpy as np
import torch
from g1_locomotion_amd import BatchMPC
import srbd_oracle as orc

names = ["linearise", "tables", "gradient", "K assembly", "barrier", "F(chol)", "W", "I", "frag", "ADMM", "rollout"]
x0, xr, ft, ct = orc.synthetic_batch(1, 10, 2000, os.environ.get("SCHED", "single"))
dev = torch.device("cuda", 0)
st_buf = torch.zeros((16, 16), dtype=torch.int64, device=dev)
for flight in (False, True):
    eng = BatchMPC(horizon=10)
    st = eng.stage()
    eng._lib.srbdqp_set_stamp_buffer(eng._h, C.c_void_p(st_buf.data_ptr()))
    for _ in range(5):
        st["x0"][0] = x0[0]; st["x_ref"][0] = xr[0]; st["foot"][0] = ft[0]; st["contact"][0] = 0 if flight else ct[0]
        eng.solve_staged(1, want_x=True)
    torch.cuda.synchronize()
    s = st_buf.cpu().numpy().astype(np.float64)[0]
    print(f"staged B=1 flight={flight} kernel={eng.kernel_name()} iters {int(st['iters'][0])}")
    if not flight:
        d = np.diff(s[:12])
        for nm, v in zip(names, d):
            print(f"  {nm:10s} {v:9.0f} cyc")
        print(f"  total {s[11] - s[0]:.0f} cyc; wall {(s[13] - s[12]) * 0.01:.2f} us")
    else:
        print("  stamps:", s[:14])
    eng.close()
