"""Diagnostic: per-phase cycles of the general kernel on the STAGED batch-1 path (its low-latency instantiation; FLAG_NO_LAT=1
in the environment: the batch instantiation), inputs / outputs in GPU-mapped host memory.
    SCHED=double python tools/wrench_stamps_staged.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import ctypes as C
import numpy as np
import torch
from g1_locomotion_amd import BatchMPC, _lib, synth

names = ["load+tables", "E / V / Bd", "T assembly", "F (chol)", "W", "I", "fragments + x_q", "ADMM", "outputs + rollout"]
sched = os.environ.get("SCHED", "double")
N = int(os.environ.get("N", "10"))
x0, xr, ft, ct = synth.synthetic_batch(8, N, 2000, sched)
dev = torch.device("cuda", 0)
st_buf = torch.zeros((16, 16), dtype=torch.int64, device=dev)
flags = _lib.FLAG_NO_LAT if os.environ.get("FLAG_NO_LAT") else 0
eng = BatchMPC(horizon=N, kernel=_lib.KERNEL_WRENCH, flags=flags)
st = eng.stage()
eng._lib.srbdqp_set_stamp_buffer(eng._h, C.c_void_p(st_buf.data_ptr()))
tot = np.zeros(9); its = []
for b in range(8):
    for _ in range(3):
        st["x0"][0] = x0[b]; st["x_ref"][0] = xr[b]; st["foot"][0] = ft[b]; st["contact"][0] = ct[b]
        eng.solve_staged(1, want_x=True)
    torch.cuda.synchronize()
    s = st_buf.cpu().numpy().astype(np.float64)[0]
    tot += np.diff(s[:10]); its.append(int(st["iters"][0]))
tot /= 8
print(f"staged B=1 kernel={eng.kernel_name()} {sched} N={N} mean iters {np.mean(its):.1f}")
for nm, v in zip(names, tot):
    print(f"  {nm:18s} {v:9.0f} cyc")
print(f"  total {tot.sum():.0f} cyc = {tot.sum() / 2.4e3:.1f} us at 2.4 GHz; per ADMM iteration {tot[7] / np.mean(its):.0f} cyc")
eng.close()
s2 = st_buf.cpu().numpy().astype(np.float64)[1]
if s2[:7].any():   # -DSRBDQP_PROFILE_WADMM build: cycles per ADMM iteration by segment (last QP; each stamp drains the LDS queue first)
    seg = s2[:7] / max(its[-1], 1)
    for nm, v in zip(["w exchange + V w + v publish", "barrier", "check decision + T^-1 v", "t exchange + x~", "fz bpermute", "cone rows + A'", "pre-test / check"], seg):
        print(f"    {nm:42s} {v:7.0f} cyc / iteration")
    print(f"    sum {seg.sum():.0f}; full convergence checks in this solve: {int(s2[7])} over {its[-1]} iterations")
