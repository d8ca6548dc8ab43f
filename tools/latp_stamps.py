"""Diagnostic (library built with -DSRBDQP_LATP_STAMPS): when wave 0 passes the barriers of the pipelined tile phases of the low-latency general kernel.
    SCHED=double python tools/latp_stamps.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import ctypes as C
import numpy as np
import torch
from g1_locomotion_amd import BatchMPC, _lib, synth
sched = os.environ.get("SCHED", "double"); N = int(os.environ.get("N", "10"))
x0, xr, ft, ct = synth.synthetic_batch(8, N, 2000, sched)
dev = torch.device("cuda", 0)
st_buf = torch.zeros((16, 16), dtype=torch.int64, device=dev)
eng = BatchMPC(horizon=N, kernel=_lib.KERNEL_WRENCH)
st = eng.stage()
eng._lib.srbdqp_set_stamp_buffer(eng._h, C.c_void_p(st_buf.data_ptr()))
names = ["A0", "B0", "M0", "A1", "B1", "M1", "A2", "B2", "M2", "A3", "M3", "A4", "end"]
acc = np.zeros((4, 13)); cnt = 0; joins = np.zeros((4, 2))
for b in range(8):
    for _ in range(3):
        st["x0"][0] = x0[b]; st["x_ref"][0] = xr[b]; st["foot"][0] = ft[b]; st["contact"][0] = ct[b]
        eng.solve_staged(1, want_x=True)
    torch.cuda.synchronize()
    s = st_buf.cpu().numpy().astype(np.float64)
    acc += s[1:5, :13] - s[0][3]; cnt += 1          # cycles since the end of the T assembly (stamp 3 of row 0)
    joins += s[1:5, 13:15] - np.array([s[0][1], s[0][2]])[None, :]
acc /= cnt
print(eng.kernel_name(), sched, "arrival of each wave at the barriers of the tile pipeline, cycles since the end of the assembly (N = 10, four block columns):")
print("barrier " + " ".join(f"{n:>6s}" for n in names))
for w in range(4):
    print(f"wave {w}  " + " ".join(f"{v:6.0f}" for v in acc[w]))
print("last    " + " ".join(f"{v:6.0f}" for v in acc.max(axis=0)))
print("segment " + " ".join(f"{v:6.0f}" for v in np.diff(np.concatenate([[0], acc.max(axis=0)]))))
e = st_buf.cpu().numpy().astype(np.float64)[5][:4]
print("phase E on wave 0 (last solve), cycles: E build", int(e[1] - e[0]), " inverse", int(e[2] - e[1]), " er / yv / V / Bd + stores", int(e[3] - e[2]), " from stamp 1 to the start of E", int(e[0] - st_buf.cpu().numpy().astype(np.float64)[0][1]))
r5 = st_buf.cpu().numpy().astype(np.float64)[5]; r0 = st_buf.cpu().numpy().astype(np.float64)[0]
if r5[4] > 0:
    print("roll-out (last solve), cycles: stamp 8 -> start", int(r5[4] - r0[8]), " u store", int(r5[5] - r5[4]), " s_j + barrier", int(r5[6] - r5[5]), " prefix + barrier", int(r5[7] - r5[6]), " x rows + store -> stamp 9", int(r0[9] - r5[7]))
joins /= cnt
print("arrival at the join behind the tables / E (cycles since stamp 1 = wave 0 past load + linearise):", np.round(joins[:, 0]).astype(int).tolist(), "  at the barrier behind the T assembly (since stamp 2):", np.round(joins[:, 1]).astype(int).tolist())
eng.close()
