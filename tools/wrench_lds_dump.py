"""Diagnostic (needs a -DSRBDQP_WRENCH_DEBUG build, SRBDQP_LIB=...): the general kernel's LDS image after its tables
against the oracle's intermediate quantities.   python tools/wrench_lds_dump.py [N] [schedule]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np, torch
import srbd_oracle as orc
from g1_locomotion_amd import BatchMPC, _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
sched = sys.argv[2] if len(sys.argv) > 2 else "mixed"
up2 = lambda v: (v + 1) & ~1
n = 12 * N
o = {}
o["x0"] = 0; o["tm"] = 14; o["J"] = o["tm"] + up2(9 * N); o["red"] = o["J"] + 36 * N; o["ct"] = o["red"] + 32
o["misc"] = o["ct"] + up2((4 * N + 7) // 8); o["sq"] = o["misc"] + 4; o["int"] = o["sq"] + 12
o["R"] = o["int"] + up2((3 * N + 6) // 2 + 1)
o["xref"] = o["R"]; o["foot"] = o["xref"] + up2(13 * N); o["pcom"] = o["foot"] + 12 * N; o["cp"] = o["pcom"] + up2(3 * N)
o["eh"] = o["cp"] + up2(9 * N); o["t1"] = o["eh"] + n; o["t2"] = o["t1"] + up2(9 * N); o["mt"] = o["t2"] + up2(9 * N)
o["gv"] = o["mt"] + up2(9 * N * (N + 1) // 2)
print(o)
x0, xr, ft, ct = orc.synthetic_batch(1, N, seed=300 + N, schedule=sched)
p = orc.SrbdParams()
with BatchMPC(horizon=N, kernel=_lib.KERNEL_WRENCH) as eng:
    d = eng.assemble_wrench(x0, xr, ft, ct)
img = d["T"][0].reshape(-1)
def cmp(name, got, want):
    got = np.asarray(got).reshape(-1); want = np.asarray(want).reshape(-1)
    e = np.abs(got - want).max()
    print(f"{name:6s} max err {e:.3e}", "" if e < 1e-9 else f"  first bad idx {np.where(np.abs(got - want) > 1e-9)[0][:8]} got {got[:6]} want {want[:6]}")
cmp("x0", img[o["x0"]:o["x0"] + 13], x0[0])
cmp("xref", img[o["xref"]:o["xref"] + 13 * N], xr[0])
cmp("foot", img[o["foot"]:o["foot"] + 12 * N], ft[0])
cmp("pcom", img[o["pcom"]:o["pcom"] + 3 * N], xr[0][:, 3:6])
cmp("sq", img[o["sq"]:o["sq"] + 12], np.sqrt(np.array(p.q_diag[:12])))
Tm = np.array([orc.rot_z(xr[0][k, 2]).T for k in range(N)])
cmp("tm", img[o["tm"]:o["tm"] + 9 * N], Tm)
cmp("cp", img[o["cp"]:o["cp"] + 9 * N], np.cumsum(Tm, axis=0))
Ib = np.diag(1.0 / np.array(p.inertia))
J = np.zeros((N, 3, 12))
for k in range(N):
    Rz = Tm[k].T
    for i in range(4):
        J[k][:, 3 * i:3 * i + 3] = Rz @ Ib @ Rz.T @ orc.skew(ft[0][k, 3 * i:3 * i + 3] - xr[0][k, 3:6])
cmp("J", img[o["J"]:o["J"] + 36 * N], J)
qp = orc.build_qp(p, x0[0], xr[0], ft[0], ct[0])
Qd = np.sqrt(np.array(p.q_diag[:12]))
eh = (qp["A_qp"] @ x0[0] - xr[0].reshape(-1)).reshape(N, 13)[:, :12] * Qd
cmp("eh", img[o["eh"]:o["eh"] + n], eh)
ints = img[o["int"]:o["R"]].view(np.int32)
print("gsz", ints[:N], "goff", ints[N:2 * N + 1], "misc", ints[2 * N + 1:2 * N + 3])
print("ct", img[o["ct"]:o["misc"]].view(np.uint8)[:4 * N], "want", ct[0].reshape(-1))
