"""Debug aid: the general kernel's assembly dump and solve statuses against the oracle, per horizon / schedule.
    python tools/wrench_debug.py [N=12] [schedule=mixed] [B=4]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np, torch
import srbd_oracle as orc
from g1_locomotion_amd import BatchMPC, _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
sched = sys.argv[2] if len(sys.argv) > 2 else "mixed"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
x0, xr, ft, ct = orc.synthetic_batch(B, N, seed=300 + N, schedule=sched)
p = orc.SrbdParams()
with BatchMPC(horizon=N, kernel=_lib.KERNEL_WRENCH) as eng:
    d = eng.assemble_wrench(x0, xr, ft, ct)
    print("goff", d["goff"])
    out = eng.solve(x0, xr, ft, ct, want_y=True)
    o32 = eng.solve(x0, xr, ft, ct, want_y=True, dtype=np.float32)
for b in range(B):
    wr = orc.wrench_reduce(p, xr[b], ft[b], ct[b])
    ng = wr["n_g"]
    Tg = d["T"][b][:ng, :ng]
    print(b, "n_g", ng, d["goff"][b][-1], "T err", np.abs(Tg - wr["T"]).max() / np.abs(wr["T"]).max(), "nan" if not np.isfinite(Tg).all() else "",
          "outside", np.abs(d["T"][b]).sum() - np.abs(Tg).sum())
    qp = orc.build_qp(p, x0[b], xr[b], ft[b], ct[b])
    qe = np.abs(d["q"][b] - qp["q"]) / np.abs(qp["q"]).max()
    print("   q err", qe.max(), "bad steps", sorted(set((np.where(qe > 1e-9)[0] // 12).tolist())))
    Te = np.abs(Tg - wr["T"]) / np.abs(wr["T"]).max()
    gst = np.repeat(np.arange(N), wr["gsz"])
    bad = np.argwhere(Te > 1e-9)
    print("   T bad step pairs", sorted(set((int(gst[r]), int(gst[c])) for r, c in bad))[:40])
    # Bd / V against the oracle (per stance variable, within its step)
    vi = wr["vi"]
    goff = wr["goff"]
    eb = ev = 0.0
    for idx, v in enumerate(vi):
        k = v // 12
        us = [i for i, vv in enumerate(vi) if vv // 12 == k]
        cols = [vi[i] % 12 for i in us]
        eb = max(eb, np.abs(d["Bd"][b][v][cols] - wr["Bd"][idx, us]).max())
        ev = max(ev, np.abs(d["Vcol"][b][v][:goff[k + 1] - goff[k]] - wr["V"][goff[k]:goff[k + 1], idx]).max())
    print("   Bd err", eb, "Vcol err", ev)
    ref = orc.update(p, x0[b], xr[b], ft[b], ct[b])
    print("   f64 status", out["status"][b], ref["status"], "iters", out["iters"][b], ref["iters"], "du", np.abs(out["u"][b] - ref["u"]).max(),
          "| f32 status", o32["status"][b], "iters", o32["iters"][b], "du", np.abs(o32["u"][b] - ref["u"]).max())
