"""Diagnostic: per-phase time of the v1 kernel from in-kernel s_memtime stamps (100 MHz), batch-1 and full batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import ctypes as C
import numpy as np
import torch
from g1_locomotion_amd import BatchMPC
from g1_locomotion_amd import synth as orc

names = ["linearise", "tables", "gradient", "K assembly", "barrier", "F(chol)", "W", "I", "frag", "ADMM", "rollout"]
for B in (1, 4096):
    x0, xr, ft, ct = orc.synthetic_batch(B, 10, 2000, os.environ.get("SCHED", "single"))
    dev = torch.device("cuda", 0)
    d = [torch.from_numpy(v).to(dev) for v in (x0, xr, ft, ct)]
    u = torch.empty((B, 10, 12), dtype=torch.float64, device=dev)
    it = torch.empty(B, dtype=torch.int32, device=dev)
    st = torch.zeros((B, 16), dtype=torch.int64, device=dev)
    eng = BatchMPC(horizon=10, max_contacts_per_step=int(os.environ.get("MAXS", "2")), kernel=int(os.environ.get("KERNEL", "0")), **({"max_iter": 35, "eps_abs": 0.0, "eps_rel": 0.0, "check_every": int(os.environ.get("CHECK_EVERY", "5"))} if os.environ.get("FIXED_ITERS") else {}))
    eng._lib.srbdqp_set_stamp_buffer(eng._h, C.c_void_p(st.data_ptr()))
    for _ in range(3):
        eng.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), u.data_ptr(), iters=it.data_ptr())
        eng.synchronize()
    s = st.cpu().numpy().astype(np.float64)
    its = it.cpu().numpy()
    dlt = np.diff(s[:, :12], axis=1) * 100.0  # s_memtime ticks are shader cycles; printed 'us' column = cycles / 10
    print(f"B={B} kernel={eng.kernel_name()} mean iters {its.mean():.1f}")
    for i, nm in enumerate(names):
        print(f"  {nm:10s} mean {dlt[:, i].mean() / 100:9.0f} cyc   median {np.median(dlt[:, i]) / 100:9.0f} cyc")
    tot = (s[:, 11] - s[:, 0]) * 100.0
    print(f"  total      mean {tot.mean() / 100:9.0f} cyc   per ADMM iteration {np.mean(dlt[:, 9] / np.maximum(its, 1)) / 100:.0f} cyc")
    if os.environ.get("PROFILE") == "ADMM":
        print("  ADMM segments (cycles/iter): matvec %.0f  Arow+rows %.0f  At(w)+update+write %.0f  check(amortised) %.0f  barrier %.0f" % tuple((s[:, c] / np.maximum(its, 1)).mean() for c in (12, 13, 14, 15, 1)))
    if not os.environ.get("PROFILE") and s[:, 13].any():
        rt = (s[:, 13] - s[:, 12]) * 10.0   # ns, 100 MHz constant clock
        clk = (s[:, 11] - s[:, 0]) / np.maximum(rt, 1.0)
        t0 = s[:, 12].min(); ends = (s[:, 13] - t0) * 1e-2; starts = (s[:, 12] - t0) * 1e-2
        print(f"  shader clock (memtime/memrealtime): median {np.median(clk):.2f} GHz; per-QP wall: mean {rt.mean()/1e3:.1f} us  p50 {np.median(rt)/1e3:.1f}  p99 {np.percentile(rt,99)/1e3:.1f}  max {rt.max()/1e3:.1f} us")
        print(f"  kernel span {ends.max():.1f} us; last start {starts.max():.1f} us; 50%/90%/99% of QPs finished by {np.percentile(ends,50):.0f}/{np.percentile(ends,90):.0f}/{np.percentile(ends,99):.0f} us; sum(wall)/512 = {rt.sum()/512e3:.1f} us")
    if os.environ.get("PROFILE") == "F" and s[:, 12:16].any():
        print("  F segments (cycles total over 8 steps): store+barrier %.0f  diag16 %.0f  panel %.0f  trailing %.0f" % tuple(s[:, 12 + i].mean() for i in range(4)))
    elif False:
        print("  ADMM segments (cycles/iter): matvec %.0f  rows+At %.0f  write+barrier %.0f  | check iterations: %.0f cycles per check" % (
            (s[:, 12] / its).mean(), (s[:, 13] / its).mean(), (s[:, 14] / np.maximum(its - its // 5, 1)).mean(), (s[:, 15] / np.maximum(its // 5, 1)).mean()))
    eng.close()
