"""Per-call time (one solve at a time, synchronised) of the 4-wave and the wave kernel over the batch size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np, torch
import srbd_oracle as orc
from g1_locomotion_amd import BatchMPC, _lib
dev = torch.device("cuda", 0)
for B in (1, 8, 32, 128, 256, 512, 1024, 1536, 2048, 3072, 4096):
    x0, xr, ft, ct = orc.synthetic_batch(B, 10, seed=5, schedule="single")
    d = [torch.from_numpy(v).to(dev) for v in (x0, xr, ft, ct)]
    u = torch.empty((B, 10, 12), dtype=torch.float64, device=dev)
    row = [B]
    for kid in (_lib.KERNEL_COMPACT, _lib.KERNEL_WAVE):
        with BatchMPC(horizon=10, max_contacts_per_step=2, kernel=kid) as eng:
            ts = []
            for i in range(30):
                torch.cuda.synchronize()
                t = time.perf_counter()
                eng.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), u.data_ptr())
                eng.synchronize()
                ts.append(time.perf_counter() - t)
            row.append(round(1e6 * float(np.median(ts[5:])), 1))
    print("B=%d  compact %.1f us  wave %.1f us" % tuple(row))
