"""Do workgroups of a short-horizon bucket run beside the one-workgroup-per-CU N = 24 kernel?  Two homogeneous batches on their own engines (own streams):
each alone, then both in flight together.    python tools/coresidency_probe.py [N_small=8] [B_small=16384]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
from g1_locomotion_amd import BatchMPC, _lib, synth
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
BS = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
dev = torch.device("cuda", 0)


def leg(N, B):
    arrs = synth.synthetic_batch(B, N, seed=40 + N, schedule="mixed")
    dd = [torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in arrs]
    u = torch.empty((B, N, 12), dtype=torch.float64, device=dev)
    e = BatchMPC(horizon=N, kernel=_lib.KERNEL_WRENCH)
    return e, (lambda: e.solve_device(B, dd[0].data_ptr(), dd[1].data_ptr(), dd[2].data_ptr(), dd[3].data_ptr(), u.data_ptr())), (dd, u)


def timeit(fs, es, K=5):
    for f in fs: f()
    for e in es: e.synchronize()
    t = time.perf_counter()
    for _ in range(K):
        for f in fs: f()
        for e in es: e.synchronize()
    return (time.perf_counter() - t) / K * 1e3


e24, f24, k24 = leg(24, 4096)
es, fs, ks = leg(NS, BS)
a, b = timeit([f24], [e24]), timeit([fs], [es])
c = timeit([f24, fs], [e24, es])
d = timeit([fs, f24], [e24, es])
print(f"N=24 x 4096 alone {a:.2f} ms; N={NS} x {BS} alone {b:.2f} ms; together (N=24 first) {c:.2f} ms, (N={NS} first) {d:.2f} ms; sum {a + b:.2f} ms")
