"""Throughput of BASELINE.json configs[4] -- a mixed-horizon fleet, N in {8, 12, 16, 24} drawn uniformly per QP, per-QP mixed-gait
contact schedules, packed step-major arrays resident in HBM -- through srbdqp_solve_ragged_device_f64 (one call: bucket
permutation + one launch per horizon bucket, each on its own stream).
    python tools/ragged_bench.py [B=16384]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
from g1_locomotion_amd import RaggedMPC, BatchMPC, _lib, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
HZ = (8, 12, 16, 24)
rng = np.random.default_rng(4)
Nq = rng.choice(HZ, size=B).astype(np.int32)
x0 = np.empty((B, 13)); xr, ft, ct = [], [], []
by = {}
for N in HZ:
    idx = np.where(Nq == N)[0]
    a, b_, c, d = synth.synthetic_batch(len(idx), N, seed=40 + N, schedule="mixed")
    by[N] = (idx, a, b_, c, d)
pos = {N: 0 for N in HZ}
for b in range(B):
    N = int(Nq[b]); idx, a, b_, c, d = by[N]; i = pos[N]; pos[N] += 1
    x0[b] = a[i]; xr.append(b_[i]); ft.append(c[i].reshape(N, 12)); ct.append(d[i].reshape(N, 4))
xr, ft, ct = np.concatenate(xr), np.concatenate(ft), np.concatenate(ct).astype(np.uint8)
rows = int(Nq.sum())
dev = torch.device("cuda", 0)
d_x0, d_xr, d_ft, d_ct = (torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (x0, xr, ft, ct))
d_u = torch.empty((rows, 12), dtype=torch.float64, device=dev)
d_st = torch.empty(B, dtype=torch.int32, device=dev); d_it = torch.empty(B, dtype=torch.int32, device=dev)
eng = RaggedMPC(horizons=HZ)
run = lambda: eng.solve_device(B, Nq, d_x0.data_ptr(), d_xr.data_ptr(), d_ft.data_ptr(), d_ct.data_ptr(), d_u.data_ptr(), status=d_st.data_ptr(), iters=d_it.data_ptr())
for _ in range(2): run()
torch.cuda.synchronize()
K = 10
t = time.perf_counter()
for _ in range(K): run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / K
st, it = d_st.cpu().numpy(), d_it.cpu().numpy()
print(f"ragged B={B} horizons {HZ} (uniform), mixed gait, fp64: {B / dt / 1e6:.3f} M QP/s  {dt * 1e3:.2f} ms/call  {rows / dt / 1e6:.1f} M horizon steps/s  solved {np.mean(st == 1):.4f}")
for N in HZ:
    m = Nq == N
    print(f"   N={N:2d}: {m.sum():6d} QPs  mean iters {it[m].mean():5.1f}  solved {np.mean(st[m] == 1):.4f}")
# the same buckets one after another through BatchMPC (what a host-side loop over homogeneous batches would get)
tot = 0.0
for N in HZ:
    idx, a, b_, c, d = by[N]
    n = len(idx)
    dd = [torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (a, b_, c, d)]
    u = torch.empty((n, N, 12), dtype=torch.float64, device=dev)
    with BatchMPC(horizon=N, kernel=_lib.KERNEL_WRENCH) as e1:
        r1 = lambda: e1.solve_device(n, dd[0].data_ptr(), dd[1].data_ptr(), dd[2].data_ptr(), dd[3].data_ptr(), u.data_ptr())
        r1(); e1.synchronize()
        t = time.perf_counter()
        for _ in range(K): r1()
        e1.synchronize()
        tot += (time.perf_counter() - t) / K
print(f"   the four buckets as separate homogeneous batches, one after another: {tot * 1e3:.2f} ms = {B / tot / 1e6:.3f} M QP/s")
