"""configs[1] step rate with and without deferred tails (SRBDQP_FLAG_DEFER_TAIL): streams x dispatch hint x timed-region length.
usage: python tools/defer_bench.py [B] [N]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from g1_locomotion_amd import BatchMPC, synth, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
NB = 4
hb = [synth.synthetic_batch(B, N, seed=1000 + 97 * j, schedule="single") for j in range(NB)]
d_in = [[torch.from_numpy(v).to(dev) for v in b] for b in hb]
NO = 8
d_u = [torch.zeros((B, N, 12), dtype=torch.float64, device=dev) for _ in range(NO)]
d_x = [torch.zeros((B, N + 1, 13), dtype=torch.float64, device=dev) for _ in range(NO)]
d_st = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(NO)]
d_it = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(NO)]
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]


def run(flags, S, hint, K, reps=3, **kw):
    with BatchMPC(horizon=N, max_contacts_per_step=2, flags=flags, **kw) as eng:
        def step(i):
            o, d = i % NO, d_in[i % NB]
            if hint == "own":
                eng.set_schedule_hint(d_it[(i % NB)].data_ptr() if False else d_it[o].data_ptr(), B)
            elif hint == "stale":
                eng.set_schedule_hint(d_it[(o + 1) % NO].data_ptr(), B)
            else:
                eng.set_schedule_hint(0, 0)
            eng.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d_u[o].data_ptr(), x_out=d_x[o].data_ptr(),
                             status=d_st[o].data_ptr(), iters=d_it[o].data_ptr(), stream=streams[i % S].cuda_stream)
        for i in range(2 * NO):
            step(i)
        eng.flush(); torch.cuda.synchronize(dev)
        best = 0.0
        for _ in range(reps):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(K):
                step(i)
            eng.flush()
            torch.cuda.synchronize(dev)
            best = max(best, B * K / (time.perf_counter() - t0))
        st = torch.stack(d_st).cpu().numpy(); it = torch.stack(d_it).cpu().numpy()
        return best / 1e6, float((st == 1).mean()), float(it.mean()), eng.kernel_name()


print(f"B = {B}, N = {N}: M QP/s (best of 3), solved, mean iters")
for K in (20, 200):
    for name, flags in (("in place", 0), ("deferred", _lib.FLAG_DEFER_TAIL)):
        for S in (1, 2, 3):
            for hint in ("none", "stale", "own"):
                v, solved, it, kn = run(flags, S, hint, K)
                print(f"K={K:4d} {name:9s} streams={S} hint={hint:6s} {v:7.2f} M QP/s  solved {solved:.4f}  iters {it:.2f}  {kn}", flush=True)
