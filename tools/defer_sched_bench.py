"""Step rate of the kernels that restart by further launches, with the passes on the caller's stream and on the tail stream (SRBDQP_FLAG_DEFER_TAIL).
usage: python tools/defer_sched_bench.py schedule N B [f32]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from g1_locomotion_amd import BatchMPC, synth, _lib

sched, N, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
f32 = len(sys.argv) > 4 and sys.argv[4] == "f32"
dev = torch.device("cuda", 0)
tdt = torch.float32 if f32 else torch.float64
NB, NO = 4, 8
hb = [synth.synthetic_batch(B, N, seed=1000 + 97 * j, schedule=sched) for j in range(NB)]
d_in = [[torch.from_numpy(v).to(dev).to(tdt) if v.dtype == np.float64 else torch.from_numpy(v).to(dev) for v in b] for b in hb]
d_u = [torch.zeros((B, N, 12), dtype=tdt, device=dev) for _ in range(NO)]
d_x = [torch.zeros((B, N + 1, 13), dtype=tdt, device=dev) for _ in range(NO)]
d_st = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(NO)]
d_it = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(NO)]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
K = max(8, min(200, (1 << 20) // B))


def run(flags, S, **kw):
    with BatchMPC(horizon=N, flags=flags, **kw) as eng:
        def step(i):
            o, d = i % NO, d_in[i % NB]
            eng.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d_u[o].data_ptr(), x_out=d_x[o].data_ptr(),
                             status=d_st[o].data_ptr(), iters=d_it[o].data_ptr(), stream=streams[i % S].cuda_stream, f32=f32)
        for i in range(NO):
            step(i)
        eng.flush(); torch.cuda.synchronize(dev)
        best = 0.0
        for _ in range(3):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(K):
                step(i)
            eng.flush()
            torch.cuda.synchronize(dev)
            best = max(best, B * K / (time.perf_counter() - t0))
        st = torch.stack(d_st).cpu().numpy(); it = torch.stack(d_it).cpu().numpy()
        return best / 1e6, float((st == 1).mean()), float(it.mean()), eng.kernel_name()


for name, flags, kw in (("restart off", 0, dict(rho_restart_iter=-1)), ("in stream", 0, {}), ("tail stream", _lib.FLAG_DEFER_TAIL, {})):
    for S in (1, 2):
        v, solved, it, kn = run(flags, S, **kw)
        print(f"{sched} N={N} B={B} {'f32' if f32 else 'f64'} {name:12s} streams={S} {v:7.2f} M QP/s  solved {solved:.4f}  iters {it:.2f}  {kn}", flush=True)
