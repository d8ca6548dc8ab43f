"""Throughput of one engine on device-resident batches for other contact schedules / horizons than the headline config.
    python tools/schedule_bench.py [schedule=double] [N=10] [B=4096]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np, torch
import srbd_oracle as orc
from g1_locomotion_amd import BatchMPC
sched = sys.argv[1] if len(sys.argv) > 1 else "double"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
x0, xr, ft, ct = orc.synthetic_batch(B, N, seed=77, schedule=sched)
dev = torch.device("cuda", 0)
d = [torch.from_numpy(v).to(dev) for v in (x0, xr, ft, ct)]
maxs = 2 if sched == "single" else 4
eng = BatchMPC(horizon=N, max_contacts_per_step=maxs)
S = 2
streams = [torch.cuda.Stream() for _ in range(S)]
u = [torch.empty((B, N, 12), dtype=torch.float64, device=dev) for _ in range(S)]
it = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(S)]
st = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(S)]
def step(i):
    eng.set_schedule_hint(it[i % S].data_ptr() if i >= S else 0, B)
    eng.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), u[i % S].data_ptr(), status=st[i % S].data_ptr(), iters=it[i % S].data_ptr(), stream=streams[i % S].cuda_stream)
for i in range(4): step(i)
torch.cuda.synchronize()
t = time.perf_counter(); K = 20
for i in range(4, 4 + K): step(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / K
print(f"{sched} N={N} B={B} kernel={eng.kernel_name()} {B / dt / 1e6:.2f} M QP/s  {dt * 1e3:.3f} ms/step  mean iters {it[0].float().mean().item():.1f} solved {(st[0] == 1).float().mean().item():.4f}")
eng.close()
