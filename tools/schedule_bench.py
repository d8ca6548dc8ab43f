"""Throughput of one engine on device-resident batches for other contact schedules / horizons than the headline config.
    python tools/schedule_bench.py [schedule=double] [N=10] [B=4096] [kernel=auto|compact|wave|wrench] [f32=0] [rho_restart_iter=0 (library default)]
Steps rotate over 4 distinct batches on 2 streams with the longest-first hint, as bench.py does."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
from g1_locomotion_amd import BatchMPC, _lib, synth
sched = sys.argv[1] if len(sys.argv) > 1 else "double"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
kern = sys.argv[4] if len(sys.argv) > 4 else "auto"
f32 = bool(int(sys.argv[5])) if len(sys.argv) > 5 else False
restart = int(sys.argv[6]) if len(sys.argv) > 6 else 0
dev = torch.device("cuda", 0)
tdt = torch.float32 if f32 else torch.float64
NB, S = 4, int(os.environ.get("STREAMS", "2"))
d = [[torch.from_numpy(v).to(dev).to(tdt) if v.dtype == np.float64 else torch.from_numpy(v).to(dev) for v in synth.synthetic_batch(B, N, seed=77 + j, schedule=sched)] for j in range(NB)]
maxs = 2 if sched == "single" else 4
kid = {"auto": _lib.KERNEL_AUTO, "compact": _lib.KERNEL_COMPACT, "wave": _lib.KERNEL_WAVE, "wrench": _lib.KERNEL_WRENCH, "split": _lib.KERNEL_SPLIT}[kern]
eng = BatchMPC(horizon=N, max_contacts_per_step=maxs, kernel=kid, rho_restart_iter=restart)
streams = [torch.cuda.Stream() for _ in range(S)]
u = [torch.empty((B, N, 12), dtype=tdt, device=dev) for _ in range(NB)]
it = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(NB)]
st = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(NB)]
def step(i):
    o = i % NB
    eng.set_schedule_hint(it[o].data_ptr() if i >= NB else 0, B)
    eng.solve_device(B, d[o][0].data_ptr(), d[o][1].data_ptr(), d[o][2].data_ptr(), d[o][3].data_ptr(), u[o].data_ptr(), status=st[o].data_ptr(), iters=it[o].data_ptr(), stream=streams[i % S].cuda_stream, f32=f32)
for i in range(8): step(i)
torch.cuda.synchronize()
K = 40 if B * N <= 4096 * 12 else 12
t = time.perf_counter()
for i in range(8, 8 + K): step(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / K
its = torch.stack(it).float(); sts = torch.stack(st)
print(f"{sched} N={N} B={B} kernel={eng.kernel_name()} {B / dt / 1e6:.2f} M QP/s  {dt * 1e3:.3f} ms/step  mean iters {its.mean().item():.1f} solved {(sts == 1).float().mean().item():.4f}")
eng.close()
