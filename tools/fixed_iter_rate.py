"""Diagnostic: QP/s of one kernel at a FIXED iteration count (eps = 0, no restart), so that builds whose set-up differs -- or is partly stubbed out
(-DSRBDQP_EXP_...) -- compare like for like.   python tools/fixed_iter_rate.py [N=10] [schedule=single] [B=65536] [iters=35] [kernel=0]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from g1_locomotion_amd import BatchMPC, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
sched = sys.argv[2] if len(sys.argv) > 2 else "single"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
K = int(sys.argv[4]) if len(sys.argv) > 4 else 35
kern = int(sys.argv[5]) if len(sys.argv) > 5 else 0
dev = torch.device("cuda", 0)
d = [torch.from_numpy(v).to(dev) for v in synth.synthetic_batch(B, N, 2000, sched)]
u = torch.empty((B, N, 12), dtype=torch.float64, device=dev)
it = torch.empty(B, dtype=torch.int32, device=dev)
mc = 2 if sched == "single" else 4
with BatchMPC(horizon=N, max_contacts_per_step=mc, kernel=kern, max_iter=K, eps_abs=1e-300, eps_rel=0.0, rho_restart_iter=-1) as eng:
    def run():
        eng.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), u.data_ptr(), iters=it.data_ptr())
    for _ in range(3): run()
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): run()
    eng.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"{eng.kernel_name()} N={N} {sched} B={B} {K} iterations: {dt * 1e3:.3f} ms per launch, {B / dt / 1e6:.2f} M QP/s")
