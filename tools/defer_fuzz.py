"""Fuzz of the deferred tails (SRBDQP_FLAG_DEFER_TAIL): a random sequence of device-buffer solves -- batch sizes, batches, streams, dispatch hints and flushes drawn at
random, every solve with its own output buffers -- against the same solves with the restart in place; every status, iteration count and force must be equal.
    python tools/defer_fuzz.py [solves=300] [seed=0]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from g1_locomotion_amd import BatchMPC, synth, _lib

n_solves = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda", 0)
N = 10
sizes = [64, 300, 1024, 2048, 4096, 5000]
pool = {B: [[torch.from_numpy(v).to(dev) for v in synth.synthetic_batch(B, N, seed=int(rng.integers(1 << 30)), schedule="single")] for _ in range(3)] for B in sizes}
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
kw = dict(max_contacts_per_step=2, kernel=_lib.KERNEL_WAVE, rho_restart_iter=int(rng.choice([20, 30, 55])), rho_restart_count=int(rng.choice([1, 2, 3])))
plan = []
for k in range(n_solves):
    B = int(rng.choice(sizes))
    plan.append(dict(B=B, j=int(rng.integers(3)), s=int(rng.integers(3)), hint=bool(rng.integers(2)), flush=rng.random() < 0.15, flush_all=rng.random() < 0.3))


def run(flags):
    # every solve's outputs exist (and their fills have run) before the first solve: the fills go to torch's current stream, the solves to others
    outs = [dict(u=torch.zeros((p["B"], N, 12), dtype=torch.float64, device=dev), x=torch.zeros((p["B"], N + 1, 13), dtype=torch.float64, device=dev),
                 st=torch.full((p["B"],), -9, dtype=torch.int32, device=dev), it=torch.zeros(p["B"], dtype=torch.int32, device=dev)) for p in plan]
    torch.cuda.synchronize(dev)
    with BatchMPC(horizon=N, flags=flags, **kw) as eng:
        last_it = {}
        for k_, p in enumerate(plan):
            B, d = p["B"], pool[p["B"]][p["j"]]
            o = outs[k_]
            if p["hint"] and B in last_it:
                eng.set_schedule_hint(last_it[B].data_ptr(), B)
            else:
                eng.set_schedule_hint(0, 0)
            eng.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), o["u"].data_ptr(), x_out=o["x"].data_ptr(),
                             status=o["st"].data_ptr(), iters=o["it"].data_ptr(), stream=streams[p["s"]].cuda_stream)
            if p["flush"]:
                eng.flush(0 if p["flush_all"] else streams[p["s"]].cuda_stream)
                if flags == 0 or p["flush_all"]:
                    torch.cuda.synchronize(dev)
                    last_it[B] = o["it"].clone()              # (a complete iteration-count array of this size, for later hints)
        eng.flush()
        torch.cuda.synchronize(dev)
    return outs


ref = run(0)
got = run(_lib.FLAG_DEFER_TAIL)
bad = 0
restarted = 0
for k, (o, r) in enumerate(zip(got, ref)):
    restarted += int((r["it"] > kw["rho_restart_iter"]).sum())
    ok = torch.equal(o["st"], r["st"]) and torch.equal(o["it"], r["it"]) and float((o["u"] - r["u"]).abs().max()) <= 1e-9 and float((o["x"] - r["x"]).abs().max()) <= 1e-11
    if not ok:
        bad += 1
        print("MISMATCH at solve", k, plan[k], int((o["st"] != r["st"]).sum()), int((o["it"] != r["it"]).sum()), float((o["u"] - r["u"]).abs().max()),
              "status counts got", torch.bincount(o["st"] + 9, minlength=12).tolist(), "ref", torch.bincount(r["st"] + 9, minlength=12).tolist(),
              "iters got max", int(o["it"].max()), "ref max", int(r["it"].max()))
print(f"defer fuzz: {n_solves} solves, restart {kw['rho_restart_iter']} x {kw['rho_restart_count']}, {restarted} continued QPs, {bad} mismatching solves")
sys.exit(1 if bad else 0)
