#!/usr/bin/env python3
"""bench.py -- QP solves/sec of the SRBD convex-MPC hot path on MI355X (BASELINE.json's metric).

A "step" = one pass of the hot path (linearise -> condense -> H,g,cone rows -> factor -> ADMM -> rollout) over one
batch of synthetic QPs whose inputs are already resident in HBM.  Workload at every N: BASELINE.json configs[1]
(B = 4096 random SRBD states per GPU, horizon 10, 2-contact alternating single support, fp64); with N > 1 ranks
each rank owns its own 4096 QPs (weak scaling, no data-path collective) and the first-step contact forces
u_opt0 are all-gathered over RCCL/xGMI every step, as north_star specifies.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline`: fp64 dense peak (78.6 TFLOP/s per MI355X, vector == matrix rate) against
the ALGORITHMIC flops of SURVEY.md section 8(d), W(N, K) with K = the measured mean ADMM iteration count.
`cpu_baseline`: oracle/srbd_oracle.c (a plain-C port of the same algorithm; the reference's own implementation is
an absent submodule) timed on this node's host cores on a bounded sample of the same batch.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

HORIZON = 10
BATCH_PER_GPU = 4096
PEAK_FP64_TFLOPS = 78.6          # MI355X dense fp64 (vector == MFMA rate), SURVEY.md section 8(d)


def algorithmic_flops(N: int, K: float) -> float:
    """W(N, K) flops per QP."""
    n, s, m = 12 * N, 13 * N, 20 * N
    return (500 * N + 2 * 13 ** 3 * N + N * (N + 1) / 2 * 2 * 13 * 13 * 12 + s * n * n
            + (2 * s * 13 + 2 * s * n) + n ** 3 / 3 + K * (2 * n * n + 10 * m))


def algorithmic_bytes(N: int) -> int:
    """HBM bytes per QP, fp64 (inputs + outputs)."""
    return (13 + 13 * N + 12 * N + 4 * N) * 8 + (12 * N + 13 * (N + 1)) * 8 + 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="QPs per GPU per step (default: configs[1])")
    ap.add_argument("--kernel", choices=["auto", "gj", "mfma", "compact", "split", "wave"], default="auto",
                    help="auto = wave (one wave per QP) for this batch size; split = the same as two kernels; compact = the 4-wave fused kernel")
    ap.add_argument("--streams", type=int, default=2,
                    help="consecutive steps alternate over this many HIP streams, so the straggler tail of one batch "
                         "(QPs that need many ADMM iterations) overlaps the bulk of the next; 1 = strictly serial steps")
    ap.add_argument("--no-sched-hint", action="store_true",
                    help="do not feed the previous step's iteration counts back as the longest-first dispatch hint")
    ap.add_argument("--max-iter", type=int, default=0, help="override srbdqp_config.max_iter (0 = library default)")
    ap.add_argument("--rho-restart", type=int, default=0, help="override srbdqp_config.rho_restart_iter (0 = library default: off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback for the hot path)"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        dist = dist_mod

    from g1_locomotion_amd import BatchMPC, _lib
    sys.path.insert(0, os.path.join(ROOT, "oracle"))   # synthetic-input generator lives with the oracle
    import srbd_oracle as orc

    N, B = HORIZON, args.batch
    x0, xr, ft, ct = orc.synthetic_batch(B, N, seed=1000 * 2 + rank, schedule="single")
    dev = torch.device("cuda", local_rank)
    d_x0 = torch.from_numpy(x0).to(dev)
    d_xr = torch.from_numpy(xr).to(dev)
    d_ft = torch.from_numpy(ft).to(dev)
    d_ct = torch.from_numpy(ct).to(dev)
    S = max(1, args.streams)
    # one set of output buffers per stream (steps in flight at the same time must not share outputs)
    d_u = [torch.empty((B, N, 12), dtype=torch.float64, device=dev) for _ in range(S)]
    d_x = [torch.empty((B, N + 1, 13), dtype=torch.float64, device=dev) for _ in range(S)]
    d_st = [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(S)]
    d_it = [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(S)]
    d_u0_all = [torch.empty((world * B, 12), dtype=torch.float64, device=dev) for _ in range(S)] if world > 1 else None

    kid = {"auto": _lib.KERNEL_AUTO, "gj": _lib.KERNEL_GJ, "mfma": _lib.KERNEL_MFMA, "compact": _lib.KERNEL_COMPACT,
           "split": _lib.KERNEL_SPLIT, "wave": _lib.KERNEL_WAVE}[args.kernel]
    # configs[1] is the 2-contact (single support) workload: at most 2 stance contact points per horizon step
    eng = BatchMPC(horizon=N, device=local_rank, kernel=kid, max_contacts_per_step=2,
                   **({"max_iter": args.max_iter} if args.max_iter > 0 else {}),
                   **({"rho_restart_iter": args.rho_restart} if args.rho_restart > 0 else {}))
    # non-default streams: the C-ABI treats a NULL stream as "the handle's own", and the HIP events that time a kernel
    # must sit on the stream the kernel is launched on
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]

    def step(i):
        st = streams[i % S]
        if not args.no_sched_hint:   # receding-horizon deployment: the iteration counts this stream's previous step produced
            eng.set_schedule_hint(d_it[i % S].data_ptr() if i >= S else 0, B)
        eng.solve_device(B, d_x0.data_ptr(), d_xr.data_ptr(), d_ft.data_ptr(), d_ct.data_ptr(), d_u[i % S].data_ptr(),
                         x_out=d_x[i % S].data_ptr(), status=d_st[i % S].data_ptr(), iters=d_it[i % S].data_ptr(),
                         stream=st.cuda_stream)

    def exchange(i):
        if dist is not None:   # all-gather of u_opt0 on the step's own stream: overlaps the next step's kernel
            with torch.cuda.stream(streams[i % S]):
                dist.all_gather_into_tensor(d_u0_all[i % S], d_u[i % S][:, 0, :].contiguous())

    torch.cuda.synchronize(dev)
    for i in range(args.warmup):
        step(i)
        exchange(i)
    torch.cuda.synchronize(dev)
    # ---- the dominant kernel in isolation (HIP events on its launch stream, one launch at a time): roofline numbers
    iso = []
    for i in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(streams[0]); step(2 * S); e1.record(streams[0])   # index >= S: same scheduling hint as the timed steps
        torch.cuda.synchronize(dev)
        iso.append(e0.elapsed_time(e1))
    kernel_ms = float(np.mean(iso))
    # the split pipeline's two kernels separately (library-side HIP events between them; a second, timed engine so
    # that the benchmarked engine carries no event records)
    parts = None
    if eng.kernel_name().startswith("split_"):
        with BatchMPC(horizon=N, device=local_rank, kernel=kid, max_contacts_per_step=2, timing=True,
                      **({"max_iter": args.max_iter} if args.max_iter > 0 else {})) as teng:
            acc = []
            for i in range(6):
                teng.set_schedule_hint(0 if args.no_sched_hint else d_it[0].data_ptr(), B)
                teng.solve_device(B, d_x0.data_ptr(), d_xr.data_ptr(), d_ft.data_ptr(), d_ct.data_ptr(), d_u[0].data_ptr(),
                                  x_out=d_x[0].data_ptr(), status=d_st[0].data_ptr(), iters=d_it[0].data_ptr(), stream=streams[0].cuda_stream)
                torch.cuda.synchronize(dev)
                pr = teng.last_kernel_parts_ms()
                if pr is not None and i > 0:
                    acc.append(pr)
            if acc:
                parts = (float(np.mean([p[0] for p in acc])), float(np.mean([p[1] for p in acc])))
    if dist is not None:
        dist.barrier()
    # ---- timed region: exactly K steps
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
        exchange(k)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    iters = d_it[0].cpu().numpy()
    status = d_st[0].cpu().numpy()
    mean_iters = float(iters.mean())
    solved_frac = float((status == _lib.SOLVED).mean())

    if rank == 0:
        total_qp = world * B * args.steps
        value = total_qp / elapsed
        flops_launch = algorithmic_flops(N, mean_iters) * B
        achieved_tf = flops_launch / (kernel_ms * 1e-3) / 1e12
        hbm_gbs = algorithmic_bytes(N) * B / (kernel_ms * 1e-3) / 1e9
        traffic = None   # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), same kernel + workload
        try:
            import glob
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))):
                d = json.load(open(f))
                if d.get("bench_kernel_name") == eng.kernel_name() and d.get("batch_per_launch") == B:
                    traffic = d["hbm"]["traffic_bytes_per_launch"]
        except Exception:
            traffic = None
        out = {
            "metric": "QP solves/sec, SRBD N=10 12-state/12-input",
            "value": value, "unit": "QP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / max(args.steps, 1), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"configs[1]: batch={B}/GPU random SRBD states, N={N}, 2-contact alternating single "
                                   f"support friction cone, fp64; u_opt0 all-gather over RCCL when n_gpus>1",
                       "horizon": N, "batch_per_gpu": B, "kernel": eng.kernel_name(), "streams": S, "longest_first_hint": not args.no_sched_hint,
                       "admm_mean_iters": mean_iters, "solved_frac": solved_frac,
                       "eps_abs": eng.cfg.eps_abs, "eps_rel": eng.cfg.eps_rel, "rho_restart_iter": int(eng.cfg.rho_restart_iter)},
            "roofline": {"bound": "mfma", "achieved": achieved_tf, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tf / PEAK_FP64_TFLOPS, "traffic": traffic,
                         "traffic_unit": "bytes/launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/*_pmc_summary.json)",
                         "algorithmic_bytes_per_launch": algorithmic_bytes(N) * B,
                         "kernel": eng.kernel_name(), "kernel_ms": kernel_ms,
                         "kernel_ms_note": ("one solve at a time (5 isolated solves after warm-up, HIP events on the launch stream)"
                                            + ("; a solve = the set-up kernel + the ADMM kernel back to back, the roofline is taken over the pair" if eng.kernel_name().startswith("split_") else "")
                                            + "; the timed region overlaps consecutive steps on %d streams" % S),
                         "algorithmic_flops_per_qp": algorithmic_flops(N, mean_iters),
                         "hbm_algorithmic_GBps": hbm_gbs, "hbm_frac_of_8TBps": hbm_gbs / 8000.0},
        }
        if parts is not None:   # per-kernel view: W(N, 0) belongs to the set-up kernel, the K (2 n^2 + 10 m) term to the ADMM kernel
            f_setup = algorithmic_flops(N, 0.0) * B
            f_admm = flops_launch - f_setup
            out["roofline"]["kernels"] = [
                {"kernel": "srbdqp_compact_kernel<%d,2,true> (set-up: linearise, condense, H, factor, K^-1)" % N, "ms": parts[0],
                 "achieved": f_setup / (parts[0] * 1e-3) / 1e12, "frac": f_setup / (parts[0] * 1e-3) / 1e12 / PEAK_FP64_TFLOPS},
                {"kernel": "srbdqp_admm_kernel<%d,2> (ADMM + roll-out, one wave per QP; its span is the 250-iteration stragglers)" % N, "ms": parts[1],
                 "achieved": f_admm / (parts[1] * 1e-3) / 1e12, "frac": f_admm / (parts[1] * 1e-3) / 1e12 / PEAK_FP64_TFLOPS}]
        if world == 1 and not args.no_latency:
            out["latency_batch1"] = latency_batch1(orc)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(orc, x0, xr, ft, ct)
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


def latency_batch1(orc, calls=2000):
    """p50/p99 of single-QP calls through the Python MPC.update() path (ctypes + H2D + kernel + D2H)."""
    from g1_locomotion_amd import MPC
    x0, xr, ft, ct = orc.synthetic_batch(64, HORIZON, seed=99, schedule="single")
    out = {}
    for warm in (False, True):
        mpc = MPC(dt=0.04, horizon=HORIZON, warm_start=warm)
        mpc.init_matrices()
        ts = []
        for i in range(calls + 50):
            b = i % 64
            mpc.x_ref_hor[:] = xr[b]
            t = time.perf_counter()
            mpc.update(list(ct[b]), list(ft[b]), xr[b][:, 3:6], x_current=x0[b].reshape(13, 1), one_rollout=True)
            ts.append(time.perf_counter() - t)
        ts = np.array(ts[50:]) * 1e6
        out["warm" if warm else "cold"] = {"p50_us": float(np.percentile(ts, 50)), "p99_us": float(np.percentile(ts, 99))}
        mpc.close()
    # the C-ABI call alone (inputs already in the staging arrays), at the default tolerance and at OSQP's default 1e-3
    from g1_locomotion_amd import BatchMPC
    for name, kw in (("c_abi", {}), ("c_abi_eps1e-3", {"eps_abs": 1e-3, "eps_rel": 1e-3})):
        with BatchMPC(horizon=HORIZON, **kw) as eng:
            st = eng.stage()
            ts = []
            for i in range(calls + 50):
                b = i % 64
                st["x0"][0] = x0[b]; st["x_ref"][0] = xr[b]; st["foot"][0] = ft[b]; st["contact"][0] = ct[b]
                t = time.perf_counter()
                eng.solve_staged(1, want_x=True)
                ts.append(time.perf_counter() - t)
            ts = np.array(ts[50:]) * 1e6
            out[name] = {"p50_us": float(np.percentile(ts, 50)), "p99_us": float(np.percentile(ts, 99))}
    out["note"] = ("cold / warm: MPC.update() from zero / from the previous call's shifted plan and duals; c_abi: "
                   "srbdqp_solve_staged_f64(B=1) alone; eps1e-3: OSQP's default tolerance instead of 1e-6")
    return out


def _cpu_share():
    """Host threads this process may really run: the affinity mask capped by the cgroup CPU quota (a 1-GPU box exposes
    every core of the node in the mask but grants only its share of them)."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                      # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = int(f.read()), int(g.read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(orc, x0, xr, ft, ct):
    """oracle/srbd_oracle.c on this node's host cores, bounded sample of the same batch."""
    import c_oracle
    p = orc.SrbdParams()
    cores = _cpu_share()
    S1 = min(2048, x0.shape[0])
    t = time.perf_counter()
    c_oracle.solve_batch(p, x0[:S1], xr[:S1], ft[:S1], ct[:S1], nthreads=1)
    t1 = time.perf_counter() - t
    Sall, reps = x0.shape[0], 24            # the whole rank-0 batch, repeated: ~15 core-seconds of CPU work in total
    c_oracle.solve_batch(p, x0, xr, ft, ct, nthreads=cores)     # warm the thread pool / page in
    t = time.perf_counter()
    for _ in range(reps):
        c_oracle.solve_batch(p, x0, xr, ft, ct, nthreads=cores)
    tall = (time.perf_counter() - t) / reps
    # the NumPy oracle (the "Python path" stand-in of SURVEY 8d) on a handful of QPs, and OSQP itself if this box has it
    t = time.perf_counter()
    for b in range(16):
        orc.update(p, x0[b], xr[b], ft[b], ct[b])
    t_np = (time.perf_counter() - t) / 16
    osqp_rate = None
    try:
        import osqp                                           # not installed in the build image; probed, never assumed
        import scipy.sparse as sp
        ts = []
        for b in range(16):
            qp = orc.build_qp(p, x0[b], xr[b], ft[b], ct[b])
            red, _, _ = orc.presolve(qp, ct[b])
            m = osqp.OSQP()
            m.setup(P=sp.csc_matrix(np.triu(red["P"])), q=red["q"], A=sp.csc_matrix(red["A"]), l=red["l"], u=red["u"], verbose=False,
                    eps_abs=p.eps_abs, eps_rel=p.eps_rel, max_iter=4000, polish=False)
            t = time.perf_counter(); m.solve(); ts.append(time.perf_counter() - t)
        osqp_rate = 1.0 / float(np.median(ts))
    except Exception:
        osqp_rate = None
    return {"value": Sall / tall, "unit": "QP/s", "cores": cores, "kind": "port",
            "numpy_oracle_qp_per_s": 1.0 / t_np, "osqp_qp_per_s": osqp_rate,
            "sample": f"the {Sall} QPs of the rank-0 batch x {reps} repetitions on {cores} threads = this box's CPU quota (cgroup cpu.max; the affinity mask shows {len(os.sched_getaffinity(0))}) (plain-C port oracle/srbd_oracle.c of the same algorithm incl. presolve, gcc -O3 -mavx2)",
            "single_thread_value": S1 / t1, "single_thread_sample": f"first {S1} QPs, 1 thread",
            "single_thread_p50_us": 1e6 * t1 / S1}


if __name__ == "__main__":
    main()
