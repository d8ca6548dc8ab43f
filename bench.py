#!/usr/bin/env python3
"""bench.py -- QP solves/sec of the SRBD convex-MPC hot path on MI355X (BASELINE.json's metric).

A "step" = one pass of the hot path (linearise -> condense -> H,g,cone rows -> factor -> ADMM -> rollout) over one
batch of synthetic QPs whose inputs are already resident in HBM.  Consecutive steps rotate over 4 DISTINCT device batches
(different seeds), as the control steps of a fleet would differ; the longest-first dispatch hint of a step is the
iteration counts the SAME batch produced the last time it was solved.

  --config 1 (default; BASELINE.json configs[1], the config the metric is quoted on):
        B = 4096 QPs per GPU, N = 10, 2-contact alternating single support, fp64
  --config 2 (configs[2]): B = 65536 per GPU, N = 20, 4-contact double support, fp32 buffers and iterations
        (fp64 assembly on chip, T factored in fp32 MFMA tiles; srbdqp_solve_batch_device_f32)

With N > 1 ranks each rank owns its own batch (weak scaling, no data-path collective) and the first-step contact forces
u_opt0 are all-gathered over RCCL/xGMI every step, as north_star specifies; the same run also times the steps without
the collective (`value_without_allgather`; --no-allgather makes that the headline).

    python bench.py [--gpus N --steps K --warmup W]          # N > 1 without a launcher: starts its own N ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.
`roofline`: `frac` = ALGORITHMIC flops of SURVEY.md section 8(d), W(N, K) of the dense 12N-variable path with K = the
measured mean ADMM iteration count, over the live kernel time, against the dense peak of the iteration dtype -- a
throughput yardstick (the kernels execute far fewer flops: presolve, closed-form assembly).  `frac_executed` = the flops
the kernel really issues (rocprofv3 PMC: (FMA x 2 + ADD + MUL) x 64 lanes + MFMA x 2048, profiles/*_pmc_summary.json of the
same kernel and batch) over the same live time: the hardware fraction.  `bound` is what those counters say.
`cpu_baseline`: oracle/srbd_oracle.c (a plain-C port of the same algorithm; the reference's own implementation is an
absent submodule) timed on this node's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np

HORIZON = 10
BATCH_PER_GPU = 4096
PEAK_FP64_TFLOPS = 78.6          # MI355X dense fp64 (vector == MFMA rate), SURVEY.md section 8(d)
PEAK_FP32_TFLOPS = 157.3         # MI355X dense fp32 vector (= fp32 MFMA rate), MI355X_MICROARCH.md
NBATCH = 4                       # distinct device batches the timed steps rotate over

CONFIGS = {
    1: dict(horizon=10, batch=4096, schedule="single", f32=False, maxs=2,
            workload="configs[1]: batch={B}/GPU random SRBD states, N=10, 2-contact alternating single support friction cone, fp64"),
    2: dict(horizon=20, batch=65536, schedule="double", f32=True, maxs=4,
            workload="configs[2]: batch={B}/GPU random SRBD states, N=20, 4-contact double support, fp32 buffers + iterations (fp64 assembly; factorisation in fp32 MFMA tiles + one fp64 refinement step)"),
}


def algorithmic_flops(N: int, K: float) -> float:
    """W(N, K) flops per QP."""
    n, s, m = 12 * N, 13 * N, 20 * N
    return (500 * N + 2 * 13 ** 3 * N + N * (N + 1) / 2 * 2 * 13 * 13 * 12 + s * n * n
            + (2 * s * 13 + 2 * s * n) + n ** 3 / 3 + K * (2 * n * n + 10 * m))


def algorithmic_bytes(N: int, esz: int = 8) -> int:
    """HBM bytes per QP (inputs + outputs): esz bytes per scalar, contact flags counted at esz as
    the survey does (4,536 B at N = 10, fp64)."""
    return (13 + 13 * N + 12 * N + 4 * N) * esz + (12 * N + 13 * (N + 1)) * esz + 8


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default: 500 for config 1, 40 for config 2)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=1)
    ap.add_argument("--batch", type=int, default=0, help="QPs per GPU per step (default: the config's)")
    ap.add_argument("--kernel", choices=["auto", "compact", "split", "wave", "wrench"], default="auto",
                    help="auto = the fastest parity-green kernel for the config")
    ap.add_argument("--streams", type=int, default=2,
                    help="consecutive steps alternate over this many HIP streams, so the straggler tail of one batch "
                         "(QPs that need many ADMM iterations) overlaps the bulk of the next; 1 = strictly serial steps")
    ap.add_argument("--no-sched-hint", action="store_true",
                    help="do not feed a batch's previous iteration counts back as the longest-first dispatch hint")
    ap.add_argument("--no-allgather", action="store_true", help="n_gpus > 1: leave the u_opt0 all-gather out of the headline value")
    ap.add_argument("--same-batch", action="store_true", help="every step solves the same batch (round-1 behaviour; A/B)")
    ap.add_argument("--max-iter", type=int, default=0, help="override srbdqp_config.max_iter (0 = library default)")
    ap.add_argument("--rho-restart", type=int, default=0, help="override srbdqp_config.rho_restart_iter (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (tests: gloo)")
    ap.add_argument("--stub-solve", action="store_true", help="tests of the launcher path only: no GPU, the solve is a stub")
    return ap.parse_args(argv)


def self_launch(args, argv):
    """--gpus N > 1 without a launcher: start N ranks of this script (one process per GPU) BEFORE anything in this process
    touches the GPU, relay rank 0's JSON line, exit with the worst child's code.  Never re-execs a process."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        p.wait()
        rc = rc or p.returncode
    sys.stdout.write(out)
    sys.stdout.flush()
    return rc


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, argv))
    cfg = CONFIGS[args.config]
    N, f32 = cfg["horizon"], cfg["f32"]
    B = args.batch or cfg["batch"]
    steps = args.steps or (500 if args.config == 1 else 40)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    stub = args.stub_solve
    if not stub:
        assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback for the hot path)"
        torch.cuda.set_device(local_rank)
    dev = torch.device("cpu") if stub else torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist_mod.init_process_group(backend="nccl", device_id=dev)
        else:
            dist_mod.init_process_group(backend=args.backend)
        dist = dist_mod

    from g1_locomotion_amd import synth
    nb = 1 if args.same_batch else NBATCH
    host_batches = [synth.synthetic_batch(B, N, seed=1000 * args.config + 97 * j + rank, schedule=cfg["schedule"]) for j in range(nb)]
    tdt = torch.float32 if f32 else torch.float64
    d_in = [[torch.from_numpy(v).to(dev).to(tdt) if v.dtype == np.float64 else torch.from_numpy(v).to(dev) for v in hb] for hb in host_batches]
    S = max(1, args.streams)
    NO = max(nb, S)                 # output sets: steps in flight at the same time must not share outputs
    d_u = [torch.zeros((B, N, 12), dtype=tdt, device=dev) for _ in range(NO)]
    d_x = [torch.zeros((B, N + 1, 13), dtype=tdt, device=dev) for _ in range(NO)]
    d_st = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(NO)]
    d_it = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(NO)]
    d_u0_all = [torch.empty((world * B, 12), dtype=tdt, device=dev) for _ in range(NO)] if world > 1 else None

    if stub:
        eng, streams, kname = None, [None] * S, "stub"
        SOLVED = 1

        def step(i):
            o = i % NO
            d_u[o].copy_(d_in[i % nb][2].reshape(B, N, 12))          # any deterministic function of the inputs
            d_st[o].fill_(1); d_it[o].fill_(5)
    else:
        from g1_locomotion_amd import BatchMPC, _lib
        SOLVED = _lib.SOLVED
        kid = {"auto": _lib.KERNEL_AUTO, "compact": _lib.KERNEL_COMPACT, "split": _lib.KERNEL_SPLIT, "wave": _lib.KERNEL_WAVE,
               "wrench": _lib.KERNEL_WRENCH}[args.kernel]
        eng = BatchMPC(horizon=N, device=local_rank, kernel=kid, max_contacts_per_step=cfg["maxs"],
                       **({"max_iter": args.max_iter} if args.max_iter > 0 else {}),
                       **({"rho_restart_iter": args.rho_restart} if args.rho_restart != 0 else {}))
        # non-default streams: the C-ABI treats a NULL stream as "the handle's own", and the HIP events that time a kernel
        # must sit on the stream the kernel is launched on
        streams = [torch.cuda.Stream(device=dev) for _ in range(S)]

        def step(i):
            st, o, d = streams[i % S], i % NO, d_in[i % nb]
            if not args.no_sched_hint:   # receding horizon: the iteration counts this batch produced the last time it was solved
                eng.set_schedule_hint(d_it[o].data_ptr() if i >= NO else 0, B)
            eng.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d_u[o].data_ptr(),
                             x_out=d_x[o].data_ptr(), status=d_st[o].data_ptr(), iters=d_it[o].data_ptr(),
                             stream=st.cuda_stream, f32=f32)

    def exchange(i):
        if dist is None:
            return
        o = i % NO
        if stub:
            dist.all_gather_into_tensor(d_u0_all[o], d_u[o][:, 0, :].contiguous())
        else:   # all-gather of u_opt0 on the step's own stream: overlaps the next step's kernel
            with torch.cuda.stream(streams[i % S]):
                dist.all_gather_into_tensor(d_u0_all[o], d_u[o][:, 0, :].contiguous())

    def sync():
        if not stub:
            torch.cuda.synchronize(dev)

    def timed(k0, K, with_exchange):
        if dist is not None:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for k in range(k0, k0 + K):
            step(k)
            if with_exchange:
                exchange(k)
        sync()
        if dist is not None:
            dist.barrier()
        sync()
        el = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el

    sync()
    for i in range(max(args.warmup, NO)):
        step(i)
        exchange(i)
    sync()
    base = max(args.warmup, NO)
    base += (-base) % (S * NO)      # keep the (step -> stream, batch, output set) phase
    # ---- the dominant kernel in isolation (HIP events on its launch stream, one launch at a time): roofline numbers
    kernel_ms = None
    if not stub:
        iso = []
        n_iso = 50 if args.config == 1 else 5       # ~12 ms / ~80 ms of isolated solves: a stable average, and the GPU is at its
        for i in range(n_iso):                       # sustained clocks when the timed region starts (disclosed in kernel_ms_note)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            k = base + S * NO * (i + 1)                                # a multiple of S and NO: stream 0, output set 0
            e0.record(streams[0]); step(k); e1.record(streams[0])
            sync()
            iso.append(e0.elapsed_time(e1))
        kernel_ms = float(np.mean(iso))
        kname = eng.kernel_name()
    # ---- timed region: exactly K steps (the headline), and the same K steps with / without the collective
    use_ag = dist is not None and not args.no_allgather
    elapsed = timed(base, steps, use_ag)
    elapsed_other = timed(base, steps, not use_ag) if dist is not None else None

    iters = torch.stack([t.cpu() for t in d_it[:nb]]).numpy()
    status = torch.stack([t.cpu() for t in d_st[:nb]]).numpy()
    mean_iters = float(iters.mean())
    solved_frac = float((status == SOLVED).mean())
    if dist is not None and use_ag:            # the collective really moved this rank's forces
        o = (base + steps - 1) % NO
        mine = d_u0_all[o][rank * B:(rank + 1) * B]
        assert torch.equal(mine, d_u[o][:, 0, :]), "all-gather result does not hold this rank's forces"

    if rank == 0:
        total_qp = world * B * steps
        value = total_qp / elapsed
        peak = PEAK_FP32_TFLOPS if f32 else PEAK_FP64_TFLOPS
        esz = 4 if f32 else 8
        out = {
            "metric": "QP solves/sec, SRBD N=%d 12-state/12-input" % N,
            "value": value, "unit": "QP/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / max(steps, 1), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if f32 else "f64", "data": "synthetic",
            "config": {"workload": cfg["workload"].format(B=B) + ("; u_opt0 all-gather over RCCL every step" if use_ag else "")
                                   + (f"; steps rotate over {nb} distinct device batches" if nb > 1 else "; every step solves the same batch"),
                       "horizon": N, "batch_per_gpu": B, "kernel": kname, "streams": S, "longest_first_hint": not args.no_sched_hint,
                       "distinct_batches": nb, "admm_mean_iters": mean_iters, "solved_frac": solved_frac,
                       "allgather_in_value": bool(use_ag)},
        }
        if eng is not None:
            c = eng.cfg
            out["config"].update({"eps_abs": max(c.eps_abs, 2e-6) if f32 else c.eps_abs, "eps_rel": max(c.eps_rel, 2e-6) if f32 else c.eps_rel,
                                  "rho": _auto_rho(N) if c.rho == 0 else c.rho, "max_iter": int(c.max_iter),
                                  "rho_restart_iter": int(c.rho_restart_iter), "set_up_dtype": "f64" if not f32 else "f64 assembly, f32 tiles (f64 tiles for QPs with a step of <= 2 stance contacts)"})
        if elapsed_other is not None:
            key = "value_without_allgather" if use_ag else "value_with_allgather"
            out[key] = total_qp / elapsed_other
        if kernel_ms is not None:
            out["roofline"] = roofline(kname, N, B, mean_iters, kernel_ms, peak, esz, 1e3 * elapsed / max(steps, 1), S)
        if world == 1 and not args.no_latency and not stub and args.config == 1:
            out["latency_batch1"] = latency_batch1(synth)
        if world == 1 and not args.no_cpu_baseline and not stub:
            out["cpu_baseline"] = cpu_baseline(args.config, N, host_batches[0])
        print(json.dumps(out), flush=True)
    if eng is not None:
        eng.close()
    if dist is not None:
        dist.destroy_process_group()


def _auto_rho(N):
    return 1.0 if N <= 10 else (1.5 if N <= 16 else 2.0)


def roofline(kname, N, B, mean_iters, kernel_ms, peak, esz, ms_per_step, S):
    flops_launch = algorithmic_flops(N, mean_iters) * B
    achieved_tf = flops_launch / (kernel_ms * 1e-3) / 1e12
    abytes = algorithmic_bytes(N, esz) * B
    hbm_gbs = abytes / (kernel_ms * 1e-3) / 1e9
    r = {"bound": "unknown (no PMC summary for this kernel / batch under profiles/)",
         "achieved": achieved_tf, "peak": peak, "unit": "TFLOP/s", "frac": achieved_tf / peak,
         "frac_note": "ALGORITHMIC-equivalent: SURVEY 8(d)'s W(N, K) of the dense 12N-variable path over the live kernel time; "
                      "a throughput yardstick, not a utilisation -- see frac_executed",
         "frac_executed": None, "traffic": None,
         "traffic_unit": "bytes per solve = per launch of this step's kernels (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/*_pmc_summary.json)",
         "algorithmic_bytes_per_launch": abytes, "kernel": kname, "kernel_ms": kernel_ms,
         "kernel_ms_note": "one solve at a time (50 isolated solves at configs[1], 5 at configs[2], after the W warm-up steps and "
                           "before the timed region -- so the GPU is also at its sustained clocks when the K timed steps start; "
                           "HIP events on the launch stream; with the rho restart on, a solve = the first pass + the pass in which "
                           "the capped QPs continue; an _f32 solve of >= 512 QPs = the fp32-tile and the fp64-tile launch); the "
                           "timed region overlaps consecutive steps on %d streams" % S,
         "algorithmic_flops_per_qp": algorithmic_flops(N, mean_iters),
         "hbm_algorithmic_GBps": hbm_gbs, "hbm_frac_of_8TBps": hbm_gbs / 8000.0,
         # the same count over the DRIVER-visible step time (the streams overlap steps): a value > 1 here would say that the
         # counted work is not what the kernel executes -- which is why frac_executed exists
         "frac_at_step_rate": flops_launch / (ms_per_step * 1e-3) / 1e12 / peak}
    try:
        import glob
        best = None
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))):
            d = json.load(open(f))
            if d.get("bench_kernel_name") == kname and d.get("batch_per_launch") == B:
                best = d
                best["_file"] = f
        if best is not None:
            r["traffic"] = best["hbm"].get("traffic_bytes_per_solve", best["hbm"]["traffic_bytes_per_launch"])
            ex = best.get("executed")
            if ex:
                fl = ex.get("flops_per_solve", ex["flops_per_launch"])
                r["executed_flops_per_launch"] = fl
                r["frac_executed"] = fl / (kernel_ms * 1e-3) / 1e12 / peak
                r["frac_executed_note"] = ex["note"]
                r["bound"] = ex["bound"]
                r["utilisation"] = ex["utilisation"]
            r["pmc_summary"] = os.path.relpath(best["_file"], ROOT)
    except Exception as e:          # the summary is evidence, not a dependency
        r["traffic_error"] = repr(e)
    return r


def latency_batch1(synth, calls=2000):
    """p50/p99 of single-QP calls through the Python MPC.update() path (ctypes + H2D + kernel + D2H)."""
    from g1_locomotion_amd import MPC
    x0, xr, ft, ct = synth.synthetic_batch(64, HORIZON, seed=99, schedule="single")
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
    # the two-phase call: the set-up (contact schedule, contact points, reference known beforehand) has run and finished;
    # timed = the second phase only, from "the measured state is in the staging array" to "the forces are there"
    with BatchMPC(horizon=HORIZON) as eng:
        st = eng.stage()
        ts = []
        for i in range(calls + 50):
            b = i % 64
            st["x0"][0] = x0[(b + 1) % 64]; st["x_ref"][0] = xr[b]; st["foot"][0] = ft[b]; st["contact"][0] = ct[b]   # a wrong prediction
            eng.prepare_staged(1)
            eng.synchronize()
            st["x0"][0] = x0[b]
            t = time.perf_counter()
            eng.solve_prepared(1, want_x=True)
            ts.append(time.perf_counter() - t)
        ts = np.array(ts[50:]) * 1e6
        out["c_abi_prepared_phase2"] = {"p50_us": float(np.percentile(ts, 50)), "p99_us": float(np.percentile(ts, 99))}
    out["note"] = ("cold / warm: MPC.update() from zero / from the previous call's shifted plan and duals; c_abi: "
                   "srbdqp_solve_staged_f64(B=1) alone (THE single-QP latency: everything between the inputs and the forces); "
                   "eps1e-3: OSQP's default tolerance instead of 1e-6; c_abi_prepared_phase2: srbdqp_solve_prepared_f64 alone after a "
                   "finished srbdqp_prepare_staged_f64 -- a different mode of operation (the factorisation ran before the state "
                   "arrived), listed beside c_abi, not instead of it")
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


def cpu_baseline(config, N, batch):
    """oracle/srbd_oracle.c on this node's host cores, bounded sample of the same workload (the checker, used here as the
    CPU baseline leg only)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import srbd_oracle as orc
    import c_oracle
    x0, xr, ft, ct = batch
    p = orc.params_for(N, rho_restart_iter=0 if config == 1 else 125)    # as the engine runs the config by default
    cores = _cpu_share()
    # sized for ~10-30 s of CPU work: config 1 = the whole 4096-QP batch x 24 (0.15 ms per QP and thread), config 2 = 1024
    # QPs x 2 (dense 240-variable factor: several ms per QP and thread)
    Sall, reps, S1 = (x0.shape[0], 24, min(2048, x0.shape[0])) if config == 1 else (min(1024, x0.shape[0]), 2, 64)
    a = [v[:Sall] for v in (x0, xr, ft, ct)]
    t = time.perf_counter()
    c_oracle.solve_batch(p, x0[:S1], xr[:S1], ft[:S1], ct[:S1], nthreads=1)
    t1 = time.perf_counter() - t
    c_oracle.solve_batch(p, *a, nthreads=cores)                 # warm the thread pool / page in
    t = time.perf_counter()
    for _ in range(reps):
        c_oracle.solve_batch(p, *a, nthreads=cores)
    tall = (time.perf_counter() - t) / reps
    # the NumPy oracle (the "Python path" stand-in of SURVEY 8d) on a handful of QPs, and OSQP itself if this box has it
    nnp = 16 if config == 1 else 4
    t = time.perf_counter()
    for b in range(nnp):
        orc.update(p, x0[b], xr[b], ft[b], ct[b])
    t_np = (time.perf_counter() - t) / nnp
    osqp_rate = None
    try:
        import osqp                                           # not installed in the build image; probed, never assumed
        import scipy.sparse as sp
        ts = []
        for b in range(nnp):
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
            "sample": f"the first {Sall} QPs of rank 0's first batch x {reps} repetitions on {cores} threads = this box's CPU quota "
                      f"(cgroup cpu.max; the affinity mask shows {len(os.sched_getaffinity(0))}) (plain-C port oracle/srbd_oracle.c of the same "
                      f"algorithm on the presolved dense QP, fp64, gcc -O3 -mavx2)",
            "single_thread_value": S1 / t1, "single_thread_sample": f"first {S1} QPs, 1 thread",
            "single_thread_p50_us": 1e6 * t1 / S1}


if __name__ == "__main__":
    main()
