#!/usr/bin/env python3
"""bench.py -- QP solves/sec of the SRBD convex-MPC hot path on MI355X (BASELINE.json's metric).

A "step" = one pass of the hot path (linearise -> condense -> H,g,cone rows -> factor -> ADMM -> rollout) over one
batch of synthetic QPs whose inputs are already resident in HBM.  Consecutive steps rotate over DISTINCT device batches
(different seeds), as the control steps of a fleet would differ.

  --config 1 (default at --gpus 1; BASELINE.json configs[1], the config the metric is quoted on):
        B = 4096 QPs per GPU, N = 10, 2-contact alternating single support, fp64
  --config 2 (configs[2]): B = 65536 per GPU, N = 20, 4-contact double support, fp32 buffers and iterations
        (fp64 assembly on chip, T factored in fp32 MFMA tiles; srbdqp_solve_batch_device_f32)
  --config 3 (default at --gpus N > 1; configs[3]): B = 65536 per GPU (524,288 over 8 GPUs), N = 10, 2-contact, fp64,
        u_opt0 all-gathered over RCCL every step
  --config 4 (configs[4]): ragged fleet, 16,384 QPs per GPU with N in {8, 12, 16, 24} drawn per QP and per-QP mixed-gait
        contact schedules, through srbdqp_solve_ragged_device_f64 (bucketed launch)

The one JSON line of the default run (`python bench.py`) carries, beside the configs[1] headline:
  value             2 HIP streams, natural QP order, tails deferred (SRBDQP_FLAG_DEFER_TAIL: a QP that reaches a rho-restart mark
                    unconverged -- 4 % of a batch -- rides in the next solve on its stream instead of holding its own launch up;
                    srbdqp_flush() INSIDE the timed region completes the last ones, so all K batches are solved when the clock stops)
  value_plain       1 stream, no hint, tails deferred: strictly serial steps in natural QP order
  value_stale_hint  2 streams, tails deferred, a longest-first hint taken from a DIFFERENT batch (a wrong hint)
  in_place          round 3's modes for comparison: every solve complete at its own launch (rho restart in place): value = 2 streams +
                    the hint from the SAME batch's previous solve, value_plain, value_stale_hint
  roofline          frac = flops the kernel really ISSUES (rocprofv3 PMC, profiles/*_pmc_summary.json of the same kernel)
                    over the live kernel time / the dense peak of the dtype; frac_algorithmic = SURVEY 8(d)'s W(N, K) of the
                    dense 12N-variable path over the same time (a throughput yardstick: the kernels execute far fewer flops)
  latency_batch1    single-QP calls, incl. the reference's own call pattern (full double support, run_simulation.py:100-106)
  also              short legs of configs[2] and configs[4] with their own roofline
  cpu_baseline      oracle/srbd_oracle.c (a plain-C port of the same algorithm; the reference's implementation is an
                    absent submodule) on this node's host cores, bounded sample of the same workload

With N > 1 ranks each rank owns its own batch (weak scaling, no data-path collective) and the first-step contact forces
u_opt0 are all-gathered over RCCL/xGMI every step, as north_star specifies; the same run also times the steps without
the collective (`value_without_allgather`; --no-allgather makes that the headline).

    python bench.py [--gpus N --steps K --warmup W]          # N > 1 without a launcher: starts its own N ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.
"""
import argparse
import glob
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
RAGGED_HORIZONS = (8, 12, 16, 24)

CONFIGS = {
    1: dict(horizon=10, batch=4096, schedule="single", f32=False, maxs=2, steps=500,
            workload="configs[1]: batch={B}/GPU random SRBD states, N=10, 2-contact alternating single support friction cone, fp64"),
    2: dict(horizon=20, batch=65536, schedule="double", f32=True, maxs=4, steps=40,
            workload="configs[2]: batch={B}/GPU random SRBD states, N=20, 4-contact double support, fp32 buffers + iterations (fp64 assembly; factorisation in fp32 MFMA tiles + one fp64 refinement step)"),
    3: dict(horizon=10, batch=65536, schedule="single", f32=False, maxs=2, steps=100,
            workload="configs[3]: batch={B}/GPU (524,288 over 8 GPUs) random SRBD states sharded over the ranks, N=10, 2-contact alternating single support, fp64"),
    4: dict(horizon=0, batch=16384, schedule="mixed", f32=False, maxs=4, steps=10,
            workload="configs[4]: ragged fleet of {B} QPs/GPU, horizon N in {{8,12,16,24}} drawn uniformly per QP, per-QP mixed-gait contact schedules, packed step-major arrays, bucketed launch (srbdqp_solve_ragged_device_f64), fp64"),
}


def algorithmic_flops(N: int, K: float) -> float:
    """W(N, K) flops per QP."""
    n, s, m = 12 * N, 13 * N, 20 * N
    return (500 * N + 2 * 13 ** 3 * N + N * (N + 1) / 2 * 2 * 13 * 13 * 12 + s * n * n
            + (2 * s * 13 + 2 * s * n) + n ** 3 / 3 + K * (2 * n * n + 10 * m))


def useful_flops(N: int, K: float, stance, wrench: bool = False) -> float:
    """W_eff: the minimal flop count of the algorithm the kernels really run on ONE QP -- the presolved (swing variables dropped)
    closed-form path, or its wrench-reduced form (general kernel) -- as opposed to SURVEY 8(d)'s W(N, K) of the dense 12N-variable
    path (`algorithmic_flops`) and to the flops the kernels ISSUE (PMC; masked lanes, padded 16x16 tiles, both triangles, redundant
    per-lane 4x4 / 6x6 inversions included).  `stance` = stance contacts per horizon step (length N, or one int for every step).

    dense presolved path (one-wave / 4-wave kernels), n = 3 sum(c), m = 5 sum(c):
        500 N [linearise] + 54 n [one 6+6+2-vector table row per variable] + 18 n(n+1)/2 [rank-6 entry: 6 + 3 multiply-adds]
        + (2 13 13 N + 12 n) [gradient: free response + one 6-vector product per variable] + n^3/3 [Cholesky] + n^3/3 [L^-1]
        + n^3/3 [K^-1 = W'W, one triangle] + K (2 n^2 + 10 m) [ADMM: K^-1 rhs + rows] + (2 13 13 N + 26 n) [roll-out]
    wrench-reduced path (general kernel): a step with c >= 3 stance contacts keeps g = 6 wrench coordinates, else g = 3c force
        variables; n_g = sum(g); T = S + E^-1 is n_g x n_g.  Per step with c >= 3: E = Y D^-1 Y' (36 c, symmetric half), its 6x6
        inverse (144), V = E^-1 Y D^-1 (216 c), Bd = D^-1 - D^-1 Y' V (108 c^2).  T: 16 n_g + 16 n_g(n_g+1)/2 [6 + 2 multiply-adds per
        entry], n_g^3 [factor, inverse, product], x_q = -K^-1 q once (one linear solve), and per iteration
        2 n_g^2 [T^-1 v] + per wrench step (36 c [V w] + 36 c [V' t] + 18 c^2 [Bd w]) + 10 m."""
    c = [int(stance)] * N if np.isscalar(stance) else [int(v) for v in stance]
    assert len(c) == N
    n, m = 3 * sum(c), 5 * sum(c)
    fixed = 500 * N + (2 * 13 * 13 * N + 12 * n) + (2 * 13 * 13 * N + 26 * n)
    if not wrench:
        return fixed + 54 * n + 18 * n * (n + 1) / 2 + n ** 3 + K * (2 * n * n + 10 * m)
    g = [6 if ci >= 3 else 3 * ci for ci in c]
    ng = sum(g)
    step_setup = sum(36 * ci + 144 + 216 * ci + 108 * ci * ci for ci in c if ci >= 3)
    solve = 2 * ng * ng + sum(72 * ci + 18 * ci * ci for ci in c if ci >= 3)
    return fixed + step_setup + 16 * ng + 16 * ng * (ng + 1) / 2 + ng ** 3 + solve + K * (solve + 10 * m)


def useful_flops_batch(N: int, contact, iters, wrench: bool) -> float:
    """sum of useful_flops over a batch: contact (B, N, 4) flags, iters (B,) iteration counts (W_eff is affine in K: grouped by pattern)"""
    c = np.asarray(contact).reshape(-1, N, 4).sum(axis=2).astype(np.int64)
    it = np.asarray(iters, dtype=np.float64).reshape(-1)
    pats, inv = np.unique(c, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    tot = 0.0
    for k, pat in enumerate(pats):
        sel = inv == k
        a = useful_flops(N, 0.0, pat, wrench)
        b = useful_flops(N, 1.0, pat, wrench) - a
        tot += a * sel.sum() + b * it[sel].sum()
    return tot


def algorithmic_bytes(N: int, esz: int = 8) -> int:
    """HBM bytes per QP (inputs + outputs): esz bytes per scalar, contact flags counted at esz as
    the survey does (4,536 B at N = 10, fp64)."""
    return (13 + 13 * N + 12 * N + 4 * N) * esz + (12 * N + 13 * (N + 1)) * esz + 8


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default: 500 / 40 / 100 / 10 for config 1 / 2 / 3 / 4)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=0,
                    help="default: 1 on one GPU, 3 (65,536 QPs per GPU, N = 10) on several")
    ap.add_argument("--batch", type=int, default=0, help="QPs per GPU per step (default: the config's)")
    ap.add_argument("--kernel", choices=["auto", "compact", "split", "wave", "wrench"], default="auto",
                    help="auto = the fastest parity-green kernel for the config")
    ap.add_argument("--streams", type=int, default=2,
                    help="consecutive steps alternate over this many HIP streams, so the straggler tail of one batch "
                         "(QPs that need many ADMM iterations) overlaps the bulk of the next; 1 = strictly serial steps")
    ap.add_argument("--in-place", action="store_true",
                    help="configs[1]: round 3's headline mode -- every solve complete at its own launch (rho restart in place) + the "
                         "longest-first hint; default: tails deferred to the next solve on the stream (SRBDQP_FLAG_DEFER_TAIL), no hint")
    ap.add_argument("--no-sched-hint", action="store_true",
                    help="do not feed a batch's previous iteration counts back as the longest-first dispatch hint")
    ap.add_argument("--no-allgather", action="store_true", help="n_gpus > 1: leave the u_opt0 all-gather out of the headline value")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and run the per-step u_opt0 all-gather even at world size 1: the collective's code path "
                         "on the one GPU at hand (rendezvous on 127.0.0.1 inside this process; no launcher, no re-exec)")
    ap.add_argument("--same-batch", action="store_true", help="every step solves the same batch (round-1 behaviour; A/B)")
    ap.add_argument("--max-iter", type=int, default=0, help="override srbdqp_config.max_iter (0 = library default)")
    ap.add_argument("--rho-restart", type=int, default=0, help="override srbdqp_config.rho_restart_iter (0 = library default)")
    ap.add_argument("--rho-restart-count", type=int, default=0, help="override srbdqp_config.rho_restart_count (0 = library default)")
    ap.add_argument("--rho", type=float, default=0.0, help="override srbdqp_config.rho (0 = library default)")
    ap.add_argument("--rho-fz-scale", type=float, default=0.0, help="override srbdqp_config.rho_fz_scale (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the value_plain / value_stale_hint variants and the configs[2] / configs[4] legs")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the configs[2] / configs[4] legs only")
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


def _auto_rho(N):
    return 0.7


def _resolved_restart(N, B, it, cnt, max_iter, kname):
    """what rho_restart_iter / rho_restart_count = 0 mean for this solve (srbdqp.h; srbdqp.hip restart_iter_of)"""
    wave = kname.startswith("wave_")
    auto = it == 0
    if auto:
        it = 55 if N <= 10 else {12: 70, 16: 80, 24: 100}.get(N, 125)
    if it <= 0 or it >= max_iter:
        return {"every": 0, "count": 0, "how": "off"}
    cnt = min(cnt if cnt > 0 else ((2 if N <= 12 else {16: 3, 24: 2}.get(N, 1)) if auto else 1), 3)     # (values above 3 mean 3: srbdqp.h)
    how = "one more launch over the same grid per pass (only the workgroups of the capped QPs do anything)"
    if wave:
        how = ("deferred: a QP at a mark hands itself to the next launch on its stream (srbdqp_flush completes the last ones)" if "defer" in kname
               else "in place, inside the one-wave kernel")
    return {"every": it, "count": cnt, "how": how}


def _auto_rho_fz(N):
    return 4.0


# ---------------------------------------------------------------------------------------------------------------------
# one configuration on this rank's GPU: device batches, outputs, engine, and step(i, mode)
# ---------------------------------------------------------------------------------------------------------------------
class Leg:
    """Homogeneous batches (configs[1], [2], [3]).  step(i, streams, hint): hint in {"own", "none", "stale"}."""

    def __init__(self, cid, B, args, rank, local_rank, dev, torch, nb, stub=False, max_streams=2, defer=False):
        from g1_locomotion_amd import synth
        cfg = CONFIGS[cid]
        self.cid, self.cfg, self.B, self.N, self.f32, self.nb, self.stub = cid, cfg, B, cfg["horizon"], cfg["f32"], nb, stub
        self.torch, self.dev, self.defer = torch, dev, defer
        N = self.N
        self.host_batches = [synth.synthetic_batch(B, N, seed=1000 * cid + 97 * j + rank, schedule=cfg["schedule"]) for j in range(nb)]
        self.tdt = tdt = torch.float32 if self.f32 else torch.float64
        self.d_in = [[torch.from_numpy(v).to(dev).to(tdt) if v.dtype == np.float64 else torch.from_numpy(v).to(dev) for v in hb]
                     for hb in self.host_batches]
        # output sets: steps in flight at the same time must not share outputs; with deferred tails a step's outputs are written until
        # rho_restart_count (2) further steps on its stream have run: twice the sets
        self.NO = NO = max(nb, max_streams) * (2 if defer else 1)
        self.d_u = [torch.zeros((B, N, 12), dtype=tdt, device=dev) for _ in range(NO)]
        self.d_x = [torch.zeros((B, N + 1, 13), dtype=tdt, device=dev) for _ in range(NO)]
        self.d_st = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(NO)]
        self.d_it = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(NO)]
        self.eng, self.kname = None, "stub"
        self.streams = [None] * max_streams
        if not stub:
            from g1_locomotion_amd import BatchMPC, _lib
            kid = {"auto": _lib.KERNEL_AUTO, "compact": _lib.KERNEL_COMPACT, "split": _lib.KERNEL_SPLIT, "wave": _lib.KERNEL_WAVE,
                   "wrench": _lib.KERNEL_WRENCH}[args.kernel]
            kw = {}
            if args.max_iter > 0: kw["max_iter"] = args.max_iter
            if args.rho_restart != 0: kw["rho_restart_iter"] = args.rho_restart
            if args.rho_restart_count != 0: kw["rho_restart_count"] = args.rho_restart_count
            if args.rho > 0: kw["rho"] = args.rho
            if args.rho_fz_scale > 0: kw["rho_fz_scale"] = args.rho_fz_scale
            if defer: kw["flags"] = _lib.FLAG_DEFER_TAIL
            self.eng = BatchMPC(horizon=N, device=local_rank, kernel=kid, max_contacts_per_step=cfg["maxs"], **kw)
            # non-default streams: the C-ABI treats a NULL stream as "the handle's own", and the HIP events that time a kernel
            # must sit on the stream the kernel is launched on
            self.streams = [torch.cuda.Stream(device=dev) for _ in range(max_streams)]

    def step(self, i, S=2, hint="own"):
        o, d = i % self.NO, self.d_in[i % self.nb]
        B, N = self.B, self.N
        if self.stub:
            self.d_u[o].copy_(d[2].reshape(B, N, 12))          # any deterministic function of the inputs
            self.d_st[o].fill_(1); self.d_it[o].fill_(5)
            return
        st = self.streams[i % S]
        if hint == "own":      # receding horizon: the iteration counts this batch produced the last time it was solved
            self.eng.set_schedule_hint(self.d_it[o].data_ptr(), B)
        elif hint == "stale":  # the counts ANOTHER batch produced: an uncorrelated, i.e. wrong, hint
            self.eng.set_schedule_hint(self.d_it[(o + 1) % self.NO].data_ptr(), B)
        else:
            self.eng.set_schedule_hint(0, 0)
        self.eng.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), self.d_u[o].data_ptr(),
                              x_out=self.d_x[o].data_ptr(), status=self.d_st[o].data_ptr(), iters=self.d_it[o].data_ptr(),
                              stream=st.cuda_stream, f32=self.f32)

    def flush(self):
        """deferred tails: enqueue what no later step has picked up (a no-op otherwise)"""
        if self.defer and self.eng is not None:
            self.eng.flush()

    def stats(self, solved_code):
        iters = self.torch.stack([t.cpu() for t in self.d_it[:self.nb]]).numpy()
        status = self.torch.stack([t.cpu() for t in self.d_st[:self.nb]]).numpy()
        self._iters = iters
        return float(iters.mean()), float((status == solved_code).mean()), iters

    def flops_bytes(self, mean_iters):
        esz = 4 if self.f32 else 8
        return algorithmic_flops(self.N, mean_iters) * self.B, algorithmic_bytes(self.N, esz) * self.B

    def useful_flops_launch(self):
        """W_eff per launch: mean over the distinct batches, every QP with its own contact pattern and iteration count (after stats())"""
        wrench = self.kname.startswith("wrench")
        return float(np.mean([useful_flops_batch(self.N, self.host_batches[j][3], self._iters[j], wrench) for j in range(self.nb)]))

    def close(self):
        if self.eng is not None:
            self.eng.close()


class RaggedLeg:
    """configs[4]: one ragged fleet per distinct batch, through srbdqp_solve_ragged_device_f64 (one call = one step)."""

    def __init__(self, B, args, rank, local_rank, dev, torch, nb):
        from g1_locomotion_amd import synth, RaggedMPC
        self.cid, self.cfg, self.B, self.N, self.f32, self.nb, self.stub = 4, CONFIGS[4], B, 0, False, nb, False
        self.torch, self.dev = torch, dev
        self.sets = []
        for j in range(nb):
            rng = np.random.default_rng(4000 + 97 * j + rank)
            Nq = rng.choice(RAGGED_HORIZONS, size=B).astype(np.int32)
            x0 = np.empty((B, 13)); rows = int(Nq.sum())
            xr = np.empty((rows, 13)); ft = np.empty((rows, 12)); ct = np.empty((rows, 4), np.uint8)
            off = np.concatenate([[0], np.cumsum(Nq)])
            for N in RAGGED_HORIZONS:
                idx = np.where(Nq == N)[0]
                a, b_, c, d = synth.synthetic_batch(len(idx), N, seed=4000 + 97 * j + N + rank, schedule="mixed")
                x0[idx] = a
                dst = (off[idx][:, None] + np.arange(N)[None, :]).reshape(-1)
                xr[dst] = b_.reshape(-1, 13); ft[dst] = c.reshape(-1, 12); ct[dst] = d.reshape(-1, 4)
            dd = [torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (x0, xr, ft, ct)]
            # two output copies per input set: with the restart passes on the tail streams a call's outputs are written until the library lets its
            # buffer set go (three calls later), so a call must not write where the call two before it may still be writing
            outs = [dict(u=torch.empty((rows, 12), dtype=torch.float64, device=dev), x=torch.empty((rows + B, 13), dtype=torch.float64, device=dev),
                         st=torch.zeros(B, dtype=torch.int32, device=dev), it=torch.zeros(B, dtype=torch.int32, device=dev)) for _ in range(2)]
            self.sets.append(dict(Nq=Nq, rows=rows, d=dd, outs=outs, off=off, ct=ct, **outs[0]))
        self.NO = nb
        from g1_locomotion_amd import _lib
        self.defer = not getattr(args, "in_place", False)
        # SRBDQP_FLAG_DEFER_TAIL: the restart passes of a bucket run on its tail stream beside the next call; flushed inside the timed steps
        self.eng = RaggedMPC(horizons=RAGGED_HORIZONS, device=local_rank, **({"flags": _lib.FLAG_DEFER_TAIL} if self.defer else {}))
        self.streams = [torch.cuda.Stream(device=dev)]
        self.kname = "ragged_wrench_f64_n8_n12_n16_n24"
        self.d_u = [s["u"] for s in self.sets]

    def flush(self):
        if self.defer:
            self.eng.flush(self.streams[0].cuda_stream)

    def step(self, i, S=1, hint="none"):
        s = self.sets[i % self.nb]
        d, o = s["d"], s["outs"][(i // self.nb) % 2]
        self.eng.solve_device(self.B, s["Nq"], d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), o["u"].data_ptr(),
                              x_out=o["x"].data_ptr(), status=o["st"].data_ptr(), iters=o["it"].data_ptr(), stream=self.streams[0].cuda_stream)

    def stats(self, solved_code):
        it = np.concatenate([s["it"].cpu().numpy() for s in self.sets]); st = np.concatenate([s["st"].cpu().numpy() for s in self.sets])
        self._it, self._Nq = it, np.concatenate([s["Nq"] for s in self.sets])
        return float(it.mean()), float((st == solved_code).mean()), it

    def flops_bytes(self, mean_iters):
        fl = by = 0.0
        for N in RAGGED_HORIZONS:
            m = self._Nq == N
            if m.any():
                fl += algorithmic_flops(N, float(self._it[m].mean())) * m.sum() / self.nb
                by += algorithmic_bytes(N, 8) * m.sum() / self.nb
        return fl, by

    def useful_flops_launch(self):
        tot = 0.0
        for s in self.sets:
            it = s["it"].cpu().numpy()
            for N in RAGGED_HORIZONS:
                idx = np.where(s["Nq"] == N)[0]
                if len(idx):
                    rows = (s["off"][idx][:, None] + np.arange(N)[None, :]).reshape(-1)
                    tot += useful_flops_batch(N, s["ct"][rows].reshape(len(idx), N, 4), it[idx], True)
        return tot / self.nb

    def close(self):
        self.eng.close()


def isolated_kernel_ms(leg, torch, n_iso, k0):
    """the step's kernels in isolation (HIP events on the launch stream, one solve at a time).  Deferred tails: a launch also runs the
    continuations of the launches before it, so the kernel's duration is the average over a run of consecutive launches on ONE stream
    (events around the whole run, the flush included), not of a launch alone on an empty list."""
    if leg.defer:
        n = max(n_iso, 20)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k = k0 + leg.NO * len(leg.streams)
        torch.cuda.synchronize(leg.dev)
        e0.record(leg.streams[0])
        for i in range(n):
            leg.step(k + i * len(leg.streams), S=len(leg.streams), hint="none")      # (a multiple of S: always stream 0)
        leg.flush()
        e1.record(leg.streams[0])
        torch.cuda.synchronize(leg.dev)
        return e0.elapsed_time(e1) / n, n
    iso = []
    for i in range(n_iso):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k = k0 + leg.NO * len(leg.streams) * (i + 1)                  # a multiple of S and NO: stream 0, output set 0
        e0.record(leg.streams[0]); leg.step(k, S=1, hint="own" if leg.cid != 4 else "none"); e1.record(leg.streams[0])
        torch.cuda.synchronize(leg.dev)
        iso.append(e0.elapsed_time(e1))
    return float(np.mean(iso)), len(iso)


def roofline(kname, flops_launch, abytes, B, kernel_ms, n_iso, peak, ms_per_step, S, N, defer=False, useful_launch=None):
    """`frac` = the flops the kernel really issues (PMC summary of the same kernel under profiles/) over the live kernel time
    and the dense peak; `frac_algorithmic` = SURVEY 8(d)'s W(N, K) over the same time."""
    alg_tf = flops_launch / (kernel_ms * 1e-3) / 1e12
    hbm_gbs = abytes / (kernel_ms * 1e-3) / 1e9
    r = {"bound": "unknown (no PMC summary for this kernel under profiles/)",
         "achieved": None, "peak": peak, "unit": "TFLOP/s", "frac": None,
         "frac_note": "frac / achieved = flops ISSUED by the step's kernels per launch (rocprofv3 PMC: SQ_INSTS_VALU_FLOPS_FP32/FP64 x 64 lanes + MFMA "
                      "x 2048, EXEC-masked lanes and padded tiles included; profiles/*_pmc_summary.json of the same kernel) over the live kernel "
                      "time below; frac_algorithmic = SURVEY 8(d)'s W(N, K) of the dense 12N-variable path over the same time -- a throughput "
                      "yardstick that the presolved, closed-form kernels beat by construction, not a utilisation",
         "achieved_algorithmic": alg_tf, "frac_algorithmic": alg_tf / peak,
         "frac_algorithmic_at_step_rate": flops_launch / (ms_per_step * 1e-3) / 1e12 / peak,
         "traffic": None,
         "traffic_unit": "HBM bytes per solve = per launch of this step's kernels (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes)",
         "algorithmic_bytes_per_launch": abytes, "kernel": kname, "kernel_ms": kernel_ms,
         "kernel_ms_note": ("average solve duration over a run of %d consecutive solves on ONE stream (HIP events on that stream around the whole run, "
                            "flush included): with deferred tails (SRBDQP_FLAG_DEFER_TAIL) the continuations / restart passes of a solve run beside the solves "
                            "behind it, so a solve alone on an idle device is not the kernel's steady state.  The timed region overlaps consecutive steps on %d stream(s)" % (n_iso, S))
                           if defer else
                           ("mean of %d isolated solves, one at a time, after the warm-up steps and before the timed region (so the GPU is at its "
                            "sustained clocks when the timed steps start); HIP events on the launch stream; with the rho restart on, a solve = the "
                            "first pass + the pass in which the capped QPs continue; an _f32 solve of >= 512 QPs = the fp32-tile and the fp64-tile "
                            "launch; a ragged solve = all its bucket launches.  The timed region overlaps consecutive steps on %d stream(s)" % (n_iso, S)),
         "algorithmic_flops_per_launch": flops_launch,
         "hbm_algorithmic_GBps": hbm_gbs, "hbm_frac_of_8TBps": hbm_gbs / 8000.0}
    if useful_launch is not None:
        # W_eff (useful_flops): the minimal flop count of the presolved / wrench-reduced algorithm the kernel runs, every QP with its own contact
        # pattern and iteration count -- neither the dense path's W (frac_algorithmic) nor what the kernel issues (frac)
        r["useful_flops_per_launch"] = useful_launch
        r["achieved_useful"] = useful_launch / (kernel_ms * 1e-3) / 1e12
        r["frac_useful"] = r["achieved_useful"] / peak
        r["frac_useful_at_step_rate"] = useful_launch / (ms_per_step * 1e-3) / 1e12 / peak
        r["frac_useful_note"] = ("W_eff = bench.useful_flops: linearise + closed-form assembly + Cholesky / L^-1 / W'W on the presolved (n_eff) or "
                                 "wrench-reduced (n_g) matrix + K x (2 n^2 + 10 m) + roll-out, one triangle, no padding, no masked lanes; over the kernel time "
                                 "(frac_useful) and over the driver-visible step time (frac_useful_at_step_rate)")
    try:
        best, scaled = None, False
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))):
            d = json.load(open(f))
            if d.get("bench_kernel_name") != kname:
                continue
            if d.get("batch_per_launch") == B or best is None or (best.get("batch_per_launch") != B and d.get("round", 0) >= best.get("round", 0)):
                if best is None or best.get("batch_per_launch") != B or d.get("batch_per_launch") == B:
                    best = d
                    best["_file"] = f
        if best is not None:
            k = B / float(best["batch_per_launch"])               # executed work per QP does not depend on the batch size
            scaled = abs(k - 1.0) > 1e-12
            r["traffic"] = k * best["hbm"].get("traffic_bytes_per_solve", best["hbm"]["traffic_bytes_per_launch"])
            ex = best.get("executed")
            if ex:
                fl = k * ex.get("flops_per_solve", ex["flops_per_launch"])
                r["executed_flops_per_launch"] = fl
                r["achieved"] = fl / (kernel_ms * 1e-3) / 1e12
                r["frac"] = r["achieved"] / peak
                # the same issued flops over the DRIVER-visible step time: with the steps of two streams overlapping, a step takes less
                # wall time than one launch lasts (profiles/r03_two_streams_timeline.txt), and an isolated launch of 4096 QPs lasts as
                # long as its slowest QP (250 iterations) -- the chip-level rate of the kernel is this one
                r["frac_at_step_rate"] = fl / (ms_per_step * 1e-3) / 1e12 / peak
                r["bound"] = ex["bound"]
                r["utilisation"] = ex["utilisation"]
            r["pmc_summary"] = os.path.relpath(best["_file"], ROOT) + (" (counted at %d QPs per launch, scaled per QP)" % best["batch_per_launch"] if scaled else "")
            # which library the counters were taken on (tools/pmc_summary.py records it) against the one loaded now
            r["pmc_summary_library_sha256"] = best.get("library_sha256")
            r["pmc_summary_git_rev"] = best.get("git_rev")
            r["pmc_summary_matches_loaded_library"] = (best.get("library_sha256") == library_sha256()) if best.get("library_sha256") else None
    except Exception as e:          # the summary is evidence, not a dependency
        r["traffic_error"] = repr(e)
    return r


_LIB_SHA = None


def library_sha256():
    """sha256 of the libsrbdqp.so this process loads (ties the line and the PMC summaries to a build)"""
    global _LIB_SHA
    if _LIB_SHA is None:
        import hashlib
        from g1_locomotion_amd import _lib
        try:
            _LIB_SHA = hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()
        except OSError:
            _LIB_SHA = "unreadable"
    return _LIB_SHA


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, argv))
    cid = args.config or (1 if (args.gpus == 1 and not args.force_dist) else 3)
    cfg = CONFIGS[cid]
    B = args.batch or cfg["batch"]
    steps = args.steps or cfg["steps"]

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
    if world > 1 or args.force_dist:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:          # --force-dist without a launcher: a one-rank rendezvous inside this process
            s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s_.getsockname()[1]); s_.close()
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist_mod.init_process_group(backend="nccl", device_id=dev)
        else:
            dist_mod.init_process_group(backend=args.backend)
        dist = dist_mod
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    nb = 1 if args.same_batch else NBATCH
    S = max(1, args.streams)
    SOLVED = 1
    if not stub:
        from g1_locomotion_amd import _lib
        SOLVED = _lib.SOLVED
    # deferred tails (round 4): configs[1] on one GPU, on the one-wave kernel with its automatic rho restart.  Not with several ranks: the all-gather of
    # u_opt0 behind every step needs that step's forces complete, and at 65,536 QPs per launch the tail is a tenth of the launch anyway.
    defer = (cid in (1, 2) and dist is None and not stub and not args.in_place and args.rho_restart >= 0 and not args.same_batch
             and (cid == 2 or (args.kernel in ("auto", "wave") and B >= 4096)))      # (configs[2]: the general kernel's restart pass on the library's tail stream)
    if cid == 4:
        leg = RaggedLeg(B, args, rank, local_rank, dev, torch, nb)
        S = 1
    else:
        leg = Leg(cid, B, args, rank, local_rank, dev, torch, nb, stub=stub, max_streams=S, defer=defer)
    N, f32 = leg.N, leg.f32
    NO = leg.NO
    hint = "none" if (args.no_sched_hint or cid == 4 or (defer and cid == 1)) else "own"      # (deferred tails on the one-wave kernel: nothing left for a hint to do)
    d_u0_all = [torch.empty((world * B, 12), dtype=torch.float32 if f32 else torch.float64, device=dev) for _ in range(NO)] if (dist is not None and cid != 4) else None

    def u0_of(o):
        return leg.d_u[o][:, 0, :].contiguous()

    def exchange(i):
        if dist is None or d_u0_all is None:
            return
        o = i % NO
        if stub:
            dist.all_gather_into_tensor(d_u0_all[o], u0_of(o))
        else:   # all-gather of u_opt0 on the step's own stream: overlaps the next step's kernel
            with torch.cuda.stream(leg.streams[i % S]):
                dist.all_gather_into_tensor(d_u0_all[o], u0_of(o))

    def sync():
        if not stub:
            torch.cuda.synchronize(dev)

    def timed(lg, k0, K, with_exchange, S_, hint_):
        if dist is not None:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for k in range(k0, k0 + K):
            lg.step(k, S=S_, hint=hint_)
            if with_exchange:
                exchange(k)
        lg.flush()                  # deferred tails: the continuations nothing has picked up yet run inside the timed region
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
        leg.step(i, S=S, hint="none" if i < NO else hint)
        exchange(i)
    leg.flush()
    sync()
    base = max(args.warmup, NO)
    base += (-base) % (S * NO)      # keep the (step -> stream, batch, output set) phase
    # ---- the dominant kernel in isolation: roofline numbers
    kernel_ms, n_iso = None, 0
    if not stub:
        kernel_ms, n_iso = isolated_kernel_ms(leg, torch, {1: 50, 2: 5, 3: 8, 4: 3}[cid], base)
        if cid != 4:
            leg.kname = leg.eng.kernel_name()
    kname = leg.kname
    # ---- timed region: exactly K steps (the headline), and the same K steps with / without the collective
    use_ag = dist is not None and not args.no_allgather and d_u0_all is not None
    elapsed = timed(leg, base, steps, use_ag, S, hint)
    elapsed_other = timed(leg, base, steps, not use_ag, S, hint) if (dist is not None and d_u0_all is not None) else None
    mean_iters, solved_frac, _ = leg.stats(SOLVED)
    if dist is not None and use_ag:            # the collective really moved this rank's forces
        o = (base + steps - 1) % NO
        mine = d_u0_all[o][rank * B:(rank + 1) * B]
        assert torch.equal(mine, leg.d_u[o][:, 0, :]), "all-gather result does not hold this rank's forces"
    # ---- the same K steps without the two things that shape the headline (configs[1] at one GPU)
    extra = {}
    def variants(lg):
        el_plain = timed(lg, base, steps, False, 1, "none")
        for i in range(lg.NO):                 # refresh every output set's counts, then hints taken from ANOTHER batch
            lg.step(base + i, S=S, hint="none")
        lg.flush()
        sync()
        el_stale = timed(lg, base, steps, False, S, "stale")
        return B * steps / el_plain, B * steps / el_stale

    if world == 1 and not stub and cid in (1, 3) and not args.no_also and (hint == "own" or defer):     # (configs[1] / the configs[3] shard on one GPU)
        extra["value_plain"], extra["value_stale_hint"] = variants(leg)
        if defer:
            extra["value_variants_note"] = ("value: %d streams, natural QP order, tails deferred to the next solve on the stream (SRBDQP_FLAG_DEFER_TAIL), flushed inside "
                                            "the timed region; value_plain: the same on 1 stream (strictly serial steps); value_stale_hint: %d streams + a "
                                            "longest-first hint from the counts of a DIFFERENT batch (a wrong hint only reorders work); in_place: the three "
                                            "with every solve complete at its own launch (rho restart in place, round 3's kernel), `value` there = %d streams + "
                                            "the hint from the SAME batch's previous solve" % (S, S, S))
            # round 3's modes on the same box, same batches: every solve complete at its own launch
            leg_ip = Leg(cid, B, args, rank, local_rank, dev, torch, nb, max_streams=S, defer=False)
            for i in range(2 * leg_ip.NO):
                leg_ip.step(i, S=S, hint="none" if i < leg_ip.NO else "own")
            sync()
            el_ip = timed(leg_ip, base, steps, False, S, "own")
            ip_it, ip_solved, _ = leg_ip.stats(SOLVED)
            ip_plain, ip_stale = variants(leg_ip)
            extra["in_place"] = {"value": B * steps / el_ip, "value_plain": ip_plain, "value_stale_hint": ip_stale, "kernel": leg_ip.eng.kernel_name(),
                                 "solved_frac": ip_solved, "admm_mean_iters": ip_it}
            leg_ip.close()
        else:
            extra["value_variants_note"] = ("value: %d streams + longest-first hint = the same batch's previous iteration counts; value_plain: 1 stream, no hint "
                                            "(strictly serial steps, natural QP order); value_stale_hint: %d streams, hint = the counts of a DIFFERENT "
                                            "batch (uncorrelated: a wrong hint only reorders work)" % (S, S))
        if args.rho_restart == 0 and kname.startswith("wave_") and B >= 4096:
            # what the rho restart in place costs and buys: the same steps with it switched off (plain fixed-rho ADMM, the algorithm of rounds 1-2)
            import copy
            args_off = copy.copy(args); args_off.rho_restart = -1
            leg_off = Leg(cid, B, args_off, rank, local_rank, dev, torch, nb, max_streams=S)
            for i in range(2 * leg_off.NO):
                leg_off.step(i, S=S, hint="none" if i < leg_off.NO else "own")
            sync()
            el_off = timed(leg_off, base, steps, False, S, "own")
            it_off, solved_off, _ = leg_off.stats(SOLVED)
            leg_off.close()
            extra["without_rho_restart"] = {"value": B * steps / el_off, "solved_frac": solved_off, "admm_mean_iters": it_off,
                                            "note": "the same steps with rho_restart_iter = -1 (plain fixed-rho ADMM; %d streams + the longest-first hint); value / solved_frac above are with the default (srbdqp.h rho_restart_iter)" % S}

    if rank == 0:
        total_qp = world * B * steps
        value = total_qp / elapsed
        peak = PEAK_FP32_TFLOPS if f32 else PEAK_FP64_TFLOPS
        out = {
            "metric": ("QP solves/sec, SRBD N=%d 12-state/12-input" % N) if cid != 4 else "QP solves/sec, SRBD mixed horizon N in {8,12,16,24} 12-state/12-input",
            "value": value, "unit": "QP/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / max(steps, 1), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if f32 else "f64", "data": "synthetic",
            "config": {"workload": cfg["workload"].format(B=B) + ("; u_opt0 all-gather over RCCL every step" if use_ag else "")
                                   + ("; tails deferred (SRBDQP_FLAG_DEFER_TAIL: the ~4 % of a batch that reach a rho-restart mark finish in the next launch on their "
                                      "stream; srbdqp_flush inside the timed region, every batch complete when the clock stops)" if defer else "")
                                   + (f"; steps rotate over {nb} distinct device batches" if nb > 1 else "; every step solves the same batch"),
                       "horizon": N if cid != 4 else list(RAGGED_HORIZONS), "batch_per_gpu": B, "kernel": kname, "streams": S,
                       "longest_first_hint": hint == "own", "deferred_tails": bool(defer), "world_size": world if dist is None else dist.get_world_size(),
                       "distinct_batches": nb, "admm_mean_iters": mean_iters, "solved_frac": solved_frac,
                       "allgather_in_value": bool(use_ag),
                       "collective": None if dist is None else {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "forced_at_world_size_1": bool(args.force_dist and world == 1),
                                                                "what": "all_gather_into_tensor of u_opt0 (B x 12) on the step's own HIP stream, every step"}},
        }
        out.update(extra)
        if not stub:
            out["library_sha256"] = library_sha256()
        if leg.eng is not None and cid != 4:
            c = leg.eng.cfg
            out["config"].update({"eps_abs": max(c.eps_abs, 2e-6) if f32 else c.eps_abs, "eps_rel": max(c.eps_rel, 2e-6) if f32 else c.eps_rel,
                                  "rho": _auto_rho(N) if c.rho == 0 else c.rho, "rho_fz_scale": _auto_rho_fz(N) if c.rho_fz_scale == 0 else c.rho_fz_scale,
                                  "max_iter": int(c.max_iter), "rho_restart_iter": int(c.rho_restart_iter), "rho_restart_count": int(c.rho_restart_count),
                                  "rho_restart_resolved": _resolved_restart(N, B, int(c.rho_restart_iter), int(c.rho_restart_count), int(c.max_iter), kname),
                                  "set_up_dtype": "f64" if not f32 else "f64 assembly, f32 tiles (f64 tiles for QPs with a step of <= 2 stance contacts)"})
        if elapsed_other is not None:
            key = "value_without_allgather" if use_ag else "value_with_allgather"
            out[key] = total_qp / elapsed_other
        if kernel_ms is not None:
            fl, by = leg.flops_bytes(mean_iters)
            out["roofline"] = roofline(kname, fl, by, B, kernel_ms, n_iso, peak, 1e3 * elapsed / max(steps, 1), S, N, defer=leg.defer,
                                       useful_launch=leg.useful_flops_launch())
        if world == 1 and not args.no_latency and not stub and cid == 1:
            from g1_locomotion_amd import synth
            out["latency_batch1"] = lat = latency_batch1(synth)
            # the p50 / p99 pairs again where the driver's record keeps them whole (north_star: single-QP p50 <= 50 us at batch 1)
            out["config"]["latency_batch1_us"] = {k: {"p50": round(v["p50_us"], 2), "p99": round(v["p99_us"], 2)}
                                                  for k, v in lat.items() if isinstance(v, dict) and "p50_us" in v}
            out["config"]["latency_batch1_us"]["calls_per_variant"] = lat["calls_per_variant"]
        host_batch0 = leg.host_batches[0] if cid != 4 else None
        leg.close()
        leg = None
        if world == 1 and not stub and cid == 1 and not args.no_also and not args.no_other_configs:
            out["also"] = also_legs(args, rank, local_rank, dev, torch, SOLVED)
        if world == 1 and not args.no_cpu_baseline and not stub and cid != 4:
            out["cpu_baseline"] = cpu_baseline(cid, N, host_batch0)
        print(json.dumps(out), flush=True)
    if leg is not None:
        leg.close()
    if dist is not None:
        dist.destroy_process_group()


def also_legs(args, rank, local_rank, dev, torch, SOLVED):
    """Short legs of the other single-GPU configurations, each with its own roofline (the headline stays configs[1])."""
    import copy
    args = copy.copy(args)          # the side legs always run the library defaults, whatever the headline was asked to A/B
    args.kernel, args.max_iter, args.rho_restart, args.rho, args.rho_fz_scale = "auto", 0, 0, 0.0, 0.0
    res = {}
    for cid, nb, steps, n_iso in ((2, 2, 10, 3), (4, 2, 5, 2)):
        try:
            cfg = CONFIGS[cid]
            B = cfg["batch"]
            S = 2 if cid == 2 else 1
            # configs[2]: the restart pass of a solve runs on the library's tail stream, beside the next solve (SRBDQP_FLAG_DEFER_TAIL; flushed inside the timed steps)
            leg = Leg(cid, B, args, rank, local_rank, dev, torch, nb, max_streams=S, defer=True) if cid != 4 else RaggedLeg(B, args, rank, local_rank, dev, torch, nb)
            hint = "own" if cid != 4 else "none"
            for i in range(2 * leg.NO):
                leg.step(i, S=S, hint="none" if i < leg.NO else hint)
            leg.flush()
            torch.cuda.synchronize(dev)
            base = 2 * leg.NO
            base += (-base) % (S * leg.NO)
            kernel_ms, n = isolated_kernel_ms(leg, torch, n_iso, base)
            if cid != 4:
                leg.kname = leg.eng.kernel_name()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for k in range(base, base + steps):
                leg.step(k, S=S, hint=hint)
            leg.flush()
            torch.cuda.synchronize(dev)
            el = time.perf_counter() - t0
            mean_iters, solved_frac, _ = leg.stats(SOLVED)
            fl, by = leg.flops_bytes(mean_iters)
            peak = PEAK_FP32_TFLOPS if leg.f32 else PEAK_FP64_TFLOPS
            res["configs[%d]" % cid] = {
                "workload": cfg["workload"].format(B=B) + f"; {steps} timed steps rotating over {nb} distinct device batches, {S} stream(s)"
                            + ("; restart passes on the library's tail stream beside the next solve, flushed inside the timed steps" if leg.defer else ""),
                "value": B * steps / el, "unit": "QP/s", "ms_per_step": 1e3 * el / steps, "steps": steps, "dtype": "f32" if leg.f32 else "f64",
                "kernel": leg.kname, "admm_mean_iters": mean_iters, "solved_frac": solved_frac,
                "roofline": roofline(leg.kname, fl, by, B, kernel_ms, n, peak, 1e3 * el / steps, S, leg.N, defer=leg.defer,
                                     useful_launch=leg.useful_flops_launch())}
            leg.close()
        except Exception as e:      # a failed side leg must not take the headline down with it; it is reported, not hidden
            res["configs[%d]" % cid] = {"error": repr(e)}
    return res


def latency_batch1(synth, calls=10000):
    """p50/p99 of single-QP calls: through the Python MPC.update() path (ctypes + staging + kernel) and through the C-ABI
    alone, for the synthetic single-support gait of configs[1] AND for the reference's own call pattern -- full double
    support on every step (g1_mujoco_sim/src/run_simulation.py:100-101 feeds [1, 1, 1, 1] per step, :106 calls update())."""
    from g1_locomotion_amd import MPC, BatchMPC
    out = {}
    sets = {"single": synth.synthetic_batch(64, HORIZON, seed=99, schedule="single"),
            "double": synth.synthetic_batch(64, HORIZON, seed=98, schedule="double")}

    def pct(ts):
        ts = np.array(ts[50:]) * 1e6
        return {"p50_us": float(np.percentile(ts, 50)), "p99_us": float(np.percentile(ts, 99))}

    def mpc_update(sched, n):
        x0, xr, ft, ct = sets[sched]
        mpc = MPC(dt=0.04, horizon=HORIZON)
        mpc.init_matrices()
        ts, its = [], []
        for i in range(n + 50):
            b = i % 64
            mpc.x_ref_hor[:] = xr[b]
            # the caller's own statements (run_simulation.py:94-103) come before the call: per-step lists, the CoM horizon, the (13, 1) state
            contact_horizon, c_horizon, p_com_horizon, x_cur = list(ct[b]), list(ft[b]), xr[b][:, 3:6].copy(), x0[b].reshape(13, 1)
            t = time.perf_counter()
            mpc.update(contact_horizon, c_horizon, p_com_horizon, x_current=x_cur, one_rollout=True)
            ts.append(time.perf_counter() - t); its.append(mpc.iters)
        mpc.close()
        return dict(pct(ts), mean_iters=float(np.mean(its[50:])))

    def c_abi(sched, n, **kw):
        x0, xr, ft, ct = sets[sched]
        with BatchMPC(horizon=HORIZON, **kw) as eng:
            st = eng.stage()
            ts, its = [], []
            for i in range(n + 50):
                b = i % 64
                st["x0"][0] = x0[b]; st["x_ref"][0] = xr[b]; st["foot"][0] = ft[b]; st["contact"][0] = ct[b]
                t = time.perf_counter()
                eng.solve_staged(1, want_x=True)
                ts.append(time.perf_counter() - t); its.append(int(st["iters"][0]))
            return dict(pct(ts), mean_iters=float(np.mean(its[50:])), kernel=eng.kernel_name())

    def mpc_update_arrays(sched, n):
        """the same call with (N, 12) / (N, 4) arrays instead of the reference's per-step lists (what the two np.concatenate cost)"""
        x0, xr, ft, ct = sets[sched]
        mpc = MPC(dt=0.04, horizon=HORIZON)
        mpc.init_matrices()
        ts = []
        for i in range(n + 50):
            b = i % 64
            mpc.x_ref_hor[:] = xr[b]
            pc, x_cur = xr[b][:, 3:6], x0[b].reshape(13, 1)
            t = time.perf_counter()
            mpc.update(ct[b], ft[b], pc, x_current=x_cur, one_rollout=True)
            ts.append(time.perf_counter() - t)
        mpc.close()
        return pct(ts)

    def closed_loop(n):
        """a CORRELATED sequence: segments of 25 consecutive control steps, each starting from a synthetic state of the mixed gait (a large
        disturbance) and then receding -- every state the previous plan's prediction, the contact schedule shifted by one step; every solve from
        zero (MPC(warm_start=True) was removed in round 5: profiles/r05_warm_start_sweep.txt)"""
        L = 25
        segs = (n + 50 + L - 1) // L
        X0, XR, FT, CT = synth.synthetic_batch(segs, HORIZON, seed=77, schedule="mixed")
        mpc = MPC(dt=0.04, horizon=HORIZON, strict=False)
        mpc.init_matrices()
        ts, its, capped = [], [], 0
        for s in range(segs):
            x = X0[s].copy()
            mpc.x_ref_hor[:] = XR[s]
            c_h = list(FT[s])
            for j in range(L):
                ct_h, x_cur = list(np.roll(CT[s], -j, axis=0)), x.reshape(13, 1)
                t = time.perf_counter()
                u0, xo = mpc.update(ct_h, c_h, None, x_current=x_cur, one_rollout=True)
                ts.append(time.perf_counter() - t); its.append(mpc.iters); capped += mpc.status == 2
                x = xo[1].copy()
        mpc.close()
        return dict(pct(ts), mean_iters=float(np.mean(its[50:])), max_iter_rate=capped / len(ts))

    out["cold"] = mpc_update("single", calls)
    out["c_abi"] = c_abi("single", calls)
    out["c_abi_eps1e-3"] = c_abi("single", calls, eps_abs=1e-3, eps_rel=1e-3)
    out["c_abi_double_support"] = c_abi("double", calls)
    out["mpc_update_double_support"] = mpc_update("double", calls)
    out["mpc_update_double_support_arrays"] = mpc_update_arrays("double", calls)
    out["closed_loop_cold"] = closed_loop(calls // 2)
    # the two-phase call: the set-up (contact schedule, contact points, reference known beforehand) has run and finished;
    # timed = the second phase only, from "the measured state is in the staging array" to "the forces are there"
    x0, xr, ft, ct = sets["single"]
    with BatchMPC(horizon=HORIZON) as eng:
        st = eng.stage()
        ts = []
        for i in range(calls // 4 + 50):
            b = i % 64
            st["x0"][0] = x0[(b + 1) % 64]; st["x_ref"][0] = xr[b]; st["foot"][0] = ft[b]; st["contact"][0] = ct[b]   # a wrong prediction
            eng.prepare_staged(1)
            eng.synchronize()
            st["x0"][0] = x0[b]
            t = time.perf_counter()
            eng.solve_prepared(1, want_x=True)
            ts.append(time.perf_counter() - t)
        out["c_abi_prepared_phase2"] = pct(ts)
    out["calls_per_variant"] = calls
    out["note"] = ("p50 / p99 over %d calls each (closed loops: %d, phase 2: %d).  " % (calls, calls // 2, calls // 4) +
                   "cold: MPC.update() on the single-support gait of configs[1] (64 unrelated QPs "
                   "in rotation, every solve from zero); c_abi: srbdqp_solve_staged_f64(B=1) alone on that gait (everything between the inputs and the forces); eps1e-3: OSQP's default "
                   "tolerance instead of 1e-6; c_abi_double_support / mpc_update_double_support: the same two calls on the REFERENCE'S OWN call "
                   "pattern, all four contact points active on every step, per-step lists as the reference passes them (run_simulation.py:94-101,106); "
                   "..._arrays: (N, 12) / (N, 4) arrays instead of the lists; closed_loop_cold: segments of 25 "
                   "consecutive control steps, each from a synthetic mixed-gait state and then receding (every state the previous plan's prediction, the contact schedule "
                   "shifted by one step), every solve from zero (the shift-based warm start of rounds 1-4 cost 41 iterations against 30 and was removed: "
                   "profiles/r05_warm_start_sweep.txt); max_iter_rate = share of the solves that end at the iteration cap; c_abi_prepared_phase2: "
                   "srbdqp_solve_prepared_f64 alone after a finished srbdqp_prepare_staged_f64 -- a different mode of operation (the "
                   "factorisation ran before the state arrived), listed beside c_abi, not instead of it")
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
    r_iter, r_count = orc.default_restart(N)
    p = orc.params_for(N, rho_restart_iter=r_iter, rho_restart_count=r_count)    # as the engine runs the config by default
    cores = _cpu_share()
    # sized for ~10-30 s of CPU WORK (thread-seconds, not wall time: ~1 s of wall on the box's 16 granted threads): N = 10 -> 4096 QPs x 24 (0.15 ms per QP and
    # thread = 15 thread-seconds), N = 20 -> 1024 QPs x 2 (dense 240-variable factor: several ms per QP and thread)
    Sall, reps, S1 = (min(4096, x0.shape[0]), 24, min(2048, x0.shape[0])) if N <= 10 else (min(1024, x0.shape[0]), 2, 64)
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
    nnp = 16 if N <= 10 else 4
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
