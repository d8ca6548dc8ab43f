"""Batch-1 latency breakdown of the staged (zero-copy) path on the GPU box.

  python tools/latency_probe.py [calls]

Prints p50/p99 of (a) the bare C-ABI call srbdqp_solve_staged_f64(h, 1, ...) with inputs already staged, with the
host spinning on the kernel's completion word and with hipStreamSynchronize, (b) the HIP-event kernel time,
(c) the Python MPC.update() path, cold / warm, and on a closed-loop sequence (consecutive, similar QPs).
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]


def pct(ts):
    ts = np.asarray(ts) * 1e6
    return {"p50_us": round(float(np.percentile(ts, 50)), 2), "p99_us": round(float(np.percentile(ts, 99)), 2),
            "min_us": round(float(ts.min()), 2)}


def main():
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    import torch  # noqa: F401  (same HIP runtime as the tests)
    import srbd_oracle as orc
    from g1_locomotion_amd import BatchMPC, MPC, _lib
    x0, xr, ft, ct = orc.synthetic_batch(64, 10, seed=99, schedule="single")
    out = {}
    for name, kw in (("spin", {}), ("stream_sync", {"flags": _lib.FLAG_NO_SPIN}), ("spin_eps1e-3", {"eps_abs": 1e-3, "eps_rel": 1e-3})):
        eng = BatchMPC(horizon=10, **kw)
        st = eng.stage()
        ts, its = [], []
        for i in range(calls + 100):
            b = i % 64
            st["x0"][0] = x0[b]; st["x_ref"][0] = xr[b]; st["foot"][0] = ft[b]; st["contact"][0] = ct[b]
            t = time.perf_counter()
            eng.solve_staged(1, want_x=True)
            ts.append(time.perf_counter() - t)
            its.append(int(st["iters"][0]))
        out[f"c_abi_{name}"] = dict(pct(ts[100:]), mean_iters=float(np.mean(its[100:])))
        eng.close()
    # two-phase call: set-up done beforehand (from a wrong predicted state), timed = srbdqp_solve_prepared_f64 alone; and both
    # phases back to back (what the two-phase form costs when nothing is known beforehand)
    eng = BatchMPC(horizon=10)
    st = eng.stage()
    ts, tb, its = [], [], []
    for i in range(calls + 100):
        b = i % 64
        st["x0"][0] = x0[(b + 1) % 64]; st["x_ref"][0] = xr[b]; st["foot"][0] = ft[b]; st["contact"][0] = ct[b]
        eng.prepare_staged(1)
        eng.synchronize()
        st["x0"][0] = x0[b]
        t = time.perf_counter()
        eng.solve_prepared(1, want_x=True)
        ts.append(time.perf_counter() - t)
        its.append(int(st["iters"][0]))
        t = time.perf_counter()
        eng.prepare_staged(1)
        eng.solve_prepared(1, want_x=True)
        tb.append(time.perf_counter() - t)
    out["c_abi_prepared_phase2"] = dict(pct(ts[100:]), mean_iters=float(np.mean(its[100:])))
    out["c_abi_two_phases_back_to_back"] = pct(tb[100:])
    eng.close()
    # PCIe-inclusive batch rate: host (pageable NumPy) buffers in, host buffers out, B = 4096
    xb, xrb, ftb, ctb = orc.synthetic_batch(4096, 10, seed=1000, schedule="single")
    eng = BatchMPC(horizon=10, max_contacts_per_step=2)
    for _ in range(3):
        eng.solve(xb, xrb, ftb, ctb)
    t = time.perf_counter()
    for _ in range(10):
        eng.solve(xb, xrb, ftb, ctb)
    dt = (time.perf_counter() - t) / 10
    out["host_buffers_batch4096"] = {"ms_per_call": round(dt * 1e3, 3), "qp_per_s": round(4096 / dt)}
    eng.close()
    eng = BatchMPC(horizon=10, timing=True)
    st = eng.stage()
    ks = []
    for i in range(300):
        b = i % 64
        st["x0"][0] = x0[b]; st["x_ref"][0] = xr[b]; st["foot"][0] = ft[b]; st["contact"][0] = ct[b]
        eng.solve_staged(1, want_x=True)
        ks.append(eng.last_kernel_ms() * 1e-3)
    out["kernel_events"] = pct(ks[50:])
    eng.close()
    for warm in (False,):       # (MPC(warm_start=True) was removed in round 5: profiles/r05_warm_start_sweep.txt)
        mpc = MPC(dt=0.04, horizon=10)
        mpc.init_matrices()
        ts = []
        for i in range(calls + 100):
            b = i % 64
            mpc.x_ref_hor[:] = xr[b]
            t = time.perf_counter()
            mpc.update(list(ct[b]), list(ft[b]), xr[b][:, 3:6], x_current=x0[b].reshape(13, 1), one_rollout=True)
            ts.append(time.perf_counter() - t)
        out["mpc_update_" + ("warm" if warm else "cold")] = pct(ts[100:])
        mpc.close()
    # closed loop: consecutive QPs of one robot (standing with a push, then stepping in place)
    from srbd_plant import SrbdPlant
    from g1_locomotion_amd import msgs
    FEET = np.array([[0.0, 0.0645, 0.0], [0.17, 0.0645, 0.0], [0.0, -0.0645, 0.0], [0.17, -0.0645, 0.0]])
    COM = np.array([0.085, 0.0, 0.598])
    for warm in (False,):
        for standing in (True, False):
            mpc = MPC(dt=0.04, horizon=10)
            mpc.init_matrices()
            real_update = mpc.update
            ts, its = [], []

            def timed(*a, **k):
                t = time.perf_counter()
                r = real_update(*a, **k)
                ts.append(time.perf_counter() - t); its.append(mpc.iters)
                return r
            mpc.update = timed
            plant = SrbdPlant(orc.SrbdParams())
            node = msgs.MpcNode(mpc, msgs.AlternatingGait(dt=0.04, standing=standing), com_target=COM)
            x = np.zeros(13); x[3:6] = COM + np.array([0.01, -0.01, -0.01]); x[0] = 0.03; x[12] = -9.80665
            t_sim, u0 = 0.0, np.zeros(12)
            for k in range(200 if standing else 25):
                xo, u0, act, land = msgs.unpack_mpc_solution(node.step(msgs.make_srbd_current(x, FEET, u0, stamp=t_sim)))
                if standing and k % 40 == 5:
                    x[9:12] += np.array([0.1, 0.05, 0.0])
                for _ in range(10):
                    x = plant.step(x, FEET, u0, 0.004)
                t_sim += 0.04
            out[f"closed_loop_{'standing' if standing else 'stepping'}_{'warm' if warm else 'cold'}"] = dict(pct(ts[5:]), mean_iters=float(np.mean(its[5:])))
            mpc.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
