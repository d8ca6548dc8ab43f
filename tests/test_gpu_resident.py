"""GPU tests of the resident batch-1 solver (SRBDQP_FLAG_RESIDENT, csrc/srbdqp_resident.hpp).

The resident kernel runs the source of the launched 4-wave kernel (compact_qp), compiled a second time inside the
request loop: the bar is the same status and iteration count and forces / roll-out / duals equal to rounding
(TOL_SAME: 1e-6 N -- measured 3e-9; fused multiply-adds are formed differently in the two compilations) -- plus the life cycle:
idle time-out and restart, explicit stop, destroy while it runs, the switch between the two kernel instantiations
(<= 2 / <= 4 stance contacts per step), and parity with the oracle.
"""
import time

import numpy as np
import pytest

import srbd_oracle as orc

pytestmark = pytest.mark.gpu

TOL_TWIN_N = 2e-3
TOL_SAME = 1e-6      # resident vs launched kernel, forces [N], duals, states


def _same(a, b, msg=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and np.abs(a - b).max() <= TOL_SAME, (msg, float(np.abs(a - b).max()))


@pytest.fixture(scope="module")
def torch_first():
    import torch  # load torch's HIP runtime before libsrbdqp.so so both share one
    assert torch.cuda.is_available()
    return torch


def _fill(st, x0, xr, ft, ct, pc=None, wu=None, wy=None):
    st["x0"][0] = x0; st["x_ref"][0] = xr; st["foot"][0] = ft; st["contact"][0] = ct
    if pc is not None:
        st["pcom"][0] = pc
    if wu is not None:
        st["warm_u"][0] = np.asarray(wu).reshape(-1); st["warm_y"][0] = wy


def _snapshot(st):
    return {k: np.array(st[k][0], copy=True) for k in ("u", "x", "y", "status", "iters")}


def _requests(N, seed):
    """A control-loop-like stream of QPs: single / double / mixed support and a flight phase."""
    reqs = []
    for sched, cnt in (("single", 6), ("double", 3), ("mixed", 4), ("single", 3)):
        x0, xr, ft, ct = orc.synthetic_batch(cnt, N, seed=seed + len(reqs), schedule=sched)
        reqs += [(x0[b], xr[b], ft[b], ct[b]) for b in range(cnt)]
    x0, xr, ft, ct = reqs[0]
    reqs.insert(5, (x0, xr, ft, np.zeros_like(ct)))          # flight: iters 0, forces 0
    return reqs


@pytest.mark.parametrize("N", [10, 8, 4])
def test_resident_solver_agrees_with_the_launched_kernel(torch_first, built_lib, N):
    from g1_locomotion_amd import BatchMPC
    rng = np.random.default_rng(7)
    with BatchMPC(horizon=N) as plain, BatchMPC(horizon=N, resident=True) as res:
        sp, sr = plain.stage(), res.stage()
        prev = None
        for i, (x0, xr, ft, ct) in enumerate(_requests(N, 300 + N)):
            kw = dict(use_pcom=(i % 3 == 1), use_warm=(prev is not None and i % 4 == 2), want_x=(i % 5 != 4), want_y=(i % 2 == 0))
            pc = xr[:, 3:6] + rng.normal(0, 0.01, (N, 3)) if kw["use_pcom"] else None
            wu, wy = (prev["u"], prev["y"]) if kw["use_warm"] else (None, None)
            outs = []
            for eng, st in ((plain, sp), (res, sr)):
                for k in ("u", "x", "y"):
                    st[k][0] = -7.0                           # outputs a request does not ask for must stay untouched
                _fill(st, x0, xr, ft, ct, pc, wu, wy)
                eng.solve_staged(1, **kw)
                outs.append(_snapshot(st))
            a, b = outs
            for k in ("status", "iters"):
                np.testing.assert_array_equal(a[k], b[k], err_msg=f"request {i} field {k}")
            for k in ("u", "x", "y"):
                _same(a[k], b[k], f"request {i} field {k}")
            if not kw["want_y"]:
                assert np.all(b["y"] == -7.0)
            if not kw["want_x"]:
                assert np.all(b["x"] == -7.0)
            if kw["want_y"] and int(a["status"]) == 1:
                prev = a
        assert res.kernel_name().startswith("resident_compact_f64_n%d" % N)
        assert res.resident_running() and not plain.resident_running()


def test_resident_solver_matches_the_oracle(torch_first, built_lib):
    from g1_locomotion_amd import BatchMPC
    N = 10
    p = orc.SrbdParams()
    with BatchMPC(horizon=N, resident=True) as res:
        st = res.stage()
        for sched, seed in (("single", 41), ("double", 42), ("mixed", 43)):
            x0, xr, ft, ct = orc.synthetic_batch(4, N, seed=seed, schedule=sched)
            for b in range(4):
                ref = orc.update(p, x0[b], xr[b], ft[b], ct[b])
                _fill(st, x0[b], xr[b], ft[b], ct[b])
                res.solve_staged(1, want_x=True, want_y=True)
                assert int(st["status"][0]) == ref["status"]
                assert abs(int(st["iters"][0]) - ref["iters"]) <= p.check_every
                assert np.abs(st["u"][0].reshape(N, 12) - ref["u"]).max() <= TOL_TWIN_N
                assert np.abs(st["x"][0] - ref["x"]).max() <= 1e-6


def test_resident_idle_timeout_restart_stop_and_destroy(torch_first, built_lib):
    from g1_locomotion_amd import BatchMPC
    N = 10
    x0, xr, ft, ct = orc.synthetic_batch(3, N, seed=77, schedule="single")
    with BatchMPC(horizon=N) as plain:
        sp = plain.stage()
        want = []
        for b in range(3):
            _fill(sp, x0[b], xr[b], ft[b], ct[b])
            plain.solve_staged(1)
            want.append(np.array(sp["u"][0], copy=True))
    res = BatchMPC(horizon=N, resident=True, resident_idle_ms=5)
    st = res.stage()
    assert not res.resident_running()                          # started by the first request, not by create
    for rnd in range(3):                                       # time-out between the requests: each one restarts the kernel
        _fill(st, x0[rnd], xr[rnd], ft[rnd], ct[rnd])
        res.solve_staged(1)
        _same(st["u"][0], want[rnd])
        assert res.resident_running()
        time.sleep(0.05)
        assert not res.resident_running()
    # back-to-back requests keep one kernel alive well past the idle time (the clock restarts with every request)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.03:
        _fill(st, x0[1], xr[1], ft[1], ct[1])
        res.solve_staged(1)
        _same(st["u"][0], want[1])
    assert res.resident_running()
    res.resident_stop()
    assert not res.resident_running()
    res.resident_stop()                                        # idempotent
    _fill(st, x0[2], xr[2], ft[2], ct[2])
    res.solve_staged(1)                                        # starts again after an explicit stop
    _same(st["u"][0], want[2])
    res.close()
    # other work of the same handle runs beside the resident kernel: a batch on the launch stream (default idle time)
    res = BatchMPC(horizon=N, resident=True)
    st = res.stage()
    _fill(st, x0[0], xr[0], ft[0], ct[0])
    res.solve_staged(1)
    xb, xrb, ftb, ctb = orc.synthetic_batch(8, N, seed=78, schedule="single")
    out = res.solve(xb, xrb, ftb, ctb)
    assert np.all(out["status"] == 1)
    assert res.resident_running()
    _fill(st, x0[1], xr[1], ft[1], ct[1])
    res.solve_staged(1)
    _same(st["u"][0], want[1])
    t0 = time.perf_counter()
    res.close()                                                # destroy while the kernel polls: it is told to leave
    assert time.perf_counter() - t0 < 0.05


def test_resident_solver_with_the_rho_restart(torch_first, built_lib):
    """The capped first pass runs in the resident kernel, the rare second pass as ordinary launches."""
    from g1_locomotion_amd import BatchMPC
    N = 10
    x0, xr, ft, ct = orc.synthetic_batch(400, N, seed=1000, schedule="single")
    p = orc.SrbdParams(rho_restart_iter=100)
    refs = [orc.update(p, x0[b], xr[b], ft[b], ct[b]) for b in range(120)]
    hard = [b for b in range(120) if refs[b]["iters"] > 100][:2]
    easy = [b for b in range(120) if refs[b]["iters"] <= 40][:2]
    assert hard and easy
    with BatchMPC(horizon=N, resident=True, rho_restart_iter=100) as res, BatchMPC(horizon=N, rho_restart_iter=100) as plain:
        sr, sp = res.stage(), plain.stage()
        for b in hard + easy + hard:
            for eng, st in ((plain, sp), (res, sr)):
                _fill(st, x0[b], xr[b], ft[b], ct[b])
                eng.solve_staged(1, want_x=True, want_y=(b in easy))
            for k in ("status", "iters"):
                np.testing.assert_array_equal(sp[k][0], sr[k][0], err_msg=f"QP {b} field {k}")
            for k in ("u", "x"):
                _same(sp[k][0], sr[k][0], f"QP {b} field {k}")
            assert int(sr["status"][0]) == refs[b]["status"] and abs(int(sr["iters"][0]) - refs[b]["iters"]) <= p.check_every
            assert np.abs(sr["u"][0].reshape(N, 12) - refs[b]["u"]).max() <= 2 * TOL_TWIN_N


def test_mpc_update_through_the_resident_solver(torch_first, built_lib):
    from g1_locomotion_amd import MPC
    N = 10
    x0, xr, ft, ct = orc.synthetic_batch(6, N, seed=55, schedule="mixed")
    p = orc.SrbdParams()
    mpc = MPC(dt=0.04, warm_start=False, resident=True)
    mpc.init_matrices()
    for b in range(6):
        ref = orc.update(p, x0[b], xr[b], ft[b], ct[b])
        mpc.x_ref_hor[:] = xr[b]
        u0, x1 = mpc.update(list(ct[b]), list(ft[b]), xr[b][:, 3:6], x_current=x0[b].reshape(13, 1), one_rollout=True)
        assert np.abs(np.asarray(u0).flatten() - ref["u"][0]).max() <= TOL_TWIN_N
        assert np.abs(x1[1, :] - ref["x"][1]).max() <= 1e-6
    mpc.close()
