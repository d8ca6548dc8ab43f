"""The staged one-QP call through the library's own AQL queue (csrc/srbdqp_aql.hpp) against the same call through hipLaunchKernelGGL (SRBDQP_NO_AQL=1): the
same kernels behind two doors, so every output is bit-equal -- first passes, restart passes queued behind them in the queue, calls interleaved with batch solves
on HIP streams, and the handles the queue is not for (deferred tails, events, no spinning)."""
import os

import numpy as np
import pytest

import srbd_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_first():
    import torch  # load torch's HIP runtime before libsrbdqp.so so both share one
    assert torch.cuda.is_available()
    return torch


@pytest.fixture(scope="module")
def aql_ready(torch_first, built_lib):
    """The queue exists on this box (it needs host-visible device memory and an HSA queue of the process's own); where it does not the library says why and runs
    the same kernels through HIP -- the tests below that are about the queue are skipped with that reason instead of failing the suite."""
    from g1_locomotion_amd import BatchMPC
    os.environ.pop("SRBDQP_NO_AQL", None)
    x0, xr, ft, ct = orc.synthetic_batch(1, 10, seed=1, schedule="double")
    with BatchMPC(horizon=10) as eng:
        _one(eng, x0[0], xr[0], ft[0], ct[0], 10, False)
        path = eng.batch1_launch_path()
    if path != "aql":
        pytest.skip("no AQL queue on this box (%s)" % path)
    return True


def _pair(N, **kw):
    """one handle on the AQL queue, one on HIP (the environment variable is read at a handle's first staged one-QP call)"""
    from g1_locomotion_amd import BatchMPC
    os.environ.pop("SRBDQP_NO_AQL", None)
    a = BatchMPC(horizon=N, **kw)
    b = BatchMPC(horizon=N, **kw)
    return a, b


def _one(eng, x0, xr, ft, ct, N, no_aql):
    st = eng.stage()
    st["x0"][0] = x0; st["x_ref"][0] = xr; st["foot"][0] = ft.reshape(N, 12); st["contact"][0] = ct.reshape(N, 4)
    if no_aql:
        os.environ["SRBDQP_NO_AQL"] = "1"
    try:
        eng.solve_staged(1, want_x=True)
    finally:
        os.environ.pop("SRBDQP_NO_AQL", None)
    return st["u"][0].copy(), st["x"][0].copy(), int(st["status"][0]), int(st["iters"][0])


@pytest.mark.parametrize("N,schedule,suffix", [(10, "double", "wrench_f64_n10_lat"), (10, "single", "compact_f64_n10_s2_lat"), (10, "mixed", "wrench_f64_n10_lat"),
                                               (8, "double", "wrench_f64_n8_lat"), (4, "single", "compact_f64_n4_s2_lat")])
def test_aql_queue_equals_the_hip_launch(torch_first, built_lib, aql_ready, N, schedule, suffix):
    B = 48
    x0, xr, ft, ct = orc.synthetic_batch(B, N, seed=7700 + N, schedule=schedule)
    a, b = _pair(N)
    with a, b:
        batch = a.solve(x0, xr, ft, ct)                       # (a batch solve on the HIP stream before, between and after the queue's kernels)
        for q in range(B):
            ra = _one(a, x0[q], xr[q], ft[q], ct[q], N, False)
            rb = _one(b, x0[q], xr[q], ft[q], ct[q], N, True)
            assert a.kernel_name() == b.kernel_name() and a.kernel_name().endswith("_lat"), (a.kernel_name(), suffix)
            assert ra[2] == rb[2] and ra[3] == rb[3]
            assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1])
            assert ra[2] == int(batch["status"][q])
            if q % 16 == 7:
                again = a.solve(x0, xr, ft, ct)
                assert np.array_equal(again["u"], batch["u"])
        assert a.batch1_launch_path() == "aql", a.batch1_launch_path()
        assert b.batch1_launch_path().startswith("hip: SRBDQP_NO_AQL")


def test_restart_passes_queue_behind_the_first_pass(torch_first, built_lib, aql_ready):
    """QPs that pass the first restart mark: the second (and third) pass are further packets in the same queue, ordered by the barrier bit."""
    N, B = 10, 4096
    x0, xr, ft, ct = orc.synthetic_batch(B, N, seed=515, schedule="double")
    a, b = _pair(N)
    with a, b:
        batch = a.solve(x0, xr, ft, ct)
        slow = np.argsort(-batch["iters"])[:24]
        assert batch["iters"][slow[-1]] > 55, "the sample holds no QP beyond the first restart mark"
        for q in slow:
            ra = _one(a, x0[q], xr[q], ft[q], ct[q], N, False)
            rb = _one(b, x0[q], xr[q], ft[q], ct[q], N, True)
            assert ra[2] == rb[2] == int(batch["status"][q]) and ra[3] == rb[3]
            assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1])
            assert ra[3] > 55
        assert a.batch1_launch_path() == "aql"


def test_handles_the_queue_is_not_for_stay_on_hip(torch_first, built_lib, aql_ready):
    from g1_locomotion_amd import BatchMPC, _lib
    N = 10
    x0, xr, ft, ct = orc.synthetic_batch(1, N, seed=3, schedule="double")
    ref = None
    for flags in (0, _lib.FLAG_DEFER_TAIL, _lib.FLAG_NO_SPIN):
        with BatchMPC(horizon=N, flags=flags) as eng:
            assert eng.batch1_launch_path() == "undecided"
            r = _one(eng, x0[0], xr[0], ft[0], ct[0], N, False)
            if ref is None:
                ref = r
                assert eng.batch1_launch_path() == "aql"
            else:
                assert eng.batch1_launch_path() in ("undecided", "aql")      # the queue may exist; these calls did not go through it
                assert np.abs(r[0] - ref[0]).max() < 1e-4 and r[2] == ref[2]           # (these flags also pick other kernel instantiations)
    with BatchMPC(horizon=N, timing=True) as eng:
        r = _one(eng, x0[0], xr[0], ft[0], ct[0], N, False)
        assert np.abs(r[0] - ref[0]).max() < 1e-4 and eng.last_kernel_ms() > 0.0


def test_mpc_update_runs_on_the_queue(torch_first, built_lib, aql_ready):
    """The reference's own call (run_simulation.py:106) ends up there."""
    from g1_locomotion_amd import MPC
    os.environ.pop("SRBDQP_NO_AQL", None)
    mpc = MPC(dt=0.04)
    mpc.init_matrices()
    N = mpc.HORIZON_LENGTH
    x0, xr, ft, ct = orc.synthetic_batch(1, N, seed=11, schedule="double")
    mpc.x0[:, 0] = x0[0]; mpc.x_ref_hor[:] = xr[0]
    u0, x1 = mpc.update(ct[0].reshape(N, 4), ft[0].reshape(N, 12), None, x_current=mpc.x0, one_rollout=True)
    o = orc.update(orc.params_for(N), x0[0], xr[0], ft[0], ct[0])
    assert np.abs(np.asarray(u0).reshape(-1) - np.asarray(o["u"]).reshape(-1)[:12]).max() < 2e-3
    assert mpc._engine.batch1_launch_path() == "aql"
    mpc.close()


@pytest.mark.parametrize("schedule", ["double", "single"])
def test_completion_word_with_checksum_never_hands_over_partial_outputs(torch_first, built_lib, schedule):
    """The one staged QP's completion word leaves the GPU right behind its outputs, without a fence, and carries their XOR; the host takes the outputs only once what it
    reads agrees with it (srbdqp_common.hpp signal_done_checksum, srbdqp.hip wait_done).  Three different QPs in turn, 4000 calls: every call returns exactly what
    the first call for its QP returned -- an output array that still held the previous call's values would show."""
    from g1_locomotion_amd import BatchMPC
    N = 10
    x0, xr, ft, ct = orc.synthetic_batch(3, N, seed=909, schedule=schedule)
    os.environ.pop("SRBDQP_NO_AQL", None)
    with BatchMPC(horizon=N) as eng:
        first = [None, None, None]
        for i in range(4000):
            q = i % 3
            r = _one(eng, x0[q], xr[q], ft[q], ct[q], N, False)
            if first[q] is None:
                first[q] = r
                o = orc.update(orc.params_for(N), x0[q], xr[q], ft[q], ct[q])
                assert r[2] == o["status"] and np.abs(r[0] - o["u"]).max() < 2e-3 and np.abs(r[1] - o["x"]).max() < 1e-4
            else:
                assert r[2] == first[q][2] and r[3] == first[q][3], i
                assert np.array_equal(r[0], first[q][0]) and np.array_equal(r[1], first[q][1]), i
        assert not np.array_equal(first[0][0], first[1][0]) and not np.array_equal(first[1][0], first[2][0])


def test_early_exits_report_through_the_completion_records_too(torch_first, built_lib, aql_ready):
    """A flight phase (no stance contact anywhere: status 1, zero forces, the roll-out of x0 under gravity) leaves the batch-1 kernels before the set-up; its
    completion goes through the same records.  Alternating with ordinary QPs, single- and double-support instantiations."""
    from g1_locomotion_amd import BatchMPC
    N = 10
    os.environ.pop("SRBDQP_NO_AQL", None)
    for schedule in ("single", "double"):
        x0, xr, ft, ct = orc.synthetic_batch(2, N, seed=31, schedule=schedule)
        flight = np.zeros_like(ct[0])
        with BatchMPC(horizon=N) as eng:
            ref = _one(eng, x0[0], xr[0], ft[0], ct[0], N, False)
            for i in range(200):
                f = _one(eng, x0[1], xr[1], ft[1], flight, N, False)
                assert f[2] == 1 and f[3] == 0 and not f[0].any(), i
                assert np.allclose(f[1][0], x0[1]) and f[1][N, 5] < x0[1][5]            # falling
                r = _one(eng, x0[0], xr[0], ft[0], ct[0], N, False)
                assert r[2] == ref[2] and r[3] == ref[3] and np.array_equal(r[0], ref[0]) and np.array_equal(r[1], ref[1]), i
            assert eng.batch1_launch_path() == "aql"
