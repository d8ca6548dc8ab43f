"""GPU parity tests proper: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs.

Tolerances (fp64 path), stated once:
  * assembly (H, g, bounds) vs oracle:           relative 1e-11 of the matrix' max entry
  * forces vs the oracle's ADMM twin:            <= 2e-3 N when the iteration counts agree within one check interval
                                                 (the two differ only by summation order; an iterate landing within
                                                 rounding of the stopping threshold may stop one check later)
  * forces vs the independent exact QP optimum:  <= 5e-2 N  (2.5e-4 of a nominal 200 N stance force; ADMM stops
                                                 at eps_abs = eps_rel = 1e-6 in the scaled variables)
"""
import numpy as np
import pytest

import srbd_oracle as orc

pytestmark = pytest.mark.gpu

TOL_TWIN_N = 2e-3
TOL_EXACT_N = 5e-2


@pytest.fixture(scope="module")
def torch_first():
    import torch  # load torch's HIP runtime before libsrbdqp.so so both share one
    assert torch.cuda.is_available()
    return torch


def _engine(N, **kw):
    from g1_locomotion_amd import BatchMPC
    return BatchMPC(horizon=N, **kw)


@pytest.mark.parametrize("N,schedule", [(10, "single"), (10, "double"), (10, "mixed"), (8, "single"), (4, "double")])
def test_assembly_matches_oracle(torch_first, built_lib, N, schedule):
    B = 6
    x0, xr, ft, ct = orc.synthetic_batch(B, N, seed=100 + N, schedule=schedule)
    with _engine(N) as eng:
        got = eng.assemble(x0, xr, ft, ct)
    p = orc.SrbdParams()
    for b in range(B):
        qp = orc.build_qp(p, x0[b], xr[b], ft[b], ct[b])
        sP = np.abs(qp["P"]).max()
        assert np.abs(got["P"][b] - qp["P"]).max() <= 1e-11 * sP
        assert np.abs(got["q"][b] - qp["q"]).max() <= 1e-11 * max(1.0, np.abs(qp["q"]).max())
        np.testing.assert_array_equal(got["l"][b], qp["l"])
        np.testing.assert_array_equal(got["u"][b], qp["u"])


@pytest.mark.parametrize("kernel", ["gj", "auto"])
@pytest.mark.parametrize("N,schedule,B", [(10, "single", 24), (10, "double", 8), (10, "mixed", 16), (8, "mixed", 8)])
def test_solve_matches_oracle_and_exact_optimum(torch_first, built_lib, kernel, N, schedule, B):
    from g1_locomotion_amd import _lib
    kid = {"gj": _lib.KERNEL_GJ, "auto": _lib.KERNEL_AUTO}[kernel]
    x0, xr, ft, ct = orc.synthetic_batch(B, N, seed=200 + N, schedule=schedule)
    with _engine(N, kernel=kid) as eng:
        out = eng.solve(x0, xr, ft, ct, want_y=True)
        assert eng.kernel_name().startswith("gj_" if kernel == "gj" else "mfma_"), eng.kernel_name()
    p = orc.SrbdParams()
    for b in range(B):
        ref = orc.update(p, x0[b], xr[b], ft[b], ct[b])
        assert out["status"][b] == ref["status"] == orc.STATUS_SOLVED
        assert abs(int(out["iters"][b]) - ref["iters"]) <= p.check_every, (b, out["iters"][b], ref["iters"])
        assert np.abs(out["u"][b] - ref["u"]).max() <= TOL_TWIN_N
        assert np.abs(out["x"][b] - ref["x"]).max() <= 1e-5
        xs, ys = orc.solve_reference(p, ref["qp"])
        assert np.abs(out["u"][b].reshape(-1) - xs * p.force_scale).max() <= TOL_EXACT_N
        # solver-independent acceptance: KKT residuals of the GPU primal/dual pair in the scaled problem
        qp = ref["qp"]
        kr = orc.kkt_residuals(qp["P"], qp["q"], qp["A"], qp["l"], qp["u"], out["u"][b].reshape(-1) / p.force_scale, out["y"][b])
        assert kr["primal"] <= 1e-4 and kr["stationarity"] <= 1e-3 * max(1.0, np.abs(qp["q"]).max()), kr


def test_warm_start_reduces_iterations(torch_first, built_lib):
    N, B = 10, 16
    x0, xr, ft, ct = orc.synthetic_batch(B, N, seed=321, schedule="single")
    with _engine(N) as eng:
        cold = eng.solve(x0, xr, ft, ct, want_y=True)
        warm = eng.solve(x0, xr, ft, ct, warm_u=cold["u"].reshape(B, -1), warm_y=cold["y"], want_y=True)
    assert (warm["status"] == orc.STATUS_SOLVED).all()
    assert (warm["iters"] <= 5).all(), warm["iters"]
    assert np.abs(warm["u"] - cold["u"]).max() <= TOL_TWIN_N
