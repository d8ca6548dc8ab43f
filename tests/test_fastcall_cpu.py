"""g1_locomotion_amd/_fastcall (csrc/fastcall.c): the CPython binding of srbdqp_update_f64 / srbdqp_solve_staged_f64 that MPC.update() uses.  No GPU here: the module
only moves pointers, so it is bound to NumPy arrays standing in for the library's staging arrays and to ctypes callbacks standing in for the two C entry
points -- what is tested is the argument handling (the reference's per-step lists, run_simulation.py:94-106), the copies and the fall-through."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def fc():
    import __graft_entry__ as g
    g.build()
    from g1_locomotion_amd import _fastcall
    return _fastcall


def _bound(fc, N=10):
    st = dict(x0=np.zeros(13), xref=np.zeros((N, 13)), foot=np.zeros((N, 12)), contact=np.zeros((N, 4), np.uint8), pcom=np.zeros((N, 3)),
              u=np.zeros((N, 12)), x=np.zeros((N + 1, 13)), status=np.zeros(1, np.int32), iters=np.zeros(1, np.int32))
    calls = []
    UPD = C.CFUNCTYPE(C.c_int, *([C.c_void_p] * 11))
    STG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32)

    def update(h, x0, xref, foot, ct, pcom, u0, u, x, s, it):
        calls.append(("update", h, x0, xref, foot, ct, pcom, u0, u, x))
        st["u"][:] = np.arange(N * 12).reshape(N, 12) + st["x0"][0]          # any function of the staged inputs
        st["x"][:] = np.arange((N + 1) * 13).reshape(N + 1, 13) * 0.5
        st["status"][0] = 1 if st["contact"].any() else 2
        return 0 if st["x0"][1] > -100.0 else -3

    def staged(h, B, a, b, c, d):
        calls.append(("staged", h, B, a, b, c, d))
        return 0
    cu, cs = UPD(update), STG(staged)
    addr = lambda f: C.cast(f, C.c_void_p).value
    p = lambda a: a.ctypes.data
    cap = fc.bind(addr(cu), addr(cs), 0xABCD, N, p(st["x0"]), p(st["xref"]), p(st["foot"]), p(st["contact"]), p(st["pcom"]), p(st["u"]), p(st["x"]),
                  p(st["status"]), p(st["iters"]))
    return cap, st, calls, (cu, cs)


def test_update_takes_the_references_call_and_copies_into_the_staging_arrays(fc):
    N = 10
    cap, st, calls, keep = _bound(fc, N)
    rng = np.random.default_rng(0)
    x0 = rng.normal(size=(13, 1)); xref = rng.normal(size=(N, 13)); feet = rng.normal(size=(N, 12))
    contact_horizon = [np.array([1, 1, 0, 1]) for _ in range(N)]                 # int64 per-step arrays, as the reference builds them
    c_horizon = [feet[k].copy() for k in range(N)]
    pcom = xref[:, 3:6].copy()
    u0, x1, status, rc = fc.update(cap, contact_horizon, c_horizon, pcom, x0, xref, True)
    assert rc == 0 and status == 1
    np.testing.assert_array_equal(st["x0"], x0.reshape(13)); np.testing.assert_array_equal(st["xref"], xref)
    np.testing.assert_array_equal(st["foot"], feet); np.testing.assert_array_equal(st["pcom"], pcom)
    np.testing.assert_array_equal(st["contact"], np.tile(np.array([1, 1, 0, 1], np.uint8), (N, 1)))
    assert u0.shape == (12, 1) and x1.shape == (N + 1, 13) and u0.flags.owndata and x1.flags.owndata
    np.testing.assert_array_equal(u0.reshape(-1), st["u"][0]); np.testing.assert_array_equal(x1, st["x"])
    name, h, px0, pxr, pft, pct, ppc, pu0, pu, px = calls[-1]
    assert h == 0xABCD and px0 == st["x0"].ctypes.data and ppc == st["pcom"].ctypes.data and pu0 == st["u"].ctypes.data and pu is None and px == st["x"].ctypes.data
    # whole arrays instead of lists, no CoM horizon, only rows 0..1 of the roll-out, bool / uint8 / float flags
    for flags in (np.ones((N, 4), bool), np.ones((N, 4), np.uint8), np.ones((N, 4)), tuple(np.ones(4, np.int32) for _ in range(N))):
        u0, x1, status, rc = fc.update(cap, flags, feet, None, x0.reshape(13), xref, False)
        assert rc == 0 and x1.shape == (2, 13) and calls[-1][6] is None and st["contact"].all()
    u0, x1, status, rc = fc.update(cap, contact_horizon, c_horizon, xref[:, 3:6], x0, xref, True)         # a strided view of x_ref_hor as the CoM horizon
    assert rc == 0
    np.testing.assert_array_equal(st["pcom"], xref[:, 3:6])
    assert fc.solve_time(cap) >= 0.0
    # a status other than SOLVED and a failing call come back as they are
    assert fc.update(cap, np.zeros((N, 4), np.uint8), feet, None, x0, xref, True)[2] == 2
    bad = x0.copy(); bad[1] = -1000.0
    assert fc.update(cap, np.ones((N, 4), np.uint8), feet, None, bad, xref, True)[3] == -3


def test_update_declines_what_it_does_not_recognise(fc):
    N = 10
    cap, st, calls, keep = _bound(fc, N)
    x0, xref, feet, ct = np.zeros((13, 1)), np.zeros((N, 13)), np.zeros((N, 12)), np.ones((N, 4), np.uint8)
    n0 = len(calls)
    assert fc.update(cap, ct, feet.astype(np.float32), None, x0, xref, True) is NotImplemented           # dtype
    assert fc.update(cap, ct, [list(r) for r in feet], None, x0, xref, True) is NotImplemented           # nested lists
    assert fc.update(cap, ct, [feet[k] for k in range(N - 1)], None, x0, xref, True) is NotImplemented   # wrong horizon
    assert fc.update(cap, ct, feet, np.zeros((N, 2)), x0, xref, True) is NotImplemented                  # CoM horizon of the wrong shape
    assert fc.update(cap, ct, feet, None, np.zeros(12), xref, True) is NotImplemented
    assert fc.update(cap, [np.array(["a"] * 4)] * N, feet, None, x0, xref, True) is NotImplemented
    assert len(calls) == n0                                                                             # the library was not entered
    with pytest.raises(TypeError):
        fc.update(cap, ct, feet)
    with pytest.raises((ValueError, TypeError)):
        fc.update(object(), ct, feet, None, x0, xref, True)


def test_solve_staged_passes_its_arguments_through(fc):
    cap, st, calls, keep = _bound(fc)
    assert fc.solve_staged(cap, 3, 1, 0, 1, 0) == 0
    assert calls[-1] == ("staged", 0xABCD, 3, 1, 0, 1, 0)
    with pytest.raises(ValueError):
        fc.bind(0, 0, 0, 10, 0, 0, 0, 0, 0, 0, 0, 0, 0)
