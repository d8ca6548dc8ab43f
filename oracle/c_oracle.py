"""ctypes wrapper of oracle/libsrbd_oracle.so (the plain-C restatement).  TEST INFRASTRUCTURE ONLY:
imported by tests/ and bench.py's cpu_baseline leg, never by the product path."""
import ctypes as C
import os
import subprocess

import numpy as np

import srbd_oracle as orc

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsrbd_oracle.so")


class CParams(C.Structure):
    _fields_ = [("dt", C.c_double), ("mass", C.c_double), ("inertia", C.c_double * 3), ("mu", C.c_double),
                ("fz_min", C.c_double), ("fz_max", C.c_double), ("q_diag", C.c_double * 13), ("r_diag", C.c_double),
                ("force_scale", C.c_double), ("rho", C.c_double), ("rho_eq_scale", C.c_double), ("sigma", C.c_double),
                ("alpha", C.c_double), ("eps_abs", C.c_double), ("eps_rel", C.c_double), ("rho_fz_scale", C.c_double),
                ("max_iter", C.c_int), ("check_every", C.c_int), ("rho_restart_iter", C.c_int), ("eliminate_swing", C.c_int),
                ("rho_restart_count", C.c_int)]


def _params(p: orc.SrbdParams) -> CParams:
    c = CParams()
    for k in ("dt", "mass", "mu", "fz_min", "fz_max", "r_diag", "force_scale", "rho", "rho_eq_scale", "sigma", "alpha",
              "eps_abs", "eps_rel", "rho_fz_scale", "max_iter", "check_every", "rho_restart_iter", "eliminate_swing", "rho_restart_count"):
        setattr(c, k, getattr(p, k))
    for i in range(3):
        c.inertia[i] = p.inertia[i]
    for i in range(13):
        c.q_diag[i] = p.q_diag[i]
    return c


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        lib = C.CDLL(_SO)
        vp = C.c_void_p
        lib.srbd_oracle_solve_batch.argtypes = [C.POINTER(CParams), C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int]
        lib.srbd_oracle_solve_batch.restype = C.c_int
        lib.srbd_oracle_solve.argtypes = [C.POINTER(CParams), C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.srbd_oracle_solve.restype = C.c_int
        lib.srbd_oracle_work_doubles.argtypes = [C.c_int]
        lib.srbd_oracle_work_doubles.restype = C.c_size_t
        _lib = lib
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def solve_batch(p: orc.SrbdParams, x0, x_ref, foot, contact, pcom=None, nthreads=1):
    lib = load()
    x0 = np.ascontiguousarray(x0, np.float64); B = x0.shape[0]
    x_ref = np.ascontiguousarray(x_ref, np.float64); N = x_ref.shape[1]
    foot = np.ascontiguousarray(foot, np.float64)
    contact = np.ascontiguousarray(np.asarray(contact) != 0, np.uint8)
    pcom = None if pcom is None else np.ascontiguousarray(pcom, np.float64)
    u = np.empty((B, N, 12)); x = np.empty((B, N + 1, 13))
    iters = np.empty(B, np.int32); status = np.empty(B, np.int32)
    cp = _params(p)
    rc = lib.srbd_oracle_solve_batch(C.byref(cp), N, B, _p(x0), _p(x_ref), _p(foot), _p(contact), _p(pcom), _p(u), _p(x),
                                     _p(iters), _p(status), int(nthreads))
    if rc != 0:
        raise RuntimeError("srbd_oracle_solve_batch failed")
    return dict(u=u, x=x, iters=iters, status=status)


def assemble(p: orc.SrbdParams, x0, x_ref, foot, contact, pcom=None):
    """P, q of ONE QP from the C restatement."""
    lib = load()
    x0 = np.ascontiguousarray(x0, np.float64).reshape(13)
    x_ref = np.ascontiguousarray(x_ref, np.float64); N = x_ref.shape[0]
    foot = np.ascontiguousarray(foot, np.float64)
    contact = np.ascontiguousarray(np.asarray(contact) != 0, np.uint8)
    pcom = None if pcom is None else np.ascontiguousarray(pcom, np.float64)
    n = 12 * N
    P = np.empty((n, n)); q = np.empty(n); u = np.empty(n); x = np.empty((N + 1, 13))
    it = C.c_int(0); st = C.c_int(0)
    work = np.empty(lib.srbd_oracle_work_doubles(N))
    cp = _params(p)
    lib.srbd_oracle_solve(C.byref(cp), N, _p(x0), _p(x_ref), _p(foot), _p(contact), _p(pcom), _p(u), _p(x), _p(P), _p(q),
                          C.addressof(it), C.addressof(st), _p(work))
    return dict(P=P, q=q, u=u.reshape(N, 12), x=x, iters=it.value, status=st.value)
