"""Host-side mirror of the reference's MPC object for the hot path, over the C-ABI HIP library.

``MPC`` keeps the duck-typed surface the reference's callers use
(g1_mujoco_sim/src/run_simulation.py:169-170,73-82,96,103,106):

    MPC = mpc.MPC(dt=0.04); MPC.init_matrices()
    MPC.x0[...] ; MPC.x_ref_hor[...] ; MPC.g ; MPC.HORIZON_LENGTH
    u_opt0, x_opt1 = MPC.update(contact_horizon, c_horizon, p_com_horizon, x_current=MPC.x0, one_rollout=True)

``solve()`` (named by BASELINE.json, no call site in the reference tree) is this build's assemble+solve step
that ``update()`` calls.  ``BatchMPC`` is the same operation for B independent QPs (NumPy host arrays, or
device pointers of HBM-resident buffers).  All compute happens in libsrbdqp.so on the GPU.
"""
from __future__ import annotations

import ctypes as C
import time
import warnings
from typing import Optional, Sequence

import numpy as np

from . import _lib
try:                                    # CPython binding of the two batch-1 calls (csrc/fastcall.c, built by __graft_entry__.build()); optional: ctypes otherwise
    from . import _fastcall
except ImportError:                     # pragma: no cover
    _fastcall = None
from ._lib import NX, NU, NC, SrbdqpError


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


_p = _ptr


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _as(a, dtype, shape, name):
    arr = np.ascontiguousarray(a, dtype=dtype)
    if arr.shape != tuple(shape):
        try:
            arr = arr.reshape(shape)
        except ValueError as e:
            raise ValueError(f"{name}: expected shape {tuple(shape)}, got {np.shape(a)}") from e
    return arr


class BatchMPC:
    """B independent SRBD convex-MPC QPs per call.  One instance <-> one HIP stream <-> one thread."""

    def __init__(self, horizon: int = 10, dt: float = 0.04, device: int = 0, kernel: int = _lib.KERNEL_AUTO,
                 timing: bool = False, **overrides):
        lib = _lib.load()
        cfg = _lib.default_config()
        cfg.horizon = int(horizon)
        cfg.dt = float(dt)
        cfg.device = int(device)
        cfg.kernel = int(kernel)
        cfg.flags = _lib.FLAG_TIMING if timing else 0
        for k, v in overrides.items():
            if not hasattr(cfg, k):
                raise TypeError(f"unknown srbdqp_config field {k!r}")
            if k in ("q_diag", "inertia"):
                arr = getattr(cfg, k)
                if len(v) != len(arr):
                    raise ValueError(f"{k}: expected {len(arr)} values")
                for i, x in enumerate(v):
                    arr[i] = float(x)
            else:
                setattr(cfg, k, type(getattr(cfg, k))(v))
        self.cfg = cfg
        self._lib = lib
        self._h = C.c_void_p()
        rc = lib.srbdqp_create(C.byref(cfg), C.byref(self._h))
        if rc != _lib.OK:
            msg = lib.srbdqp_last_error(None)
            self._h = C.c_void_p()
            raise SrbdqpError(f"srbdqp_create failed ({rc}): {msg.decode() if msg else '?'}")
        self.N = int(horizon)
        self.n = NU * self.N
        self.m = _lib.ROWS_PER_STEP * self.N

    # -- lifetime ------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._stage = None                      # the views point into memory the library is about to free
            self._lib.srbdqp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- host-buffer API -------------------------------------------------------------------------------
    def solve(self, x0, x_ref, foot, contact, pcom=None, warm_u=None, warm_y=None, want_x=True, want_y=False,
              dtype=np.float64):
        """Solve B QPs.  Returns dict(u (B,N,12) [N], x (B,N+1,13), y (B,20N), status (B,), iters (B,)).
        dtype=np.float32 goes through srbdqp_solve_batch_f32: fp32 buffers and fp32 ADMM iterations (fp64 set-up)."""
        N, n, m = self.N, self.n, self.m
        dt = np.dtype(dtype)
        if dt not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise TypeError("dtype must be float64 or float32")
        x0 = np.ascontiguousarray(x0, dtype=dt)
        B = x0.size // NX
        x0 = _as(x0, dt, (B, NX), "x0")
        x_ref = _as(x_ref, dt, (B, N, NX), "x_ref")
        foot = _as(foot, dt, (B, N, NU), "foot")
        contact = _as(np.asarray(contact) != 0, np.uint8, (B, N, NC), "contact")
        pcom = None if pcom is None else _as(pcom, dt, (B, N, 3), "pcom")
        warm_u = None if warm_u is None else _as(warm_u, dt, (B, n), "warm_u")
        warm_y = None if warm_y is None else _as(warm_y, dt, (B, m), "warm_y")
        u = np.empty((B, N, NU), dt)
        x = np.empty((B, N + 1, NX), dt) if want_x else None
        y = np.empty((B, m), dt) if want_y else None
        status = np.empty(B, np.int32)
        iters = np.empty(B, np.int32)
        fn = self._lib.srbdqp_solve_batch_f64 if dt == np.dtype(np.float64) else self._lib.srbdqp_solve_batch_f32
        rc = fn(self._h, B, _ptr(x0), _ptr(x_ref), _ptr(foot), _ptr(contact), _ptr(pcom), _ptr(warm_u), _ptr(warm_y),
                _ptr(u), _ptr(x), _ptr(y), _ptr(status), _ptr(iters))
        _lib.check(rc, self._h)
        return dict(u=u, x=x, y=y, status=status, iters=iters)

    def assemble(self, x0, x_ref, foot, contact, pcom=None):
        """QP data in the scaled variables as the shipped compact / one-wave kernel builds it (srbdqp_assemble_f64):
        dict(P (B,n,n), q (B,n), l (B,m), u (B,m)); rows / columns of swing-contact variables are 0 (presolve)."""
        N, n, m = self.N, self.n, self.m
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        B = x0.size // NX
        x0 = _as(x0, np.float64, (B, NX), "x0")
        x_ref = _as(x_ref, np.float64, (B, N, NX), "x_ref")
        foot = _as(foot, np.float64, (B, N, NU), "foot")
        contact = _as(np.asarray(contact) != 0, np.uint8, (B, N, NC), "contact")
        pcom = None if pcom is None else _as(pcom, np.float64, (B, N, 3), "pcom")
        P = np.empty((B, n, n)); q = np.empty((B, n)); lo = np.empty((B, m)); hi = np.empty((B, m))
        rc = self._lib.srbdqp_assemble_f64(self._h, B, _ptr(x0), _ptr(x_ref), _ptr(foot), _ptr(contact), _ptr(pcom),
                                           _ptr(P), _ptr(q), _ptr(lo), _ptr(hi))
        _lib.check(rc, self._h)
        return dict(P=P, q=q, l=lo, u=hi)

    def assemble_wrench(self, x0, x_ref, foot, contact, pcom=None):
        """What the general kernel builds before its factorisation (srbdqp_assemble_wrench_f64): dict(T (B,6N,6N), q (B,12N),
        Bd (B,12N,12), Vcol (B,12N,6), Vrow (B,12N,6), goff (B,N+1) int)."""
        N, n = self.N, self.n
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        B = x0.size // NX
        x0 = _as(x0, np.float64, (B, NX), "x0")
        x_ref = _as(x_ref, np.float64, (B, N, NX), "x_ref")
        foot = _as(foot, np.float64, (B, N, NU), "foot")
        contact = _as(np.asarray(contact) != 0, np.uint8, (B, N, NC), "contact")
        pcom = None if pcom is None else _as(pcom, np.float64, (B, N, 3), "pcom")
        T = np.empty((B, 6 * N, 6 * N)); q = np.empty((B, n)); bl = np.empty((B, n, 24)); go = np.empty((B, N + 1))
        rc = self._lib.srbdqp_assemble_wrench_f64(self._h, B, _ptr(x0), _ptr(x_ref), _ptr(foot), _ptr(contact), _ptr(pcom),
                                                  _ptr(T), _ptr(q), _ptr(bl), _ptr(go))
        _lib.check(rc, self._h)
        return dict(T=T, q=q, Bd=bl[:, :, 0:12].copy(), Vcol=bl[:, :, 12:18].copy(), Vrow=bl[:, :, 18:24].copy(), goff=go.astype(int))

    # -- device-buffer API -----------------------------------------------------------------------------
    def solve_device(self, B, x0, x_ref, foot, contact, u_out, x_out=0, y_out=0, status=0, iters=0, pcom=0,
                     warm_u=0, warm_y=0, stream=0, f32=False):
        """Enqueue a solve on HBM-resident buffers.  Every argument is a raw device address (int), e.g.
        ``tensor.data_ptr()``; 0 = absent.  ``stream`` is a hipStream_t address (0 = the handle's own stream).
        f32=True: the buffers hold float32 (srbdqp_solve_batch_device_f32).  Does not synchronise."""
        v = lambda p: C.c_void_p(int(p)) if p else None
        fn = self._lib.srbdqp_solve_batch_device_f32 if f32 else self._lib.srbdqp_solve_batch_device_f64
        rc = fn(self._h, int(B), v(x0), v(x_ref), v(foot), v(contact), v(pcom), v(warm_u), v(warm_y), v(u_out), v(x_out),
                v(y_out), v(status), v(iters), v(stream))
        _lib.check(rc, self._h)

    def set_schedule_hint(self, iters_prev_ptr=0, length=0):
        """Device address and length of the previous step's iters[] (or 0): longest-first dispatch for the next device
        solves of at most `length` QPs.  A hint without a length is an error (it would silently never apply)."""
        if iters_prev_ptr and int(length) <= 0:
            raise ValueError("set_schedule_hint: pass the number of entries of iters_prev (length > 0) with the pointer")
        _lib.check(self._lib.srbdqp_set_schedule_hint(self._h, C.c_void_p(int(iters_prev_ptr)) if iters_prev_ptr else None,
                                                      int(length) if iters_prev_ptr else 0), self._h)

    def flush(self, stream=0):
        """FLAG_DEFER_TAIL: enqueue the continuations no later solve has picked up (srbdqp_flush); stream = a hipStream_t address, 0 = every
        stream this engine has launched on.  Does not synchronise.  Until the flush has completed in stream order the device arrays of the earlier
        solve_device() calls -- their OUTPUTS and their INPUTS (a continuation rebuilds its QP from the pointers of the launch it came from) -- must
        stay untouched (include/srbdqp.h, SRBDQP_FLAG_DEFER_TAIL)."""
        _lib.check(self._lib.srbdqp_flush(self._h, C.c_void_p(int(stream)) if stream else None), self._h)

    # -- low-latency staged API (small batches; the single-robot control loop) --------------------------
    def stage(self):
        """NumPy views of the library's pinned, GPU-mapped staging arrays (dict; first axis = capacity)."""
        if getattr(self, "_stage", None) is None:
            st = _lib.Stage()
            _lib.check(self._lib.srbdqp_stage_ptrs(self._h, C.byref(st)), self._h)
            cap, N, n, m = st.capacity, self.N, self.n, self.m

            def view(addr, ctype, shape):
                cnt = int(np.prod(shape))
                return np.ctypeslib.as_array((ctype * cnt).from_address(addr)).reshape(shape)
            d, u8, i32 = C.c_double, C.c_uint8, C.c_int32
            self._stage = dict(
                capacity=cap, x0=view(st.x0, d, (cap, NX)), x_ref=view(st.x_ref, d, (cap, N, NX)),
                foot=view(st.foot, d, (cap, N, NU)), contact=view(st.contact, u8, (cap, N, NC)),
                pcom=view(st.pcom, d, (cap, N, 3)), warm_u=view(st.warm_u, d, (cap, n)), warm_y=view(st.warm_y, d, (cap, m)),
                u=view(st.u, d, (cap, N, NU)), x=view(st.x, d, (cap, N + 1, NX)), y=view(st.y, d, (cap, m)),
                status=view(st.status, i32, (cap,)), iters=view(st.iters, i32, (cap,)))
            self._fcb = None
            if _fastcall is not None:       # the same two C entry points through their function pointers (no ctypes trampoline per call)
                raw = _lib.load_raw()
                addr = lambda f: C.cast(f, C.c_void_p).value
                self._fcb = _fastcall.bind(addr(raw.srbdqp_update_f64), addr(raw.srbdqp_solve_staged_f64), self._h.value, N, st.x0, st.x_ref, st.foot,
                                           st.contact, st.pcom, st.u, st.x, st.status, st.iters)
        return self._stage

    def solve_staged(self, B=1, use_pcom=False, use_warm=False, want_x=True, want_y=False):
        """One kernel launch over the first B staged QPs; inputs are read and outputs written in the staging arrays."""
        fcb = getattr(self, "_fcb", None)
        if fcb is not None:
            rc = _fastcall.solve_staged(fcb, int(B), int(use_pcom), int(use_warm), int(want_x), int(want_y))
            if rc:
                _lib.check(rc, self._h)
            return
        _lib.check(self._lib.srbdqp_solve_staged_f64(self._h, int(B), int(use_pcom), int(use_warm), int(want_x), int(want_y)), self._h)

    def prepare_staged(self, B=1, use_pcom=False):
        """Two-phase call, phase 1 (srbdqp_prepare_staged_f64): set-up of the first B staged QPs from a PREDICTED x0 (whatever the
        staging x0 holds); asynchronous."""
        _lib.check(self._lib.srbdqp_prepare_staged_f64(self._h, int(B), int(use_pcom)), self._h)

    def solve_prepared(self, B=1, want_x=True, want_y=False):
        """Two-phase call, phase 2 (srbdqp_solve_prepared_f64): x0 is read from the staging arrays again, the rest is as prepared."""
        _lib.check(self._lib.srbdqp_solve_prepared_f64(self._h, int(B), int(want_x), int(want_y)), self._h)

    def synchronize(self):
        _lib.check(self._lib.srbdqp_synchronize(self._h), self._h)

    # -- the steps either side of the QP (include/srbdqp_cascade.h) ----------------------------------------
    def swing(self, p_start, p_final, z_middle, progress, final_velocity_z=-0.02, first_half_share=0.80, want_coeff=False):
        """Batched swing-foot trajectory (swing_trajectory.py:38-89): p_start, p_final (B,3); z_middle, progress (B,).
        Returns dict(pos (B,3), vel_z (B,), acc_z (B,)[, coeff (B,7)])."""
        ps, pf = _c(p_start, np.float64).reshape(-1, 3), _c(p_final, np.float64).reshape(-1, 3)
        B = ps.shape[0]
        zm = np.ascontiguousarray(np.broadcast_to(np.asarray(z_middle, np.float64).reshape(-1), (B,)))
        t = np.ascontiguousarray(np.broadcast_to(np.asarray(progress, np.float64).reshape(-1), (B,)))
        if pf.shape[0] != B:
            raise ValueError("p_start and p_final must have the same number of rows")
        pos, vz, az = np.empty((B, 3)), np.empty(B), np.empty(B)
        co = np.empty((B, 7)) if want_coeff else None
        rc = self._lib.srbdqp_swing_f64(self._h, B, _p(ps), _p(pf), _p(zm), _p(t), float(final_velocity_z),
                                        float(first_half_share), _p(pos), _p(vz), _p(az), _p(co))
        _lib.check(rc, self._h)
        out = dict(pos=pos, vel_z=vz, acc_z=az)
        if want_coeff:
            out["coeff"] = co
        return out

    def wbid_reference(self, x_next, u0, foot, as_written=True):
        """Batched MPC -> WBID reference mapping (wbid.py:243-296): x_next (B,13), u0 (B,12), foot (B,12) or (B,4,3).
        Returns dict(R (B,3,3), base_vel (B,6), base_acc (B,6), com_acc (B,3), com_pos (B,3), com_vel (B,3), wrench (B,4,3))."""
        x, u, f = _c(x_next, np.float64).reshape(-1, NX), _c(u0, np.float64).reshape(-1, NU), _c(foot, np.float64).reshape(-1, NU)
        B = x.shape[0]
        if u.shape[0] != B or f.shape[0] != B:
            raise ValueError("x_next, u0 and foot must have the same number of rows")
        R, bv, ba, ca = np.empty((B, 3, 3)), np.empty((B, 6)), np.empty((B, 6)), np.empty((B, 3))
        rc = self._lib.srbdqp_wbid_reference_f64(self._h, B, _p(x), _p(u), _p(f), int(bool(as_written)), _p(R), _p(bv), _p(ba), _p(ca))
        _lib.check(rc, self._h)
        return dict(R=R, base_vel=bv, base_acc=ba, com_acc=ca, com_pos=x[:, 3:6].copy(), com_vel=x[:, 9:12].copy(),
                    wrench=u.reshape(B, 4, 3).copy())

    def _gait_struct(self, period_steps, double_support_steps, com_target, hip_offset_y):
        g = _lib.Gait()
        g.struct_size = C.sizeof(_lib.Gait)
        g.period_steps, g.double_support_steps = int(period_steps), int(double_support_steps)
        for i in range(3):
            g.com_target[i] = float(com_target[i])
        g.hip_offset_y = float(hip_offset_y)
        return g

    def mpc_inputs(self, x0, feet, stamp, v_ref, com_target, standing=None, period_steps=6, double_support_steps=1,
                   hip_offset_y=0.0645):
        """The step before the QP for B robots (msgs.MpcNode.step + msgs.AlternatingGait for one robot):
        x0 (B,13), feet (B,12) or (B,4,3), stamp (B,), v_ref (B,2), standing (B,) flags or None.
        Returns dict(x_ref (B,N,13), foot (B,N,12), contact (B,N,4) uint8, pcom (B,N,3), landing (B,3)) -- the arrays
        solve() / solve_device() take."""
        x = _c(x0, np.float64).reshape(-1, NX)
        B, N = x.shape[0], self.N
        f = _c(feet, np.float64).reshape(B, NU)
        t = _c(stamp, np.float64).reshape(B)
        v = _c(v_ref, np.float64).reshape(B, 2)
        sd = None if standing is None else _c(standing, np.uint8).reshape(B)
        g = self._gait_struct(period_steps, double_support_steps, com_target, hip_offset_y)
        xr, ft, ct = np.empty((B, N, NX)), np.empty((B, N, NU)), np.empty((B, N, NC), dtype=np.uint8)
        pc, lp = np.empty((B, N, 3)), np.empty((B, 3))
        rc = self._lib.srbdqp_mpc_inputs_f64(self._h, B, _p(x), _p(f), _p(t), _p(v), _p(sd), C.byref(g), _p(xr), _p(ft), _p(ct), _p(pc), _p(lp))
        _lib.check(rc, self._h)
        return dict(x_ref=xr, foot=ft, contact=ct, pcom=pc, landing=lp)

    def mpc_inputs_device(self, B, x0, feet, stamp, v_ref, com_target, x_ref, foot, contact, pcom, landing=0, standing=0,
                          period_steps=6, double_support_steps=1, hip_offset_y=0.0645, stream=0):
        """Device-pointer form of mpc_inputs(): raw addresses in, launch enqueued behind `stream`, returns at once."""
        v = lambda q: C.c_void_p(int(q)) if q else None
        g = self._gait_struct(period_steps, double_support_steps, com_target, hip_offset_y)
        rc = self._lib.srbdqp_mpc_inputs_device_f64(self._h, int(B), v(x0), v(feet), v(stamp), v(v_ref), v(standing), C.byref(g),
                                                    v(x_ref), v(foot), v(contact), v(pcom), v(landing), v(stream))
        _lib.check(rc, self._h)

    def last_kernel_ms(self) -> float:
        return float(self._lib.srbdqp_last_kernel_ms(self._h))

    def last_kernel_parts_ms(self):
        """(set-up ms, ADMM ms) of the last timed solve when it ran as the split pipeline, else None."""
        a, b = C.c_double(), C.c_double()
        if self._lib.srbdqp_last_kernel_parts_ms(self._h, C.byref(a), C.byref(b)) != _lib.OK:
            return None
        return float(a.value), float(b.value)

    def kernel_name(self) -> str:
        return self._lib.srbdqp_kernel_name(self._h).decode()

    def batch1_launch_path(self) -> str:
        """"aql" (the library's own HSA queue) or "hip: <why>" for the staged one-QP call; "undecided" before the first one."""
        return self._lib.srbdqp_batch1_launch_path(self._h).decode()


class RaggedMPC:
    """Mixed-horizon batches (BASELINE.json configs[4]) over srbdqp_solve_ragged_*: QPs in any order, each with its own
    horizon and contact schedule; the library sorts them into horizon buckets and launches every bucket on its own HIP
    stream, all in flight together."""

    def __init__(self, horizons=(8, 12, 16, 24), dt: float = 0.04, device: int = 0, **overrides):
        lib = _lib.load()
        cfg = _lib.default_config()
        cfg.dt = float(dt)
        cfg.device = int(device)
        for k, v in overrides.items():
            if not hasattr(cfg, k):
                raise TypeError(f"unknown srbdqp_config field {k!r}")
            setattr(cfg, k, type(getattr(cfg, k))(v))
        self.horizons = tuple(int(h) for h in horizons)
        hz = np.ascontiguousarray(self.horizons, dtype=np.int32)
        self._lib = lib
        self._h = C.c_void_p()
        rc = lib.srbdqp_ragged_create(C.byref(cfg), _ptr(hz), len(hz), C.byref(self._h))
        if rc != _lib.OK:
            msg = lib.srbdqp_ragged_last_error(None)
            self._h = C.c_void_p()
            raise SrbdqpError(f"srbdqp_ragged_create failed ({rc}): {msg.decode() if msg else '?'}")

    def _check(self, rc):
        if rc != _lib.OK:
            msg = self._lib.srbdqp_ragged_last_error(self._h)
            raise SrbdqpError(f"srbdqp error {rc}: {msg.decode() if msg else '?'}")

    def solve_packed(self, N_per_qp, x0, x_ref, foot, contact, want_x=True, dtype=np.float64):
        """Step-major packed host arrays (include/srbdqp.h): x0 (B,13), x_ref (sum N,13), foot (sum N,12), contact (sum N,4).
        Returns dict(u (sum N,12), x (sum N + B,13), status (B,), iters (B,), off (B+1,) row offsets).
        dtype=np.float32: fp32 buffers and iterations (srbdqp_solve_ragged_f32)."""
        dt = np.dtype(dtype)
        Nq = np.ascontiguousarray(N_per_qp, dtype=np.int32)
        B, rows = Nq.size, int(Nq.sum())
        x0 = _as(x0, dt, (B, NX), "x0")
        x_ref = _as(x_ref, dt, (rows, NX), "x_ref")
        foot = _as(foot, dt, (rows, NU), "foot")
        contact = _as(np.asarray(contact) != 0, np.uint8, (rows, NC), "contact")
        u = np.empty((rows, NU), dt); x = np.empty((rows + B, NX), dt) if want_x else None
        status = np.empty(B, np.int32); iters = np.empty(B, np.int32)
        fn = self._lib.srbdqp_solve_ragged_f64 if dt == np.dtype(np.float64) else self._lib.srbdqp_solve_ragged_f32
        self._check(fn(self._h, B, _ptr(Nq), _ptr(x0), _ptr(x_ref), _ptr(foot), _ptr(contact), _ptr(u), _ptr(x), _ptr(status), _ptr(iters)))
        return dict(u=u, x=x, status=status, iters=iters, off=np.concatenate([[0], np.cumsum(Nq)]))

    def solve_device(self, B, N_per_qp, x0, x_ref, foot, contact, u_out, x_out=0, status=0, iters=0, stream=0, f32=False,
                     warm_u=0, warm_y=0, y_out=0):
        """Packed arrays resident in HBM (raw device addresses); N_per_qp is a HOST int32 array.  Does not synchronise.
        f32=True: float32 buffers and iterations.  warm_u (sum N,12) [N] / warm_y (sum N,20) / y_out (sum N,20): warm start in,
        dual solution out (srbdqp_solve_ragged_warm_device_*)."""
        Nq = np.ascontiguousarray(N_per_qp, dtype=np.int32)
        v = lambda p: C.c_void_p(int(p)) if p else None
        if warm_u or warm_y or y_out:
            fn = self._lib.srbdqp_solve_ragged_warm_device_f32 if f32 else self._lib.srbdqp_solve_ragged_warm_device_f64
            self._check(fn(self._h, int(B), _ptr(Nq), v(x0), v(x_ref), v(foot), v(contact), v(warm_u), v(warm_y), v(u_out), v(x_out),
                           v(y_out), v(status), v(iters), v(stream)))
        else:
            fn = self._lib.srbdqp_solve_ragged_device_f32 if f32 else self._lib.srbdqp_solve_ragged_device_f64
            self._check(fn(self._h, int(B), _ptr(Nq), v(x0), v(x_ref), v(foot), v(contact), v(u_out), v(x_out), v(status), v(iters), v(stream)))

    def flush(self, stream=0):
        """flags=FLAG_DEFER_TAIL: make `stream` (0 = the object's own) wait for the restart passes still running on the buckets' tail streams
        (srbdqp_ragged_flush).  Does not synchronise.  The input and output arrays of the earlier solve_device() calls must stay untouched until the
        flush has completed in stream order (the passes read the former and write the latter)."""
        self._check(self._lib.srbdqp_ragged_flush(self._h, C.c_void_p(int(stream)) if stream else None))

    def solve(self, problems):
        """problems: sequence of dicts(x0 (13,), x_ref (N,13), foot (N,12), contact (N,4)) with per-QP N.
        Returns a list of dicts(u (N,12), x (N+1,13), status, iters) in the input order."""
        Nq = [int(np.asarray(pr["x_ref"]).shape[0]) for pr in problems]
        for N in Nq:
            if N not in self.horizons:
                raise ValueError(f"no engine for horizon {N} (have {sorted(self.horizons)})")
        if any("pcom" in pr for pr in problems):
            raise ValueError("the ragged path takes the CoM horizon from x_ref (no separate pcom)")
        res = self.solve_packed(Nq, np.stack([pr["x0"] for pr in problems]), np.concatenate([pr["x_ref"] for pr in problems]),
                                np.concatenate([pr["foot"] for pr in problems]), np.concatenate([pr["contact"] for pr in problems]))
        off = res["off"]
        return [dict(u=res["u"][off[i]:off[i + 1]], x=res["x"][off[i] + i:off[i + 1] + i + 1], status=int(res["status"][i]),
                     iters=int(res["iters"][i])) for i in range(len(problems))]

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.srbdqp_ragged_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MPC:
    """Drop-in for ``srbd_mpc.mpc.MPC`` on the hot path (run_simulation.py:169-170,73-82,96,103,106)."""

    def __init__(self, dt: float = 0.04, horizon: int = 10, device: int = 0, strict: bool = True, **overrides):
        """Every solve starts from zero.  (Rounds 1-4 had `warm_start=True`: the previous plan and duals shifted by one step.  Removed in round 5 -- on this
        ADMM (sigma -> 0, alpha = 1.6) an error of the starting point decays by |1 - alpha| = 0.6 per iteration whatever else happens, and the dual residual
        sees it through P: the shifted plan is 0.2 |x*| from the new optimum but ~400 |q| from it through P (zero: 1 |q|), i.e. log(400) / log(1 / 0.6) = 12
        iterations WORSE than zero: 41 against 30 on closed loops, profiles/r05_warm_start_sweep.txt, DESIGN.md section 2.  The C-ABI keeps warm_u / warm_y.)
        strict=True (default): a solve that ends with a negative status (SRBDQP_NUMERICAL, SRBDQP_CONTACT_BOUND: the kernel
        returned all-zero forces) raises SrbdqpError instead of handing zeros to the WBID step; a solve that stops at the
        iteration cap (SRBDQP_MAX_ITER) returns its best iterate with a RuntimeWarning.  strict=False returns whatever came
        back; ``status`` / ``iters`` / ``solve_time`` always hold the outcome of the last call."""
        self.dt = float(dt)
        self.HORIZON_LENGTH = int(horizon)
        self.g = -9.80665                       # ros_run_simulation.py:58
        self.x0 = np.zeros((NX, 1))
        self.x0[12] = self.g
        self.x_ref_hor = np.zeros((self.HORIZON_LENGTH, NX))
        self.x_ref_hor[:, 12] = self.g
        if "warm_start" in overrides:
            raise TypeError("MPC(warm_start=...) was removed in round 5: a shift-based start costs this ADMM 10 iterations instead of saving any "
                            "(profiles/r05_warm_start_sweep.txt); BatchMPC.solve(warm_u=, warm_y=) remains for callers with a better start")
        self.strict = bool(strict)
        self._solve_time = 0.0                  # seconds spent in the last solve (the node's solve-time statistic)
        self._fcb = None                        # the CPython binding of srbdqp_update_f64 (csrc/fastcall.c), once bound
        self._device = device
        self._overrides = overrides
        self._engine: Optional[BatchMPC] = None
        self._u_opt = None
        self._x_opt = None
        self._status = 0
        self._iters = 0
        self._last_fc = False                   # ... through the CPython binding (solve_time is read from it)
        self._last_fast = False                 # the last call went through update()'s bound fast path: results are read from the staging arrays
        self._upd = None                        # srbdqp_update_f64 with every argument bound (see _bind)

    # outcome of the last call.  The fast path of update() leaves everything in the library's staging arrays and these read it there on demand.
    @property
    def solve_time(self) -> float:
        """seconds inside the library during the last call"""
        return _fastcall.solve_time(self._fcb) if (self._last_fc and self._fcb is not None) else self._solve_time

    @property
    def status(self) -> int:
        return int(self._s_status[0]) if self._last_fast else self._status

    @property
    def iters(self) -> int:
        return int(self._s_iters[0]) if self._last_fast else self._iters

    @property
    def u_opt(self):
        """(N, 12) last optimal forces [N]"""
        return self._s_u.copy() if self._last_fast else self._u_opt

    @property
    def x_opt(self):
        """(N+1, 13) last roll-out"""
        return self._s_x.copy() if self._last_fast else self._x_opt

    def init_matrices(self):
        """Allocate the engine (stream + device workspace).  run_simulation.py:170."""
        if self._engine is None:
            self._engine = BatchMPC(horizon=self.HORIZON_LENGTH, dt=self.dt, device=self._device, **self._overrides)
        return self

    def _bind(self):
        """Bind srbdqp_update_f64 once: NumPy views of the staging arrays for this robot's QP and the eleven arguments of the call as
        ctypes objects (inputs and outputs ARE the staging arrays, so the C side copies nothing).  A call is then
        ``self._upd(*self._args)``: no boxing, no argument conversion."""
        if self._engine is None:
            self.init_matrices()
        eng = self._engine
        st = eng.stage()
        self._s_x0, self._s_xref, self._s_pcom = st["x0"][0], st["x_ref"][0], st["pcom"][0]
        self._s_x0c = st["x0"][0].reshape(NX, 1)        # the reference passes MPC.x0, a (13, 1) column (run_simulation.py:73-77,106): same-shape assignment is the cheapest copy
        self._s_foot2, self._s_ct2 = st["foot"][0], st["contact"][0]
        self._s_foot, self._s_ct = st["foot"][0].reshape(-1), st["contact"][0].reshape(-1)
        self._s_u, self._s_x = st["u"][0], st["x"][0]
        self._s_u0, self._s_x01 = st["u"][0][0].reshape(NU, 1), st["x"][0][:2]
        self._s_status, self._s_iters = st["status"], st["iters"]
        P = lambda a: C.c_void_p(a.ctypes.data)
        h = C.c_void_p(eng._h.value)
        ins = (h, P(self._s_x0), P(self._s_xref), P(self._s_foot), P(self._s_ct))
        outs = (P(self._s_u), None, P(self._s_x), None, None)         # u0_out = the staging u (in place), no second plan copy, x in place
        self._args_pcom = ins + (P(self._s_pcom),) + outs
        self._args_nopcom = ins + (None,) + outs
        self._upd = _lib.load_raw().srbdqp_update_f64
        self._fcb = getattr(eng, "_fcb", None)
        return self._upd

    def solve(self, x_current, x_ref_hor, c_horizon, contact_horizon, p_com_horizon=None):
        """Assemble + solve one QP on the GPU; returns (u (N,12) newtons, x (N+1,13)).
        Inputs are written straight into the library's pinned staging arrays (no hipMemcpy on this path)."""
        if self._engine is None:
            self.init_matrices()
        self._last_fast = self._last_fc = False
        eng, N = self._engine, self.HORIZON_LENGTH
        st = eng.stage()
        st["x0"][0] = np.asarray(x_current, dtype=np.float64).reshape(NX)
        st["x_ref"][0] = np.asarray(x_ref_hor, dtype=np.float64).reshape(N, NX)
        # the reference passes per-step lists (run_simulation.py:94-101): concatenating straight into the staging
        # arrays is ~1 us cheaper than np.asarray(list) + copy; anything irregular takes the general path
        try:
            np.concatenate(c_horizon, out=st["foot"][0].reshape(-1))
            np.concatenate(contact_horizon, out=st["contact"][0].reshape(-1), casting="unsafe")   # 0 / non-zero flags
        except (ValueError, TypeError):
            st["foot"][0] = np.asarray(c_horizon, dtype=np.float64).reshape(N, NU)
            st["contact"][0] = np.asarray(contact_horizon).reshape(N, NC) != 0
        use_pcom = p_com_horizon is not None
        if use_pcom:
            st["pcom"][0] = np.asarray(p_com_horizon, dtype=np.float64).reshape(N, 3)
        t0 = time.perf_counter()
        eng.solve_staged(1, use_pcom=use_pcom, use_warm=False, want_x=True, want_y=False)
        self._solve_time = time.perf_counter() - t0
        self._status = int(st["status"][0])
        self._iters = int(st["iters"][0])
        if self._status != _lib.SOLVED:
            self._not_solved(self._status, self._iters, 4)
        self._u_opt = st["u"][0].copy()
        self._x_opt = st["x"][0].copy()
        return self._u_opt, self._x_opt

    def _not_solved(self, status, iters, stacklevel):
        """strict mode: a failed solve raises, a stop at the iteration cap warns (the reference's consumer has no status handling,
        ros_run_simulation.py:188-218: without this a zero-force plan would go straight to the WBID step)"""
        if not self.strict:
            return
        if status < 0:
            raise SrbdqpError(f"MPC solve failed with status {status} "
                              f"({'non-finite inputs or a singular contact geometry' if status == _lib.NUMERICAL else 'more stance contacts in a step than max_contacts_per_step'}); "
                              "the kernel returned zero forces")
        if status == _lib.MAX_ITER:
            warnings.warn(f"MPC solve stopped at the iteration cap ({iters} iterations): best iterate returned", RuntimeWarning, stacklevel=stacklevel)

    def prepare(self, contact_horizon: Sequence, c_horizon: Sequence, p_com_horizon=None, x_predicted=None):
        """Two-phase form of update() for loops that know the contact schedule, the contact points and x_ref_hor before the state
        estimate arrives (NOT the reference's call pattern, which measures the feet with the state: run_simulation.py:94-97):
        the factorisation runs now, asynchronously; update_prepared(x_current) then only patches the gradient and iterates.
        x_predicted: any finite guess of the state (default: self.x0)."""
        if self._engine is None:
            self.init_matrices()
        eng, N = self._engine, self.HORIZON_LENGTH
        st = eng.stage()
        st["x0"][0] = np.asarray(self.x0 if x_predicted is None else x_predicted, dtype=np.float64).reshape(NX)
        st["x_ref"][0] = np.asarray(self.x_ref_hor, dtype=np.float64).reshape(N, NX)
        st["foot"][0] = np.asarray(c_horizon, dtype=np.float64).reshape(N, NU)
        st["contact"][0] = np.asarray(contact_horizon).reshape(N, NC) != 0
        if p_com_horizon is not None:
            st["pcom"][0] = np.asarray(p_com_horizon, dtype=np.float64).reshape(N, 3)
        eng.prepare_staged(1, use_pcom=p_com_horizon is not None)

    def update_prepared(self, x_current=None, one_rollout: bool = True):
        """Second phase of prepare(): returns what update() returns, for the measured state x_current (default: self.x0)."""
        eng = self._engine
        st = eng.stage()
        self._last_fast = self._last_fc = False
        st["x0"][0] = np.asarray(self.x0 if x_current is None else x_current, dtype=np.float64).reshape(NX)
        t0 = time.perf_counter()
        eng.solve_prepared(1, want_x=True)
        self._solve_time = time.perf_counter() - t0
        self._status, self._iters = int(st["status"][0]), int(st["iters"][0])
        if self._status != _lib.SOLVED:
            self._not_solved(self._status, self._iters, 3)
        self._u_opt, self._x_opt = st["u"][0].copy(), st["x"][0].copy()
        return self._u_opt[0].reshape(NU, 1).copy(), (self._x_opt.copy() if one_rollout else self._x_opt[:2].copy())

    def update(self, contact_horizon: Sequence, c_horizon: Sequence, p_com_horizon, x_current=None,
               one_rollout: bool = True):
        """run_simulation.py:106.  Returns (u_opt0 (12,1), x_opt1) where x_opt1[1] is the next state.
        one_rollout=True -> x_opt1 has the whole roll-out (N+1, 13); False -> only rows 0..1.

        One C call (srbdqp_update_f64) with every argument bound once: the inputs go straight into the library's pinned staging arrays,
        the results are copied out of them once.  What Python adds to the C call is what NumPy needs to gather the reference's per-step
        lists (two np.concatenate of N small arrays: ~2.4 us of ~4.5 us in total); (N, 12) / (N, 4) arrays instead of lists cost
        ~2 us less."""
        upd = self._upd
        if upd is None:
            upd = self._bind()
        fcb = self._fcb
        if fcb is not None:                        # csrc/fastcall.c: the lists walked with the C API, straight into the staging arrays, results as fresh arrays
            r = _fastcall.update(fcb, contact_horizon, c_horizon, p_com_horizon, self.x0 if x_current is None else x_current, self.x_ref_hor, one_rollout)
            if r is not NotImplemented:            # (anything it does not recognise -- other dtypes, shapes, nested lists -- takes the NumPy path below)
                u0, x1, st, rc = r
                self._last_fast = self._last_fc = True
                if rc:
                    _lib.check(rc, self._engine._h)
                if st != 1:                        # (_lib.SOLVED)
                    self._not_solved(st, int(self._s_iters[0]), 3)
                return u0, x1
        self._last_fc = False
        try:
            x_cur = self.x0 if x_current is None else x_current
            try:
                self._s_x0c[...] = x_cur
            except ValueError:                     # not a (13, 1) column: (13,) and the like
                self._s_x0[:] = x_cur.reshape(NX)
            self._s_xref[:] = self.x_ref_hor
            if type(c_horizon) is np.ndarray:
                self._s_foot2[:] = c_horizon
            else:                                  # the reference passes per-step lists (run_simulation.py:94-101)
                np.concatenate(c_horizon, out=self._s_foot)
            if type(contact_horizon) is np.ndarray:
                self._s_ct2[:] = contact_horizon
            else:
                np.concatenate(contact_horizon, out=self._s_ct, casting="unsafe")   # 0 / non-zero flags
            if p_com_horizon is None:
                args = self._args_nopcom
            else:
                self._s_pcom[:] = p_com_horizon
                args = self._args_pcom
        except (ValueError, TypeError, AttributeError):   # anything irregular (nested lists, other shapes): the general path
            return self._update_general(contact_horizon, c_horizon, p_com_horizon, x_current, one_rollout)
        t0 = time.perf_counter()
        rc = upd(*args)
        self._solve_time = time.perf_counter() - t0
        self._last_fast = True
        if rc:
            _lib.check(rc, self._engine._h)
        if self._s_status[0] != 1:                 # (_lib.SOLVED)
            self._not_solved(int(self._s_status[0]), int(self._s_iters[0]), 3)
        return self._s_u0.copy(), (self._s_x.copy() if one_rollout else self._s_x01.copy())

    def _update_general(self, contact_horizon, c_horizon, p_com_horizon, x_current, one_rollout):
        x_cur = self.x0 if x_current is None else x_current
        u, x = self.solve(x_cur, self.x_ref_hor, c_horizon, contact_horizon, p_com_horizon)
        u_opt0 = u[0].reshape(NU, 1).copy()
        x_opt1 = x.copy() if one_rollout else x[:2].copy()
        return u_opt0, x_opt1

    def close(self):
        self._upd = None
        self._last_fast = self._last_fc = False
        if self._engine is not None:
            self._engine.close()
            self._engine = None
