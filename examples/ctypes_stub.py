"""The smallest ctypes binding of include/srbdqp.h a maintainer of ioloizou/g1_locomotion would write to put the engine behind
`MPC.update()` (g1_mujoco_sim/src/run_simulation.py:106,111).  INTEGRATION.md section 2 shows this file verbatim;
tests/test_cabi_cpu.py imports it (struct layout against the library) and tests/test_gpu_parity.py solves a QP through it.
It depends on nothing but the shared library -- g1_locomotion_amd/_lib.py is the full binding.
"""
import ctypes as C
import os

import numpy as np

LIB = os.environ.get("SRBDQP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "g1_locomotion_amd", "libsrbdqp.so")


class Config(C.Structure):            # struct srbdqp_config, include/srbdqp.h (field for field, in order)
    _fields_ = [("struct_size", C.c_int32), ("horizon", C.c_int32), ("device", C.c_int32), ("flags", C.c_int32),
                ("kernel", C.c_int32), ("max_iter", C.c_int32), ("check_every", C.c_int32),
                ("max_contacts_per_step", C.c_int32), ("rho_restart_iter", C.c_int32), ("rho_restart_count", C.c_int32),
                ("dt", C.c_double), ("mass", C.c_double), ("inertia", C.c_double * 3), ("mu", C.c_double),
                ("fz_min", C.c_double), ("fz_max", C.c_double), ("q_diag", C.c_double * 13), ("r_diag", C.c_double),
                ("force_scale", C.c_double), ("rho", C.c_double), ("rho_eq_scale", C.c_double),   # rho = 0: automatic (0.7)
                ("sigma", C.c_double), ("alpha", C.c_double), ("eps_abs", C.c_double), ("eps_rel", C.c_double),
                ("rho_fz_scale", C.c_double)]                                                      # 0: automatic (4)


def load(path=LIB):
    lib = C.CDLL(path)
    lib.srbdqp_last_error.restype = C.c_char_p
    lib.srbdqp_last_error.argtypes = [C.c_void_p]
    lib.srbdqp_update_f64.argtypes = [C.c_void_p] + [C.c_void_p] * 10
    return lib


def default_config(lib):
    cfg = Config()
    cfg.struct_size = C.sizeof(Config)              # the library checks this BEFORE it writes: a stale stub is refused, not overrun
    if lib.srbdqp_default_config(C.byref(cfg)) != 0:
        raise RuntimeError(lib.srbdqp_last_error(None).decode())
    return cfg


class MPC:
    """MPC(dt) + init_matrices() + update(): the three calls of run_simulation.py:169-170,106, with the attributes the caller fills
    between them (x0 (13, 1), x_ref_hor (HORIZON_LENGTH, 13), g: run_simulation.py:73-82,96,103)."""

    def __init__(self, dt=0.04, lib=None):
        self.lib = lib or load()
        self.cfg = default_config(self.lib)
        self.cfg.dt = dt
        self.HORIZON_LENGTH = self.cfg.horizon
        self.g = -9.80665
        self.x0 = np.zeros((13, 1)); self.x0[12] = self.g
        self.x_ref_hor = np.zeros((self.HORIZON_LENGTH, 13)); self.x_ref_hor[:, 12] = self.g
        self.h = C.c_void_p()

    def init_matrices(self):
        if self.lib.srbdqp_create(C.byref(self.cfg), C.byref(self.h)) != 0:      # no GPU: SRBDQP_E_NO_DEVICE, there is no CPU fallback
            raise RuntimeError(self.lib.srbdqp_last_error(None).decode())

    def update(self, contact_horizon, c_horizon, p_com_horizon, x_current=None, one_rollout=True):
        """the reference's call, run_simulation.py:106: update(contact_horizon, c_horizon, p_com_horizon, x_current=MPC.x0, one_rollout=True)"""
        N = self.HORIZON_LENGTH
        x0 = np.ascontiguousarray(self.x0 if x_current is None else x_current, np.float64).reshape(13)
        xr = np.ascontiguousarray(self.x_ref_hor, np.float64).reshape(N, 13)
        ft = np.ascontiguousarray(c_horizon, np.float64).reshape(N, 12)
        ct = np.ascontiguousarray(np.asarray(contact_horizon) != 0, np.uint8).reshape(N, 4)
        pc = None if p_com_horizon is None else np.ascontiguousarray(p_com_horizon, np.float64).reshape(N, 3)
        u0, x = np.empty(12), np.empty((N + 1, 13))
        st, it = C.c_int32(), C.c_int32()
        p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        rc = self.lib.srbdqp_update_f64(self.h, p(x0), p(xr), p(ft), p(ct), p(pc), p(u0), None, p(x), C.addressof(st), C.addressof(it))
        if rc != 0:
            raise RuntimeError(self.lib.srbdqp_last_error(self.h).decode())
        self.status, self.iters = st.value, it.value                              # 1 solved, 2 iteration cap, < 0 failed (forces 0)
        return u0.reshape(12, 1), (x if one_rollout else x[:2])                   # u_opt0, x_opt1 (row 1 = the next state)

    def close(self):
        if self.h:
            self.lib.srbdqp_destroy(self.h)
            self.h = C.c_void_p()
