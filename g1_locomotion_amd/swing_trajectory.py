"""Drop-in for the reference's ``swing_trajectory`` module (g1_mujoco_sim/src/swing_trajectory.py): same class, same
method names and return types.  The reference evaluates one foot at one instant per call, about three times per 1 ms
simulation step (ros_run_simulation.py:246-256): those scalar getters are plain host arithmetic on the coefficients the
last ``calculate_coeff()`` cached -- as in the reference, where they read ``self.coeff`` -- because a kernel launch per
scalar (4 H2D copies + launch + 4 D2H + sync) is slower than the seven multiply-adds it replaces.  The batched evaluation
for many feet / instants is the HIP kernel behind ``srbdqp_swing_f64`` (include/srbdqp_cascade.h): ``BatchMPC.swing`` and
the two 100-sample curve methods below; tests/test_gpu_cascade.py holds the two to the same golden vectors.

    from g1_locomotion_amd import swing_trajectory          # instead of: import swing_trajectory
    traj = swing_trajectory.SwingTrajectory()

Plotting (plot_trajectory) is not reproduced.
"""
from typing import Optional

import numpy as np

from .mpc import BatchMPC

_engine: Optional[BatchMPC] = None


def _shared_engine() -> BatchMPC:
    global _engine
    if _engine is None:
        _engine = BatchMPC(horizon=10)
    return _engine


class SwingTrajectory:
    """Sixth-order polynomial on z (zero velocity/acceleration at both ends except a small downward landing velocity),
    sine-then-linear interpolation on x, y (swing_trajectory.py:5-13)."""

    FINAL_VELOCITY_Z = -0.02       # swing_trajectory.py:50
    FIRST_HALF_SHARE = 0.80        # swing_trajectory.py:58

    def __init__(self, engine: Optional[BatchMPC] = None):
        self._eng = engine
        self.reset()

    # -- setters (swing_trajectory.py:27-36) ---------------------------------------------------------------
    def set_positions_xy(self, p_x_start, p_x_final, p_y_start, p_y_final):
        self.p_x_start, self.p_x_final = float(p_x_start), float(p_x_final)
        self.p_y_start, self.p_y_final = float(p_y_start), float(p_y_final)

    def set_positions_z(self, p_z_start, p_z_middle, p_z_final):
        self.p_z_start, self.p_z_final, self.p_z_middle = float(p_z_start), float(p_z_final), float(p_z_middle)

    def reset(self):
        self.p_z_start = self.p_z_middle = self.p_z_final = 0.0
        self.p_x_start = self.p_x_final = self.p_y_start = self.p_y_final = 0.0
        self.coeff = np.zeros(7)

    # -- evaluation -------------------------------------------------------------------------------------
    def _eval(self, ts, want_coeff=False):
        eng = self._eng or _shared_engine()
        ts = np.atleast_1d(np.asarray(ts, dtype=np.float64))
        B = ts.shape[0]
        ps = np.tile([self.p_x_start, self.p_y_start, self.p_z_start], (B, 1))
        pf = np.tile([self.p_x_final, self.p_y_final, self.p_z_final], (B, 1))
        return eng.swing(ps, pf, np.full(B, self.p_z_middle), ts, self.FINAL_VELOCITY_Z, self.FIRST_HALF_SHARE, want_coeff=want_coeff)

    def calculate_coeff(self):
        """swing_trajectory.py:38-52: closed-form coefficients of the sixth-order polynomial through (0, z_start), (0.5,
        z_middle), (1, z_final) with zero end velocities / accelerations except the landing velocity.  Cached in
        ``self.coeff``; the z getters read them, as in the reference."""
        zs, zm, zf, vf = self.p_z_start, self.p_z_middle, self.p_z_final, self.FINAL_VELOCITY_Z
        # the reference solves a constant 7 x 7 system; its inverse applied to [z_s, 0, 0, z_m, z_f, v_f, 0] is these integer
        # combinations (the same ones the kernel uses, srbdqp_cascade.hpp swing_coefficients())
        self.coeff = np.array([zs, 0.0, 0.0,
                               -42.0 * zs + 64.0 * zm - 22.0 * zf + 6.0 * vf,
                               111.0 * zs - 192.0 * zm + 81.0 * zf - 23.0 * vf,
                               -102.0 * zs + 192.0 * zm - 90.0 * zf + 27.0 * vf,
                               32.0 * zs - 64.0 * zm + 32.0 * zf - 10.0 * vf])

    def calculate_position_xy(self, cycle_progress):
        """swing_trajectory.py:54-67: a sine covers FIRST_HALF_SHARE of the distance in the first half of the cycle, the rest
        is linear."""
        t, share = float(cycle_progress), self.FIRST_HALF_SHARE
        phase = share * np.sin(np.pi * t) if t <= 0.5 else share + (t - 0.5) * (1.0 - share) * 2.0
        return (float((1.0 - phase) * self.p_x_start + phase * self.p_x_final),
                float((1.0 - phase) * self.p_y_start + phase * self.p_y_final))

    def calculate_position_z(self, t):
        c = self.coeff
        return float(c[0] + t * (c[1] + t * (c[2] + t * (c[3] + t * (c[4] + t * (c[5] + t * c[6]))))))

    def calculate_velocity_z(self, t):
        c = self.coeff
        return float(c[1] + t * (2 * c[2] + t * (3 * c[3] + t * (4 * c[4] + t * (5 * c[5] + t * 6 * c[6])))))

    def calculate_acceleration_z(self, t):
        c = self.coeff
        return float(2 * c[2] + t * (6 * c[3] + t * (12 * c[4] + t * (20 * c[5] + t * 30 * c[6]))))

    def calculate_trajectory_xy(self):
        """100 samples over the cycle (swing_trajectory.py:69-74): list of (x, y).  Batched: the HIP kernel."""
        pos = self._eval(np.linspace(0, 1, 100))["pos"]
        return [(float(p[0]), float(p[1])) for p in pos]

    def calculate_all_trajectories_z(self):
        """100 samples of z, z', z'' (swing_trajectory.py:91-104): three lists.  Batched: the HIP kernel."""
        r = self._eval(np.linspace(0, 1, 100))
        return list(r["pos"][:, 2]), list(r["vel_z"]), list(r["acc_z"])
