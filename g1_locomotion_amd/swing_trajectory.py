"""Drop-in for the reference's ``swing_trajectory`` module (g1_mujoco_sim/src/swing_trajectory.py): same class, same
method names and return types, evaluated by the batched HIP kernel behind ``srbdqp_swing_f64``
(include/srbdqp_cascade.h).  The reference evaluates one foot at one instant per call
(ros_run_simulation.py:246-256); the batched entry point for many feet / instants is ``BatchMPC.swing``.

    from g1_locomotion_amd import swing_trajectory          # instead of: import swing_trajectory
    traj = swing_trajectory.SwingTrajectory()

There is no CPU fallback: every getter is one C-ABI call on the GPU.  Plotting (plot_trajectory) is not reproduced.
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
        """swing_trajectory.py:38-52.  The getters below always use the coefficients of the CURRENT positions; this
        call publishes them in ``self.coeff`` as the reference does."""
        self.coeff = self._eval([0.0], want_coeff=True)["coeff"][0].copy()

    def calculate_position_xy(self, cycle_progress):
        pos = self._eval([cycle_progress])["pos"][0]
        return float(pos[0]), float(pos[1])

    def calculate_position_z(self, t):
        return float(self._eval([t])["pos"][0, 2])

    def calculate_velocity_z(self, t):
        return float(self._eval([t])["vel_z"][0])

    def calculate_acceleration_z(self, t):
        return float(self._eval([t])["acc_z"][0])

    def calculate_trajectory_xy(self):
        """100 samples over the cycle (swing_trajectory.py:69-74): list of (x, y)."""
        pos = self._eval(np.linspace(0, 1, 100))["pos"]
        return [(float(p[0]), float(p[1])) for p in pos]

    def calculate_all_trajectories_z(self):
        """100 samples of z, z', z'' (swing_trajectory.py:91-104): three lists."""
        r = self._eval(np.linspace(0, 1, 100))
        return list(r["pos"][:, 2]), list(r["vel_z"]), list(r["acc_z"])
