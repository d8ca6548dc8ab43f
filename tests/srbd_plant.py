"""Test helper: a headless single-rigid-body plant (nonlinear attitude dynamics, RK4) standing in for MuJoCo + WBID
(config 1 is plumbing: there is no ROS / MuJoCo in this environment), and an oracle-backed MPC object with the
reference surface so that the message adapter can be exercised on CPU."""
import numpy as np

import srbd_oracle as orc


def euler_rate_matrix(rpy):
    r, p, _ = rpy
    cr, sr, cp, tp = np.cos(r), np.sin(r), np.cos(p), np.tan(p)
    return np.array([[1, sr * tp, cr * tp], [0, cr, -sr], [0, sr / cp, cr / cp]])


def rot_zyx(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


class SrbdPlant:
    """x = [rpy, com, omega_world, v, g]; forces at fixed world points (feet on the ground)."""

    def __init__(self, params: orc.SrbdParams):
        self.p = params

    def deriv(self, x, feet, u):
        R = rot_zyx(x[0:3])
        Iw = R @ np.diag(self.p.inertia) @ R.T
        w = x[6:9]
        F = u.reshape(4, 3)
        tau = sum(np.cross(feet[i] - x[3:6], F[i]) for i in range(4))
        dx = np.zeros(13)
        dx[0:3] = euler_rate_matrix(x[0:3]) @ (R.T @ w)          # body rates -> Euler rates
        dx[3:6] = x[9:12]
        dx[6:9] = np.linalg.solve(Iw, tau - np.cross(w, Iw @ w))
        dx[9:12] = F.sum(0) / self.p.mass + np.array([0, 0, x[12]])
        return dx

    def step(self, x, feet, u, h):
        k1 = self.deriv(x, feet, u); k2 = self.deriv(x + 0.5 * h * k1, feet, u)
        k3 = self.deriv(x + 0.5 * h * k2, feet, u); k4 = self.deriv(x + h * k3, feet, u)
        return x + h / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)


class OracleMPC:
    """An object with the reference's MPC surface whose update() runs the CPU oracle (tests only)."""

    def __init__(self, dt=0.04, horizon=10, **kw):
        self.dt, self.HORIZON_LENGTH, self.g = dt, horizon, -9.80665
        self.x0 = np.zeros((13, 1)); self.x0[12] = self.g
        self.x_ref_hor = np.zeros((horizon, 13)); self.x_ref_hor[:, 12] = self.g
        self.params = orc.SrbdParams(dt=dt, **kw)

    def init_matrices(self):
        return self

    def update(self, contact_horizon, c_horizon, p_com_horizon, x_current=None, one_rollout=True):
        x0 = (self.x0 if x_current is None else np.asarray(x_current)).reshape(13)
        r = orc.update(self.params, x0, self.x_ref_hor, np.asarray(c_horizon), np.asarray(contact_horizon), pcom_hor=p_com_horizon)
        self.status, self.iters = r["status"], r["iters"]
        return r["u"][0].reshape(12, 1), (r["x"] if one_rollout else r["x"][:2])
