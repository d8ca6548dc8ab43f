"""Seeded synthetic inputs of the SRBD convex-MPC hot path: what bench.py times the engine on and
what the parity tests feed to both sides.  Product-side data -- plain NumPy, no solver code; the oracle re-exports it.

Constants with in-tree evidence: CoM target (g1_mujoco_sim/src/run_simulation.py:81), hip offset and contact geometry
(g1_description/g1_23dof.urdf:86,285-294), swing / support switch every 0.25 s (ros_run_simulation.py:148).
"""
import numpy as np

NX, NU, NC = 13, 12, 4
COM_TARGET = np.array([5.26790425e-02, 7.44339342e-05, 5.97983255e-01])   # run_simulation.py:81
HIP_Y = 0.064452                                                          # g1_23dof.urdf hip offset
HEEL_X, TOE_X = -0.05, 0.12                                               # g1_23dof.urdf:285-294


def synthetic_batch(B, N, seed, schedule="single", dt=0.04, gravity=-9.80665):
    """Seeded inputs: x0 (B,13), x_ref (B,N,13), foot (B,N,12), contact (B,N,4) uint8.

    schedule: "single" = alternating single support, switch every 6 steps, random phase & first foot
              "double" = all four points active
              "mixed"  = per-QP random phase gait with double-support overlap
    """
    rng = np.random.default_rng(seed)
    x0 = np.zeros((B, NX))
    x0[:, 0:2] = rng.uniform(-0.2, 0.2, (B, 2))
    x0[:, 2] = rng.uniform(-np.pi, np.pi, B)
    x0[:, 3:5] = rng.uniform(-0.1, 0.1, (B, 2))
    x0[:, 5] = 0.598 + rng.uniform(-0.05, 0.05, B)
    x0[:, 6:9] = rng.uniform(-0.5, 0.5, (B, 3))
    x0[:, 9:12] = rng.uniform(-0.5, 0.5, (B, 3))
    x0[:, 12] = gravity
    v_ref = rng.uniform(-0.3, 0.3, (B, 2))
    x_ref = np.zeros((B, N, NX))
    k = np.arange(1, N + 1)[None, :]
    x_ref[:, :, 2] = x0[:, 2:3]                                   # hold current yaw
    x_ref[:, :, 3] = COM_TARGET[0] + v_ref[:, 0:1] * k * dt
    x_ref[:, :, 4] = COM_TARGET[1] + v_ref[:, 1:2] * k * dt
    x_ref[:, :, 5] = COM_TARGET[2]
    x_ref[:, :, 9] = v_ref[:, 0:1]
    x_ref[:, :, 10] = v_ref[:, 1:2]
    x_ref[:, :, 12] = gravity
    foot = np.zeros((B, N, NU))
    fx0 = rng.uniform(-0.05, 0.05, (B, 2))
    fy = np.stack([HIP_Y + rng.uniform(-0.02, 0.02, B), -HIP_Y + rng.uniform(-0.02, 0.02, B)], 1)
    c, s = np.cos(x0[:, 2]), np.sin(x0[:, 2])
    for f in range(2):
        for h, off in enumerate((HEEL_X, TOE_X)):
            px, py = fx0[:, f] + off, fy[:, f]
            i = 2 * f + h
            foot[:, :, 3 * i + 0] = (COM_TARGET[0] + c * px - s * py)[:, None]
            foot[:, :, 3 * i + 1] = (COM_TARGET[1] + s * px + c * py)[:, None]
            foot[:, :, 3 * i + 2] = 0.0
    contact = np.ones((B, N, NC), np.uint8)
    if schedule in ("single", "mixed"):
        phase = rng.integers(0, 12, B)
        for kk in range(N):
            ph = (phase + kk) % 12
            left_stance = ph < 6
            if schedule == "single":
                contact[:, kk, 0:2] = left_stance[:, None]
                contact[:, kk, 2:4] = ~left_stance[:, None]
            else:
                ds = (ph % 6) == 0                                # one double-support step per switch
                contact[:, kk, 0:2] = (left_stance | ds)[:, None]
                contact[:, kk, 2:4] = (~left_stance | ds)[:, None]
    return x0, x_ref, foot, contact
