"""CPU restatement (NumPy) of the steps either side of the QP that SURVEY.md 8(f) ranks next.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's CPU leg, never by the product.

  * swing-foot trajectory -- follows g1_mujoco_sim/src/swing_trajectory.py:38-89.  PINNED: tests/golden/swing_golden.npz
    holds outputs of the reference class itself (generated in the build container by tests/golden/make_swing_golden.py,
    which imports the reference module); tests/test_cascade_oracle.py checks this restatement against them.
  * MPC -> WBID reference mapping -- follows g1_mujoco_sim/src/wbid.py:232-297.  Parity unpinned: wbid.py cannot be
    imported here (pyopensot, xbot2_interface, tf are absent) and the reference has no tests; the rotation is
    tf.transformations.euler_matrix's published 'sxyz' formula.
  * the step before the QP (gait schedule, landing position, input horizons) -- the schedule is inside the
    absent module: own design, the one g1_locomotion_amd/msgs.py runs on the host for one robot; mpc_inputs() below restates it
    for B robots and tests/test_cascade_oracle.py checks it against what msgs.MpcNode.step hands to MPC.update().
"""
import numpy as np

FINAL_VELOCITY_Z = -0.02      # swing_trajectory.py:50
FIRST_HALF_SHARE = 0.80       # swing_trajectory.py:58
GRAVITY = -9.80665            # wbid.py:286
TORSO_INERTIA = (8.20564e-2, 8.05015e-2, 0.32353e-2)   # wbid.py:262-266


def swing_system():
    """The 7x7 boundary-condition matrix of swing_trajectory.py:40-46 (rows: z(0), z'(0), z''(0), z(1/2), z(1), z'(1), z''(1))."""
    h = 0.5
    return np.array([[1, 0, 0, 0, 0, 0, 0],
                     [0, 1, 0, 0, 0, 0, 0],
                     [0, 0, 2, 0, 0, 0, 0],
                     [h ** k for k in range(7)],
                     [1] * 7,
                     [k for k in range(7)],
                     [k * (k - 1) for k in range(7)]], dtype=np.float64)


def swing_coeff(z_start, z_middle, z_final, final_velocity_z=FINAL_VELOCITY_Z):
    """calculate_coeff (swing_trajectory.py:38-52): coefficients, lowest power first.  Inputs broadcast; returns (..., 7)."""
    zs, zm, zf = np.broadcast_arrays(np.asarray(z_start, np.float64), np.asarray(z_middle, np.float64), np.asarray(z_final, np.float64))
    b = np.zeros(zs.shape + (7,))
    b[..., 0] = zs; b[..., 3] = zm; b[..., 4] = zf; b[..., 5] = final_velocity_z
    return np.linalg.solve(swing_system(), b[..., None])[..., 0]


def swing_phase(t, share=FIRST_HALF_SHARE):
    """calculate_position_xy's interpolation weight (swing_trajectory.py:54-64)."""
    t = np.asarray(t, np.float64)
    return np.where(t <= 0.5, share * np.sin(np.pi * t), share + (t - 0.5) * (1 - share) * 2)


def swing_eval(p_start, p_final, z_middle, t, final_velocity_z=FINAL_VELOCITY_Z, share=FIRST_HALF_SHARE):
    """Batched twin of the SwingTrajectory getters: p_start, p_final (B,3); z_middle, t (B,).
    Returns dict(pos (B,3), vel_z (B,), acc_z (B,), coeff (B,7))."""
    ps, pf = np.asarray(p_start, np.float64).reshape(-1, 3), np.asarray(p_final, np.float64).reshape(-1, 3)
    t = np.asarray(t, np.float64).reshape(-1)
    c = swing_coeff(ps[:, 2], np.asarray(z_middle, np.float64).reshape(-1), pf[:, 2], final_velocity_z)
    ph = swing_phase(t, share)
    k = np.arange(7)
    tp = t[:, None] ** k                                              # [1, t, ..., t^6]            (:76-79)
    tv = np.where(k >= 1, k * t[:, None] ** np.maximum(k - 1, 0), 0.0)              # derivative basis (:81-84)
    ta = np.where(k >= 2, k * (k - 1) * t[:, None] ** np.maximum(k - 2, 0), 0.0)    # second derivative (:86-89)
    pos = np.stack([(1 - ph) * ps[:, 0] + ph * pf[:, 0], (1 - ph) * ps[:, 1] + ph * pf[:, 1], (c * tp).sum(1)], axis=1)
    return dict(pos=pos, vel_z=(c * tv).sum(1), acc_z=(c * ta).sum(1), coeff=c)


def euler_matrix_sxyz(roll, pitch, yaw):
    """tf.transformations.euler_matrix(ai, aj, ak) for the default axes 'sxyz' (static x, y, z = Rz(ak) Ry(aj) Rx(ai));
    the 3x3 rotation block used at wbid.py:247."""
    si, sj, sk = np.sin(roll), np.sin(pitch), np.sin(yaw)
    ci, cj, ck = np.cos(roll), np.cos(pitch), np.cos(yaw)
    cc, cs, sc, ss = ci * ck, ci * sk, si * ck, si * sk
    return np.array([[cj * ck, sj * sc - cs, sj * cc + ss],
                     [cj * sk, sj * ss + cc, sj * cs - sc],
                     [-sj, cj * si, cj * ci]])


def wbid_reference(x_next, u0, foot_positions, mass, inertia=TORSO_INERTIA, as_written=True):
    """One robot, the arithmetic of WBID.setReference (wbid.py:243-296) in its order.  x_next (13,), u0 (12,),
    foot_positions (4,3).  Returns dict(R (3,3), base_vel (6,), base_acc (6,), com_acc (3,), com_pos, com_vel, wrench (4,3))."""
    x = np.asarray(x_next, np.float64).reshape(13)
    u = np.asarray(u0, np.float64).reshape(12)
    feet = np.asarray(foot_positions, np.float64).reshape(4, 3)
    R = euler_matrix_sxyz(x[0], x[1], x[2])                                     # :246-247
    velocity = np.hstack((x[9:12], x[6:9]))                                     # :256-258
    inertia_inv = np.linalg.inv(np.diag(inertia))                               # :262-269
    r = feet - np.tile(x[3:6], (4, 1))                                          # :271
    s = np.zeros((1, 3))
    for i in range(4):                                                          # :273-275
        s = s + np.cross(r[i, :], x[6:9])
    ang_acc = inertia_inv @ s.T                                                 # :279
    base_acc = np.vstack((np.zeros((3, 1)), ang_acc)).reshape(6)                # :278-282
    if as_written:
        sum_forces = np.sum(np.reshape(u, (3, 4)), axis=1)                      # :290, exactly as written
    else:
        sum_forces = np.sum(np.reshape(u, (4, 3)), axis=0)                      # the per-axis sums
    com_acc = sum_forces / mass + np.array([0, 0, GRAVITY])                     # :286,291
    return dict(R=R, base_vel=velocity, base_acc=base_acc, com_acc=com_acc, com_pos=x[3:6].copy(), com_vel=x[9:12].copy(),
                wrench=u.reshape(4, 3).copy())                                  # :294-297


def wbid_reference_batch(x_next, u0, foot, mass, inertia=TORSO_INERTIA, as_written=True):
    xs, us, fs = np.asarray(x_next).reshape(-1, 13), np.asarray(u0).reshape(-1, 12), np.asarray(foot).reshape(-1, 12)
    outs = [wbid_reference(xs[i], us[i], fs[i], mass, inertia, as_written) for i in range(xs.shape[0])]
    return {k: np.stack([o[k] for o in outs]) for k in ("R", "base_vel", "base_acc", "com_acc")}


def mpc_inputs(x0, feet, stamp, v_ref, com_target, N, dt, standing=None, period_steps=6, double_support_steps=1, hip_offset_y=0.0645):
    """The QP inputs of B robots as msgs.MpcNode.step builds them for one (same operations in the same order, so the GPU
    kernel can be compared bit for bit): x0 (B,13), feet (B,12), stamp (B,), v_ref (B,2), standing (B,) or None.
    Returns dict(x_ref (B,N,13), foot (B,N,12), contact (B,N,4) uint8, pcom (B,N,3), landing (B,3))."""
    x0 = np.asarray(x0, np.float64).reshape(-1, 13)
    B = x0.shape[0]
    feet = np.asarray(feet, np.float64).reshape(B, 12)
    stamp = np.asarray(stamp, np.float64).reshape(B)
    v_ref = np.asarray(v_ref, np.float64).reshape(B, 2)
    standing = np.zeros(B, bool) if standing is None else np.asarray(standing).reshape(B) != 0
    com_target = np.asarray(com_target, np.float64)
    k = np.arange(1, N + 1, dtype=np.float64)
    xr = np.zeros((B, N, 13))
    moving = np.any(v_ref != 0.0, axis=1)
    xr[:, :, 2] = x0[:, None, 2]
    xr[:, :, 3] = np.where(moving[:, None], x0[:, None, 3] + v_ref[:, None, 0] * k[None, :] * dt, com_target[0])
    xr[:, :, 4] = np.where(moving[:, None], x0[:, None, 4] + v_ref[:, None, 1] * k[None, :] * dt, com_target[1])
    xr[:, :, 5] = com_target[2]
    xr[:, :, 9] = v_ref[:, None, 0]
    xr[:, :, 10] = v_ref[:, None, 1]
    xr[:, :, 12] = x0[:, None, 12]
    foot = np.repeat(feet[:, None, :], N, axis=1)
    v3 = np.concatenate([v_ref, np.zeros((B, 1))], axis=1)
    pcom = x0[:, None, 3:6] + v3[:, None, :] * (k[None, :, None] - 1.0) * dt
    k0 = np.floor(stamp / dt + 1e-9).astype(np.int64)
    ph = (k0[:, None] + np.arange(N)[None, :]) % (2 * period_steps)
    left = ph < period_steps
    ds = (ph % period_steps) < double_support_steps
    cl = (standing[:, None] | left | ds).astype(np.uint8)
    cr = (standing[:, None] | ~left | ds).astype(np.uint8)
    contact = np.stack([cl, cl, cr, cr], axis=2)
    T = period_steps * dt
    side = np.where(contact[:, 0, 0] == 0, 1.0, -1.0)
    landing = np.zeros((B, 3))
    landing[:, 0] = x0[:, 3] + 0.5 * T * x0[:, 9] + 0.03 * (x0[:, 9] - v_ref[:, 0])
    landing[:, 1] = x0[:, 4] + side * hip_offset_y + 0.5 * T * x0[:, 10] + 0.03 * (x0[:, 10] - v_ref[:, 1])
    return dict(x_ref=xr, foot=foot, contact=contact, pcom=pcom, landing=landing)
