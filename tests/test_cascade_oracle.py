"""SURVEY.md 8(f) rows 3-4 on CPU: the oracle's restatement of the swing-foot trajectory against vectors produced BY THE
REFERENCE ITSELF (tests/golden/swing_golden.npz, see make_swing_golden.py), and first-principles checks of the MPC->WBID
reference mapping (whose reference module cannot be imported anywhere: parity unpinned)."""
import os

import numpy as np
import pytest

import cascade_oracle as co

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "swing_golden.npz")
TOL_SWING = 1e-12      # fp64; the reference solves its 7x7 system by LU, so coefficients differ in the last bits


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_swing_oracle_reproduces_the_reference_outputs(gold):
    r = co.swing_eval(gold["p_start"], gold["p_final"], gold["z_middle"], gold["progress"])
    assert gold["progress"].shape[0] >= 800
    assert np.abs(r["coeff"] - gold["coeff"]).max() < 1e-11
    assert np.abs(r["pos"] - gold["pos"]).max() < TOL_SWING
    assert np.abs(r["vel_z"] - gold["vel_z"]).max() < TOL_SWING
    assert np.abs(r["acc_z"] - gold["acc_z"]).max() < 1e-11


def test_swing_reference_curves(gold):
    t = np.linspace(0, 1, 100)
    r = co.swing_eval(np.tile([0.0, 0.1, 0.0], (100, 1)), np.tile([0.2, 0.1, 0.0], (100, 1)), np.full(100, 0.05), t)
    assert np.abs(r["pos"][:, 2] - gold["curve_z"]).max() < TOL_SWING
    assert np.abs(r["vel_z"] - gold["curve_vz"]).max() < TOL_SWING and np.abs(r["acc_z"] - gold["curve_az"]).max() < 1e-11
    assert np.abs(r["pos"][:, :2] - gold["curve_xy"]).max() < TOL_SWING


def test_swing_boundary_conditions_and_constant_inverse():
    """The polynomial meets the seven conditions it is defined by, and the integer inverse the HIP kernel uses is exact."""
    zs, zm, zf = 0.01, 0.07, -0.005
    c = co.swing_coeff(zs, zm, zf)
    k = np.arange(7)
    val = lambda t: (c * t ** k).sum()
    d1 = lambda t: (c[1:] * k[1:] * t ** (k[1:] - 1)).sum()
    d2 = lambda t: (c[2:] * k[2:] * (k[2:] - 1) * t ** (k[2:] - 2)).sum()
    assert abs(val(0) - zs) < 1e-14 and abs(d1(0.0)) < 1e-14 and abs(d2(0.0)) < 1e-13
    assert abs(val(0.5) - zm) < 1e-13 and abs(val(1.0) - zf) < 1e-12 and abs(d1(1.0) + 0.02) < 1e-12 and abs(d2(1.0)) < 1e-11
    inv = np.linalg.inv(co.swing_system())
    cols = np.array([[1, 0, 0, -42, 111, -102, 32], [0, 0, 0, 64, -192, 192, -64], [0, 0, 0, -22, 81, -90, 32], [0, 0, 0, 6, -23, 27, -10]], dtype=float).T
    assert np.abs(inv[:, [0, 3, 4, 5]] - cols).max() < 1e-10
    # x-y weight: continuous at the half cycle, ends at 1
    assert abs(co.swing_phase(0.5) - 0.8) < 1e-15 and abs(co.swing_phase(1.0) - 1.0) < 1e-15 and co.swing_phase(0.0) == 0.0


def test_wbid_reference_first_principles():
    rng = np.random.default_rng(5)
    x = rng.uniform(-0.5, 0.5, 13); x[12] = -9.80665
    u = rng.uniform(-50, 200, 12)
    feet = rng.uniform(-0.2, 0.2, (4, 3))
    mass = 34.13385728
    r = co.wbid_reference(x, u, feet, mass)
    R = r["R"]
    assert np.abs(R @ R.T - np.eye(3)).max() < 1e-14 and abs(np.linalg.det(R) - 1) < 1e-14
    # 'sxyz' = rotate about fixed x, then fixed y, then fixed z
    cx, sx, cy, sy, cz, sz = np.cos(x[0]), np.sin(x[0]), np.cos(x[1]), np.sin(x[1]), np.cos(x[2]), np.sin(x[2])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]); Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    assert np.abs(R - Rz @ Ry @ Rx).max() < 1e-15
    assert np.array_equal(r["base_vel"], np.r_[x[9:12], x[6:9]]) and np.all(r["base_acc"][:3] == 0)
    want = np.linalg.inv(np.diag(co.TORSO_INERTIA)) @ np.cross(feet - x[3:6], x[6:9]).sum(0)
    assert np.abs(r["base_acc"][3:] - want).max() < 1e-9 * max(1.0, np.abs(want).max())
    # the force sum as written (wbid.py:290) vs per axis: they agree only on the total of all twelve entries
    a_w = co.wbid_reference(x, u, feet, mass, as_written=True)["com_acc"]
    a_c = co.wbid_reference(x, u, feet, mass, as_written=False)["com_acc"]
    assert np.abs(a_w - (np.array([u[0:4].sum(), u[4:8].sum(), u[8:12].sum()]) / mass + [0, 0, -9.80665])).max() < 1e-12
    assert np.abs(a_c - (u.reshape(4, 3).sum(0) / mass + [0, 0, -9.80665])).max() < 1e-12
    assert abs((a_w - a_c).sum()) < 1e-10 and np.abs(a_w - a_c).max() > 1e-3


def test_swing_drop_in_scalar_getters_run_on_the_host_from_cached_coefficients(gold):
    """The class the simulator calls ~3 x per 1 ms step (ros_run_simulation.py:246-256): its scalar getters are host
    arithmetic on the coefficients cached by calculate_coeff() -- no GPU, no engine -- and reproduce the reference's outputs."""
    from g1_locomotion_amd import swing_trajectory
    s = swing_trajectory.SwingTrajectory(engine=None)
    idx = np.linspace(0, gold["progress"].shape[0] - 1, 60).astype(int)
    for i in idx:
        ps, pf, zm, t = gold["p_start"][i], gold["p_final"][i], float(gold["z_middle"][i]), float(gold["progress"][i])
        s.reset()
        s.set_positions_xy(ps[0], pf[0], ps[1], pf[1]); s.set_positions_z(ps[2], zm, pf[2]); s.calculate_coeff()
        assert np.abs(s.coeff - gold["coeff"][i]).max() < 1e-11
        x, y = s.calculate_position_xy(t)
        assert abs(x - gold["pos"][i, 0]) < TOL_SWING and abs(y - gold["pos"][i, 1]) < TOL_SWING
        assert abs(s.calculate_position_z(t) - gold["pos"][i, 2]) < TOL_SWING
        assert abs(s.calculate_velocity_z(t) - gold["vel_z"][i]) < TOL_SWING and abs(s.calculate_acceleration_z(t) - gold["acc_z"][i]) < 1e-11
    # stale coefficients are used until calculate_coeff() is called again, as in the reference
    z_before = s.calculate_position_z(0.3)
    s.set_positions_z(0.5, 0.9, 0.5)
    assert s.calculate_position_z(0.3) == z_before


def test_mpc_inputs_oracle_equals_what_the_mpc_node_hands_to_update():
    """the batched restatement of the step before the QP (gait schedule, landing position, input horizons)
    against the host node (msgs.MpcNode.step + msgs.AlternatingGait) robot by robot -- bit for bit: standing / walking,
    zero / non-zero commanded velocity, stamps spread over several gait periods."""
    from g1_locomotion_amd import msgs

    class Recorder:
        def __init__(self, N=10, dt=0.04):
            self.HORIZON_LENGTH, self.dt, self.g = N, dt, -9.80665
            self.x0, self.x_ref_hor, self.calls = np.zeros((13, 1)), np.zeros((N, 13)), []

        def update(self, ch, c_h, pc, x_current=None, one_rollout=True):
            self.calls.append((np.array(ch), np.array(c_h), np.array(pc), self.x_ref_hor.copy()))
            return np.zeros(12), np.zeros((2, 13))

    rng = np.random.default_rng(0)
    com = np.array([0.05268, 7.44e-5, 0.59798])
    for vref in ((0.0, 0.0), (0.3, -0.1)):
        for standing in (False, True):
            m = Recorder()
            node = msgs.MpcNode(m, msgs.AlternatingGait(dt=0.04, standing=standing), com_target=com, v_ref=vref)
            B = 9
            x0 = rng.normal(size=(B, 13)) * 0.1
            x0[:, 12] = -9.80665
            x0 = np.stack([msgs.state_to_vec(msgs.vec_to_state(x)) for x in x0])      # what survives the message's field types
            feet, st = rng.normal(size=(B, 12)), rng.uniform(0.0, 3.0, size=B)
            outs = [node.step(msgs.make_srbd_current(x0[b], feet[b].reshape(4, 3), np.zeros(12), stamp=float(st[b]))) for b in range(B)]
            o = co.mpc_inputs(x0, feet, st, np.tile(vref, (B, 1)), com, 10, 0.04, standing=np.full(B, standing))
            for b in range(B):
                ch, c_h, pc, xr = m.calls[b]
                assert np.array_equal(ch, o["contact"][b]) and np.array_equal(c_h, o["foot"][b])
                assert np.array_equal(pc, o["pcom"][b]) and np.array_equal(xr, o["x_ref"][b])
                lp = outs[b].landing_position
                assert np.array_equal(np.array([lp.x, lp.y, lp.z]), o["landing"][b])
                assert [c.active for c in outs[b].contacts] == [bool(v) for v in o["contact"][b, 0]]
