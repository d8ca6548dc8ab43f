"""GPU parity of the cascade kernels (include/srbdqp_cascade.h) through the C-ABI: swing-foot trajectory against vectors
produced by the reference itself, and both kernels against the CPU oracle on seeded batches.  Tolerance: fp64, 1e-12
absolute on positions / velocities / rotation entries (inputs are O(1)), 1e-10 relative on accelerations."""
import os

import numpy as np
import pytest

import cascade_oracle as co

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "swing_golden.npz")


@pytest.fixture(scope="module")
def eng():
    import torch  # noqa: F401
    from g1_locomotion_amd import BatchMPC
    e = BatchMPC(horizon=10)
    yield e
    e.close()


def test_swing_kernel_reproduces_the_reference_outputs(eng):
    g = np.load(GOLD)
    r = eng.swing(g["p_start"], g["p_final"], g["z_middle"], g["progress"], want_coeff=True)
    assert eng.kernel_name() == "swing_f64"
    assert np.abs(r["coeff"] - g["coeff"]).max() < 1e-11
    assert np.abs(r["pos"] - g["pos"]).max() < 1e-12 and np.abs(r["vel_z"] - g["vel_z"]).max() < 1e-12
    assert np.abs(r["acc_z"] - g["acc_z"]).max() < 1e-11


def test_swing_drop_in_class_matches_reference_curves(eng):
    from g1_locomotion_amd import swing_trajectory
    g = np.load(GOLD)
    s = swing_trajectory.SwingTrajectory(engine=eng)
    s.reset()
    s.set_positions_xy(0.0, 0.2, 0.1, 0.1); s.set_positions_z(0.0, 0.05, 0.0); s.calculate_coeff()
    assert np.abs(s.coeff - np.array([0, 0, 0, 3.08, -9.14, 9.06, -3.0])).max() < 1e-12     # SURVEY.md 8c's check value
    pz, vz, az = s.calculate_all_trajectories_z()
    assert len(pz) == 100 and np.abs(np.array(pz) - g["curve_z"]).max() < 1e-12
    assert np.abs(np.array(vz) - g["curve_vz"]).max() < 1e-12 and np.abs(np.array(az) - g["curve_az"]).max() < 1e-11
    assert np.abs(np.array(s.calculate_trajectory_xy()) - g["curve_xy"]).max() < 1e-12
    x, y = s.calculate_position_xy(0.25)
    assert abs(x - 0.2 * 0.8 * np.sin(np.pi * 0.25)) < 1e-15 and y == pytest.approx(0.1, abs=1e-15)
    assert abs(s.calculate_position_z(0.5) - 0.05) < 1e-13 and abs(s.calculate_velocity_z(1.0) + 0.02) < 1e-12


def test_swing_large_batch_vs_oracle_and_edges(eng):
    rng = np.random.default_rng(11)
    B = 200_003                                           # not a multiple of the workgroup or grid size
    ps, pf = rng.uniform(-0.5, 0.5, (B, 3)), rng.uniform(-0.5, 0.5, (B, 3))
    zm, t = rng.uniform(0.0, 0.2, B), rng.uniform(0.0, 1.0, B)
    t[:4] = [0.0, 0.5, 1.0, np.nextafter(0.5, 1.0)]
    r = eng.swing(ps, pf, zm, t, want_coeff=True)
    o = co.swing_eval(ps, pf, zm, t)
    assert np.abs(r["pos"] - o["pos"]).max() < 1e-12 and np.abs(r["vel_z"] - o["vel_z"]).max() < 1e-11
    assert np.abs(r["acc_z"] - o["acc_z"]).max() < 1e-10 and np.abs(r["coeff"] - o["coeff"]).max() < 1e-10
    assert np.abs(r["pos"][0] - ps[0]).max() < 1e-15 and np.abs(r["pos"][2] - pf[2]).max() < 1e-12      # t = 0, t = 1
    assert eng.swing(np.zeros((0, 3)), np.zeros((0, 3)), np.zeros(0), np.zeros(0))["pos"].shape == (0, 3)   # empty batch


def test_wbid_reference_kernel_vs_oracle(eng):
    rng = np.random.default_rng(12)
    B = 4099
    x = rng.uniform(-0.6, 0.6, (B, 13)); x[:, 2] = rng.uniform(-np.pi, np.pi, B); x[:, 12] = -9.80665
    u = rng.uniform(-100, 400, (B, 12))
    feet = rng.uniform(-0.3, 0.3, (B, 12))
    for as_written in (True, False):
        r = eng.wbid_reference(x, u, feet, as_written=as_written)
        assert eng.kernel_name() == "wbid_reference_f64"
        o = co.wbid_reference_batch(x, u, feet, eng.cfg.mass, tuple(eng.cfg.inertia), as_written=as_written)
        assert np.abs(r["R"] - o["R"]).max() < 1e-12
        assert np.array_equal(r["base_vel"], o["base_vel"])
        assert np.abs(r["base_acc"] - o["base_acc"]).max() < 1e-10 * max(1.0, np.abs(o["base_acc"]).max())
        assert np.abs(r["com_acc"] - o["com_acc"]).max() < 1e-12 * max(1.0, np.abs(o["com_acc"]).max())
    assert np.array_equal(r["com_pos"], x[:, 3:6]) and np.array_equal(r["wrench"].reshape(B, 12), u)


def test_cascade_end_to_end_one_control_step(eng):
    """QP -> WBID references on the GPU, the way ros_run_simulation.py:137-141 chains them."""
    import srbd_oracle as orc
    x0, xr, ft, ct = orc.synthetic_batch(8, 10, seed=3, schedule="single")
    out = eng.solve(x0, xr, ft, ct)
    r = eng.wbid_reference(out["x"][:, 1], out["u"][:, 0], ft[:, 0], as_written=False)
    # Newton: m a = sum f + m g, with the engine's own first-step forces
    want = out["u"][:, 0].reshape(8, 4, 3).sum(1) / eng.cfg.mass + np.array([0, 0, -9.80665])
    assert np.abs(r["com_acc"] - want).max() < 1e-9


def test_mpc_inputs_kernel_is_bit_exact_and_feeds_the_solve(eng):
    """SURVEY 8(f) row 2 on the GPU: srbdqp_mpc_inputs_f64 against the oracle bit for bit (every output is one rounded
    operation per step of the host node's arithmetic), for standing and walking robots with and without a commanded
    velocity; then the fleet step stays on the device -- mpc_inputs_device -> solve_device(pcom) -> wbid_reference_device
    -- and matches the same chain through host buffers."""
    import torch
    rng = np.random.default_rng(5)
    B, N = 517, eng.N
    com = np.array([0.05268, 7.44e-5, 0.59798])
    x0 = rng.normal(size=(B, 13)) * 0.05
    x0[:, 3:6] += com
    x0[:, 12] = -9.80665
    half = np.array([[-0.05, 0.1, -0.6], [0.12, 0.1, -0.6], [-0.05, -0.1, -0.6], [0.12, -0.1, -0.6]])
    feet = (x0[:, None, 3:6] + half[None] + rng.normal(size=(B, 4, 3)) * 0.01).reshape(B, 12)
    stamp = rng.uniform(0.0, 5.0, size=B)
    v_ref = np.where(rng.random((B, 1)) < 0.5, 0.0, rng.normal(size=(B, 2)) * 0.3)
    standing = (rng.random(B) < 0.25).astype(np.uint8)
    r = eng.mpc_inputs(x0, feet, stamp, v_ref, com, standing=standing)
    assert eng.kernel_name() == "mpc_inputs_f64"
    o = co.mpc_inputs(x0, feet, stamp, v_ref, com, N, eng.cfg.dt, standing=standing)
    for k in ("x_ref", "foot", "contact", "pcom", "landing"):
        assert np.array_equal(r[k], o[k]), k
    assert set(np.unique(r["contact"].sum(axis=2))) <= {2, 4}
    # the whole fleet step on device buffers
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_x0, d_ft, d_st, d_v, d_sd = t(x0), t(feet), t(stamp), t(v_ref), t(standing)
    d_xr = torch.empty((B, N, 13), dtype=torch.float64, device=dev); d_f = torch.empty((B, N, 12), dtype=torch.float64, device=dev)
    d_ct = torch.empty((B, N, 4), dtype=torch.uint8, device=dev); d_pc = torch.empty((B, N, 3), dtype=torch.float64, device=dev)
    d_lp = torch.empty((B, 3), dtype=torch.float64, device=dev)
    d_u = torch.empty((B, N, 12), dtype=torch.float64, device=dev); d_x = torch.empty((B, N + 1, 13), dtype=torch.float64, device=dev)
    d_s = torch.empty(B, dtype=torch.int32, device=dev)
    eng.mpc_inputs_device(B, d_x0.data_ptr(), d_ft.data_ptr(), d_st.data_ptr(), d_v.data_ptr(), com, d_xr.data_ptr(), d_f.data_ptr(),
                          d_ct.data_ptr(), d_pc.data_ptr(), landing=d_lp.data_ptr(), standing=d_sd.data_ptr())
    eng.solve_device(B, d_x0.data_ptr(), d_xr.data_ptr(), d_f.data_ptr(), d_ct.data_ptr(), d_u.data_ptr(), x_out=d_x.data_ptr(),
                     status=d_s.data_ptr(), pcom=d_pc.data_ptr())
    eng.synchronize()
    host = eng.solve(x0, o["x_ref"], o["foot"], o["contact"], pcom=o["pcom"])
    assert np.array_equal(d_s.cpu().numpy(), host["status"]) and (host["status"] > 0).all()
    assert np.array_equal(d_u.cpu().numpy(), host["u"]) and np.array_equal(d_lp.cpu().numpy(), o["landing"])
    fz = host["u"][:, 0].reshape(B, 4, 3)[:, :, 2]
    assert np.all(fz[o["contact"][:, 0] == 0] == 0.0) and np.all(fz[o["contact"][:, 0] == 1] >= 10.0 - 1e-3)


@pytest.mark.parametrize("N", [4, 8, 12, 16, 20, 24])
def test_mpc_inputs_kernel_every_horizon(N):
    """The input kernel is instantiated per horizon (index divisions by a compile-time N): every instantiation against the
    oracle bit for bit, with tile remainders (B not a multiple of 32), negative stamps and a long gait period."""
    import torch  # noqa: F401
    from g1_locomotion_amd import BatchMPC
    rng = np.random.default_rng(100 + N)
    B = 77
    com = np.array([0.05, 0.0, 0.6])
    x0 = rng.normal(size=(B, 13)) * 0.1
    feet = rng.normal(size=(B, 12))
    stamp = rng.uniform(-2.0, 50.0, size=B)
    v_ref = rng.normal(size=(B, 2)) * 0.2
    v_ref[::3] = 0.0
    standing = (rng.random(B) < 0.2).astype(np.uint8)
    with BatchMPC(horizon=N) as e:
        r = e.mpc_inputs(x0, feet, stamp, v_ref, com, standing=standing, period_steps=9, double_support_steps=2, hip_offset_y=0.08)
        o = co.mpc_inputs(x0, feet, stamp, v_ref, com, N, e.cfg.dt, standing=standing, period_steps=9, double_support_steps=2, hip_offset_y=0.08)
    for k in ("x_ref", "foot", "contact", "pcom", "landing"):
        assert np.array_equal(r[k], o[k]), k
