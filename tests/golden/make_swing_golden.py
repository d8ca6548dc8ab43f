"""Generates tests/golden/swing_golden.npz FROM THE REFERENCE ITSELF: imports
/root/reference/g1_mujoco_sim/src/swing_trajectory.py (the only hot-path-adjacent module of the reference that imports in
the build container) and records inputs and outputs of its SwingTrajectory class.  Run in the build
container only (the reference does not exist on the GPU box); the .npz it writes is the committed fixture.

    MPLBACKEND=Agg python tests/golden/make_swing_golden.py
"""
import os
import sys

import numpy as np

REF = "/root/reference/g1_mujoco_sim/src"


def main():
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path.insert(0, REF)
    import swing_trajectory  # the reference module
    rng = np.random.default_rng(20251010)
    cases = []
    # the reference's own use (ros_run_simulation.py:296-312): start on the ground, apex 0.05, land on the ground
    cases.append((np.array([0.0, 0.1, 0.0]), np.array([0.2, 0.1, 0.0]), 0.05))
    cases.append((np.array([-0.05, -0.0645, 0.0]), np.array([0.12, -0.07, 0.0]), 0.05))
    for _ in range(30):
        ps = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.2, 0.2), rng.uniform(-0.02, 0.03)])
        pf = np.array([rng.uniform(-0.3, 0.5), rng.uniform(-0.2, 0.2), rng.uniform(-0.02, 0.03)])
        cases.append((ps, pf, rng.uniform(0.02, 0.12)))
    ts = np.concatenate([np.linspace(0.0, 1.0, 21), [0.5, 0.49999, 0.50001, 0.123456, 0.987654]])
    P0, P1, ZM, T, POS, VZ, AZ, CO = [], [], [], [], [], [], [], []
    for ps, pf, zm in cases:
        s = swing_trajectory.SwingTrajectory()
        s.reset()
        s.set_positions_xy(ps[0], pf[0], ps[1], pf[1])
        s.set_positions_z(ps[2], zm, pf[2])
        s.calculate_coeff()
        for t in ts:
            x, y = s.calculate_position_xy(t)
            P0.append(ps); P1.append(pf); ZM.append(zm); T.append(t)
            POS.append([x, y, s.calculate_position_z(t)])
            VZ.append(s.calculate_velocity_z(t)); AZ.append(s.calculate_acceleration_z(t)); CO.append(s.coeff.copy())
    # the 100-sample curves of calculate_all_trajectories_z / calculate_trajectory_xy for the first case
    s = swing_trajectory.SwingTrajectory()
    s.set_positions_xy(0.0, 0.2, 0.1, 0.1); s.set_positions_z(0.0, 0.05, 0.0); s.calculate_coeff()
    pz, vz, az = s.calculate_all_trajectories_z()
    xy = np.array(s.calculate_trajectory_xy())
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "swing_golden.npz")
    np.savez_compressed(out, p_start=np.array(P0), p_final=np.array(P1), z_middle=np.array(ZM), progress=np.array(T),
                        pos=np.array(POS, dtype=np.float64), vel_z=np.array(VZ), acc_z=np.array(AZ), coeff=np.array(CO),
                        curve_z=np.array(pz), curve_vz=np.array(vz), curve_az=np.array(az), curve_xy=xy)
    print("wrote", out, len(T), "samples")


if __name__ == "__main__":
    main()
