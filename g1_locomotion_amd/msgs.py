"""Plain-Python mirrors of the g1_msgs wire format and the MPC node's message mapping.

The live demo talks to the MPC over two topics typed by g1_msgs/SRBD_state
(g1_mujoco_sim/src/ros_run_simulation.py:501,504).  There is no ROS in this environment, so the messages are
dataclasses with the same field names and nesting as g1_msgs/msg/{State,ContactPoint,SRBD_state}.msg; a rospy node
only has to copy field by field.  `MpcNode.step()` is what `mpc_to_wbid_node`
(g1_mujoco_sim/launch/mpc_wbid_simulation.launch:6) does per received /srbd_current message:

    /srbd_current (states_horizon[0] = current state, contacts[i].position)      ros_run_simulation.py:21-79
        -> MPC.update(...)                                                        run_simulation.py:106
    /mpc_solution (states_horizon[i] = predicted states, contacts[i].force = u_opt0, contacts[i].active,
                   landing_position)                                              ros_run_simulation.py:188-218

The gait schedule and the landing position live inside the absent g1_mpc module; `AlternatingGait` is this build's
own minimal stand-in (fixed-period alternating single support, Raibert-style landing point).
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import List

import numpy as np

CONTACT_NAMES = ("left_foot_line_contact_lower", "left_foot_line_contact_upper",
                 "right_foot_line_contact_lower", "right_foot_line_contact_upper")   # ros_run_simulation.py:65


@dataclass
class Vector3:
    x: float = 0.0
    y: float = 0.0
    z: float = 0.0

    def arr(self):
        return np.array([self.x, self.y, self.z], dtype=np.float64)


@dataclass
class State:                       # g1_msgs/msg/State.msg
    trajectory_index: int = 0
    orientation: Vector3 = field(default_factory=Vector3)        # roll, pitch, yaw
    position: Vector3 = field(default_factory=Vector3)           # CoM
    angular_velocity: Vector3 = field(default_factory=Vector3)
    linear_velocity: Vector3 = field(default_factory=Vector3)
    gravity: float = -9.80665                                    # float32 on the wire (State.msg:20)


@dataclass
class ContactPoint:                # g1_msgs/msg/ContactPoint.msg
    name: str = ""
    position: Vector3 = field(default_factory=Vector3)
    force: Vector3 = field(default_factory=Vector3)
    active: bool = False


@dataclass
class SRBDState:                   # g1_msgs/msg/SRBD_state.msg
    stamp: float = 0.0
    frame_id: str = "SRBD"
    states_horizon: List[State] = field(default_factory=list)
    contacts: List[ContactPoint] = field(default_factory=list)
    landing_position: Vector3 = field(default_factory=Vector3)


def state_to_vec(s: State) -> np.ndarray:
    """13-vector in the order the simulator unpacks it (ros_run_simulation.py:199-211)."""
    return np.concatenate([s.orientation.arr(), s.position.arr(), s.angular_velocity.arr(), s.linear_velocity.arr(),
                           [np.float64(np.float32(s.gravity))]])


def vec_to_state(x, index=0) -> State:
    x = np.asarray(x, dtype=np.float64).reshape(13)
    return State(index, Vector3(*x[0:3]), Vector3(*x[3:6]), Vector3(*x[6:9]), Vector3(*x[9:12]), float(x[12]))


def make_srbd_current(x, foot_positions, forces=None, stamp=0.0) -> SRBDState:
    """What publish_current_state() sends (ros_run_simulation.py:21-79): one state, four contact points."""
    foot_positions = np.asarray(foot_positions, dtype=np.float64).reshape(4, 3)
    forces = np.zeros(12) if forces is None else np.asarray(forces, dtype=np.float64).reshape(-1)
    msg = SRBDState(stamp=stamp, states_horizon=[vec_to_state(x)])
    for i, name in enumerate(CONTACT_NAMES):
        msg.contacts.append(ContactPoint(name, Vector3(*foot_positions[i]), Vector3(*forces[3 * i:3 * i + 3]), False))
    return msg


def unpack_mpc_solution(msg: SRBDState):
    """What callback_mpc_solution() extracts (ros_run_simulation.py:188-218): x_opt, u_opt0, contact_states, landing."""
    x_opt = np.stack([state_to_vec(s) for s in msg.states_horizon])
    u_opt0 = np.concatenate([c.force.arr() for c in msg.contacts])
    contact_states = np.array([c.active for c in msg.contacts], dtype=bool)
    return x_opt, u_opt0, contact_states, msg.landing_position.arr()


class AlternatingGait:
    """Fixed-period alternating single support with an optional double-support overlap (own design; the reference's
    schedule is inside the absent module).  swing_steps = 0.25 s / dt by default (ros_run_simulation.py:148)."""

    def __init__(self, dt=0.04, swing_time=0.25, double_support_steps=1, standing=False):
        self.dt, self.period = dt, max(1, int(round(swing_time / dt)))
        self.ds, self.standing = int(double_support_steps), bool(standing)

    def contact_horizon(self, t0: float, N: int) -> np.ndarray:
        """(N, 4) flags for steps starting at time t0; column order = CONTACT_NAMES."""
        out = np.ones((N, 4), dtype=np.uint8)
        if self.standing:
            return out
        k0 = int(np.floor(t0 / self.dt + 1e-9))
        for k in range(N):
            ph = (k0 + k) % (2 * self.period)
            left_stance = ph < self.period
            ds = (ph % self.period) < self.ds
            out[k, 0:2] = 1 if (left_stance or ds) else 0
            out[k, 2:4] = 1 if ((not left_stance) or ds) else 0
        return out

    def landing_position(self, com, v_com, v_ref, hip_offset_y, swing_is_left: bool, swing_time=None) -> np.ndarray:
        """Raibert-style point-contact landing position of the swinging foot, on the ground plane."""
        T = self.period * self.dt if swing_time is None else swing_time
        side = 1.0 if swing_is_left else -1.0
        p = np.array([com[0] + 0.5 * T * v_com[0] + 0.03 * (v_com[0] - v_ref[0]),
                      com[1] + side * hip_offset_y + 0.5 * T * v_com[1] + 0.03 * (v_com[1] - v_ref[1]), 0.0])
        return p


class MpcNode:
    """Message-in / message-out wrapper of an MPC object (`mpc` needs the reference surface: x0, x_ref_hor,
    HORIZON_LENGTH, g, update()).  One call to step() per /srbd_current message."""

    def __init__(self, mpc, gait: AlternatingGait, com_target, v_ref=(0.0, 0.0), hip_offset_y=0.0645):
        self.mpc, self.gait = mpc, gait
        # what the reference node publishes on /mpc_statistics (index 4 is plotted as the solve time,
        # g1_mujoco_sim/config/MPC_QP_layout.xml:258-259): outcome and wall time of the last step()
        self.statistics = {"solve_time": 0.0, "status": 0, "iters": 0}
        self.com_target = np.asarray(com_target, dtype=np.float64)
        self.v_ref = np.asarray(v_ref, dtype=np.float64)
        self.hip_offset_y = hip_offset_y

    def step(self, msg: SRBDState) -> SRBDState:
        mpc, N = self.mpc, self.mpc.HORIZON_LENGTH
        x0 = state_to_vec(msg.states_horizon[0])
        mpc.x0[:] = x0.reshape(13, 1)
        mpc.x_ref_hor[:] = 0.0
        k = np.arange(1, N + 1)
        mpc.x_ref_hor[:, 2] = x0[2]
        mpc.x_ref_hor[:, 3] = self.com_target[0] + 0.0 * k if not np.any(self.v_ref) else x0[3] + self.v_ref[0] * k * mpc.dt
        mpc.x_ref_hor[:, 4] = self.com_target[1] if not np.any(self.v_ref) else x0[4] + self.v_ref[1] * k * mpc.dt
        mpc.x_ref_hor[:, 5] = self.com_target[2]
        mpc.x_ref_hor[:, 9] = self.v_ref[0]
        mpc.x_ref_hor[:, 10] = self.v_ref[1]
        mpc.x_ref_hor[:, 12] = x0[12]
        feet = np.concatenate([c.position.arr() for c in msg.contacts])
        c_horizon = [feet.copy() for _ in range(N)]                       # run_simulation.py:94-97
        contact_horizon = self.gait.contact_horizon(msg.stamp, N)
        # lever arms r_i = c_i - p_com: linearise about the MEASURED CoM (plus the commanded drift), not about the target --
        # with the torso-only inertia of wbid.py:261-266 a 1 cm CoM offset is a 40 rad/s^2 modelling error otherwise
        p_com_horizon = x0[3:6][None, :] + np.concatenate([self.v_ref, [0.0]])[None, :] * (k[:, None] - 1) * mpc.dt
        t0 = time.perf_counter()
        u_opt0, x_opt1 = mpc.update(list(contact_horizon), c_horizon, p_com_horizon, x_current=mpc.x0, one_rollout=True)
        self.statistics = {"solve_time": float(getattr(mpc, "solve_time", 0.0)) or (time.perf_counter() - t0),
                           "status": int(getattr(mpc, "status", 0)), "iters": int(getattr(mpc, "iters", 0))}
        out = SRBDState(stamp=msg.stamp)
        out.states_horizon = [vec_to_state(x_opt1[i], i) for i in range(x_opt1.shape[0])]
        u = np.asarray(u_opt0, dtype=np.float64).reshape(-1)
        for i, name in enumerate(CONTACT_NAMES):
            out.contacts.append(ContactPoint(name, msg.contacts[i].position, Vector3(*u[3 * i:3 * i + 3]), bool(contact_horizon[0, i])))
        swing_left = not bool(contact_horizon[0, 0])
        lp = self.gait.landing_position(x0[3:6], x0[9:12], self.v_ref, self.hip_offset_y, swing_left)
        out.landing_position = Vector3(*lp)
        return out
