"""MI355X-native batched SRBD convex-MPC QP engine: the hot path of ioloizou/g1_locomotion's g1_mpc
(`MPC.update`, g1_mujoco_sim/src/run_simulation.py:106) as hand-written HIP behind a C-ABI.

    from g1_locomotion_amd import mpc
    MPC = mpc.MPC(dt=0.04); MPC.init_matrices()
"""
from . import _lib  # noqa: F401
from . import mpc  # noqa: F401
from . import swing_trajectory  # noqa: F401
from .mpc import MPC, BatchMPC, RaggedMPC  # noqa: F401
from ._lib import SrbdqpError  # noqa: F401

__all__ = ["mpc", "swing_trajectory", "MPC", "BatchMPC", "RaggedMPC", "SrbdqpError"]
