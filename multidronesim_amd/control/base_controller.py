"""Controller protocol shared by every controller of the package.

Same two entry points the reference's controllers expose (its ``control/base_controller.py``), because the driver
scripts call exactly these: ``set_desired_trajectory(...)`` once per control step and drone, then ``compute(obs)``.
Batched variants (``compute_batched``) live on the concrete classes; this base only pins the names and the env handle."""
from __future__ import annotations


class BaseController:
    """Holds the env (a multidronesim_amd CtrlAviary: constants + the device handle the kernels run on)."""

    def __init__(self, env):
        self.env = env

    def set_desired_trajectory(self, robot_idx, desired_pos, desired_vel, desired_acc, desired_yaw, desired_omega):
        """Remember the set-point of drone ``robot_idx`` for the next ``compute``: position, velocity, acceleration
        (3-vectors, world frame), yaw [rad] and yaw rate [rad/s].  Concrete controllers keep what they use."""
        return None

    def compute(self, obs, skip_low_level=False):
        """Map one 20-float observation to the controller's output (RPM, or the pair (action, u) for the LQR family).
        ``skip_low_level`` asks the LQR family for the mid-level input only."""
        return None
