"""control/low_level/yank_omega_ctrl.py of the reference: ``YankOmegaController(env)`` -- a wrapper
around the ThrustOmegaController that integrates the yank input over one control step,
``thrust = cur_thrust + yank * control_timestep`` (:49-53), before the body-rate PID.

The reference only ever calls it through ``LQRYankOmegaController.compute_low_level``
(control/lqr/lqr_YO_controller.py:85-97) with ``cur_thrust = calc_z_thrust(env, obs)``; that
composition is one HIP kernel here (mds_yank_omega_compute).  The PID memory is the handle's
ThrustOmega memory, as in the reference where the wrapper owns a ThrustOmegaController."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ... import _capi as capi
from ..._device import stream_ptr, to_device
from .thrust_omega_ctrl import ThrustOmegaController as TOC


class YankOmegaController:
    def __init__(self, env):
        self.thrust_omega_ctrl = TOC(env)
        self.env = env
        self.hover_thrust = env.G * env.M
        self.cur_thrust = self.hover_thrust

    def reset(self):
        self.thrust_omega_ctrl.reset()
        self.cur_thrust = self.hover_thrust

    def yank2thrust(self, yank, control_timestep, cur_thrust):
        return cur_thrust + yank * control_timestep

    def compute_low_level_batched(self, u, obs):
        """u [E,D,4] = (yank N/s, target body rates), obs [E,D,20] (world rates, last clipped RPM) -> rpm [E,D,4]."""
        env = self.env
        ut = to_device(u, env.device, env.dtype).reshape(env.n, 4)
        ot = to_device(obs, env.device, env.dtype).reshape(env.n, capi.OBS_DIM)
        rpm = torch.empty((env.NUM_ENVS, env.NUM_DRONES, 4), dtype=env.dtype, device=env.device)
        capi.check(env._lib.mds_yank_omega_compute(env._h, C.c_void_p(ut.data_ptr()), C.c_void_p(ot.data_ptr()),
                                                   C.c_void_p(rpm.data_ptr()), C.c_void_p(stream_ptr(env.device))),
                   "mds_yank_omega_compute")
        self.thrust_omega_ctrl.control_counter += 1
        return rpm

    def computeControlFromInput(self, u, control_timestep, cur_ang_vel, cur_thrust):
        """Reference signature (single drone, slot 0): u (4,) = (yank, w), body rates (3,), current thrust -> rpm (4,)."""
        u_thrust = np.array(u, dtype=np.float64)
        u_thrust[0] = self.yank2thrust(u_thrust[0], control_timestep, cur_thrust)
        return self.thrust_omega_ctrl.computeControlFromInput(u_thrust, control_timestep, cur_ang_vel)
