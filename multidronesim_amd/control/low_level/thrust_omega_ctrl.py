"""control/low_level/thrust_omega_ctrl.py of the reference: ``ThrustOmegaController(env)`` -- the
body-rate PID between the CBF output [F, w] and the motor RPMs (simulations/CBFTest.py:348).

``computeControlFromInput(u, control_timestep, cur_ang_vel)`` keeps the reference signature; the
arithmetic (:81-132) and the PID memory live on the GPU (mds_thrust_omega_*).  ``control_timestep``
must equal the env's CTRL_TIMESTEP (the only value the reference passes)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ... import _capi as capi
from ..._device import stream_ptr, to_device


class ThrustOmegaController:
    def __init__(self, env):
        self.env = env
        self.DRONE_MODEL = env.DRONE_MODEL
        self.KF = env.KF
        self.P_COEFF_OMEGA_TOR = np.array([17500., 17500., 17500.])
        self.I_COEFF_OMEGA_TOR = np.array([10., 10., 10.])
        self.D_COEFF_OMEGA_TOR = np.array([0., 0., 0.])
        self.PWM2RPM_SCALE, self.PWM2RPM_CONST = 0.2685, 4070.3
        self.MIN_PWM, self.MAX_PWM = 20000, 65535
        self.reset()

    def reset(self):
        self.control_counter = 0
        env = self.env
        capi.check(env._lib.mds_lowlevel_reset(env._h, C.c_void_p(stream_ptr(env.device))), "mds_lowlevel_reset")

    def compute_batched(self, u, cur_ang_vel_body):
        """u [E,D,4] = (thrust, target body rates), cur_ang_vel_body [E,D,3] -> rpm [E,D,4]."""
        env = self.env
        ut = to_device(u, env.device, env.dtype).reshape(env.n, 4)
        wt = to_device(cur_ang_vel_body, env.device, env.dtype).reshape(env.n, 3)
        rpm = torch.empty((env.NUM_ENVS, env.NUM_DRONES, 4), dtype=env.dtype, device=env.device)
        capi.check(env._lib.mds_thrust_omega_from_rates(env._h, C.c_void_p(ut.data_ptr()), C.c_void_p(wt.data_ptr()),
                                                        C.c_void_p(rpm.data_ptr()), C.c_void_p(stream_ptr(env.device))),
                   "mds_thrust_omega_from_rates")
        self.control_counter += 1
        return rpm

    def compute_low_level_batched(self, u, obs):
        """LQROmegaController.compute_low_level for every drone: obs [E,D,20] world rates -> body."""
        env = self.env
        ut = to_device(u, env.device, env.dtype).reshape(env.n, 4)
        ot = to_device(obs, env.device, env.dtype).reshape(env.n, 20)
        rpm = torch.empty((env.NUM_ENVS, env.NUM_DRONES, 4), dtype=env.dtype, device=env.device)
        capi.check(env._lib.mds_thrust_omega_compute(env._h, C.c_void_p(ut.data_ptr()), C.c_void_p(ot.data_ptr()),
                                                     C.c_void_p(rpm.data_ptr()), C.c_void_p(stream_ptr(env.device))),
                   "mds_thrust_omega_compute")
        self.control_counter += 1
        return rpm

    def computeControlFromInput(self, u, control_timestep, cur_ang_vel):
        """Reference signature (single drone): u (4,), body rates (3,) -> rpm (4,).  Applied to slot 0."""
        env = self.env
        if abs(control_timestep - env.CTRL_TIMESTEP) > 1e-12:
            raise ValueError("control_timestep must be env.CTRL_TIMESTEP")
        U = np.zeros((env.n, 4))
        W = np.zeros((env.n, 3))
        U[0], W[0] = np.asarray(u, dtype=np.float64), np.asarray(cur_ang_vel, dtype=np.float64)
        return self.compute_batched(U, W).reshape(-1, 4)[0].double().cpu().numpy()
