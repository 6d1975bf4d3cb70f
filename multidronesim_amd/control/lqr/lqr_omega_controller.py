"""control/lqr/lqr_omega_controller.py of the reference: ``LQROmegaController(env, lin_model,
to_controller)`` -- the nominal controller of simulations/CBFTest.py (:290-293).

The gain comes from the continuous ARE on the host exactly like the reference
(``compute_gain_matrix``, :53-57: a one-off 9x9 solve at construction); the per-step
``u = -K e`` (:90-119) and ``compute_low_level`` (:77-88) run in HIP kernels."""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.linalg as la
import torch

from ... import _capi as capi
from ..._device import stream_ptr, to_device
from ..base_controller import BaseController


class LQROmegaController(BaseController):
    def __init__(self, env, lin_model, to_controller=None, debug=False, use_noisy_model=False):
        super().__init__(env)
        self.to_controller = to_controller
        max_thrust = env.MAX_THRUST
        rflat = [1 / (max_thrust ** 2), 1 / (0.1 ** 2), 1 / (0.1 ** 2), 1 / (0.1 ** 2)]
        self.R = np.diag(rflat)
        qflat = [1 / ((np.pi / 20) ** 2)] * 2 + [1 / ((np.pi / 40) ** 2)] + [1 / (.15 ** 2)] * 3 + [1 / (.05 ** 2)] * 3
        self.Q = np.diag(qflat)
        self.lin_model = lin_model
        self.use_noisy_model = use_noisy_model
        self.A = lin_model.Ahat if use_noisy_model else lin_model.A
        self.B = lin_model.Bhat if use_noisy_model else lin_model.B
        self.desired_pos = self.desired_vel = self.desired_yaw = None
        self.compute_gain_matrix()

    def compute_gain_matrix(self):
        self.P = la.solve_continuous_are(self.A, self.B, self.Q, self.R, e=None, s=None, balanced=True)
        self.K = la.solve(self.R, self.B.T @ self.P)
        K = np.ascontiguousarray(self.K, dtype=np.float64)
        capi.check(self.env._lib.mds_set_lqr_omega_gain(self.env._h, capi.as_double_ptr(K)), "mds_set_lqr_omega_gain")

    def set_desired_trajectory(self, robot_idx, desired_pos, desired_vel, desired_acc, desired_yaw, desired_omega):
        self.desired_pos = desired_pos
        self.desired_vel = desired_vel
        self.desired_yaw = desired_yaw

    def compute_batched(self, obs, des):
        """obs [E,D,20], des [E,D,11] -> u [E,D,4] = (F, wx, wy, wz) after cap_u."""
        env = self.env
        o = to_device(obs, env.device, env.dtype).reshape(env.n, capi.OBS_DIM)
        d = to_device(des, env.device, env.dtype).reshape(env.n, capi.DES_DIM)
        u = torch.empty((env.NUM_ENVS, env.NUM_DRONES, 4), dtype=env.dtype, device=env.device)
        capi.check(env._lib.mds_lqr_omega_compute(env._h, C.c_void_p(o.data_ptr()), C.c_void_p(d.data_ptr()), C.c_void_p(u.data_ptr()),
                                                  C.c_void_p(stream_ptr(env.device))), "mds_lqr_omega_compute")
        return u

    def compute(self, obs, skip_low_level=False):
        """Reference signature (single drone, slot 0 of the env's batch): -> (action | None, u)."""
        env = self.env
        O_ = np.zeros((env.n, capi.OBS_DIM))
        O_[:, 6] = 1.0
        O_[0] = np.asarray(obs, dtype=np.float64)
        Dd = np.zeros((env.n, capi.DES_DIM))
        Dd[0, 0:3], Dd[0, 3:6], Dd[0, 9] = self.desired_pos, self.desired_vel, self.desired_yaw
        u = self.compute_batched(O_, Dd).reshape(-1, 4)[0].double().cpu().numpy()
        if skip_low_level:
            return None, u
        return self.compute_low_level(u, obs), u

    def compute_low_level(self, u, obs, idx=0):
        env = self.env
        U = np.zeros((env.n, 4))
        O_ = np.zeros((env.n, capi.OBS_DIM))
        O_[:, 6] = 1.0
        U[0], O_[0] = np.asarray(u, dtype=np.float64), np.asarray(obs, dtype=np.float64)
        return self.to_controller.compute_low_level_batched(U, O_).reshape(-1, 4)[0].double().cpu().numpy()
