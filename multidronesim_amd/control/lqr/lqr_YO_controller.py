"""control/lqr/lqr_YO_controller.py of the reference: ``LQRYankOmegaController(env, lin_model,
yo_controller)`` -- the nominal controller of the order-3 CBF demo (simulations/CBFTestOrd3.py:294-297).

State x = [r, p, y, F, vx, vy, vz, x, y, z] (F from calc_z_thrust of the obs), input u = [yank, wx, wy, wz].
The gain is the host-side continuous ARE of the reference (``compute_gain_matrix`` :59-64, Bryson weights
:18-41); ``u = -K e`` (:99-124) and ``compute_low_level`` (:85-97) run in HIP kernels."""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.linalg as la
import torch

from ... import _capi as capi
from ..._device import stream_ptr, to_device
from ..base_controller import BaseController


class LQRYankOmegaController(BaseController):
    def __init__(self, env, lin_model, yo_controller=None, debug=False, use_noisy_model=False, Q=None, R=None):
        super().__init__(env)
        self.yo_controller = yo_controller
        if R is None:
            max_yank = (env.MAX_THRUST / env.CTRL_TIMESTEP) / 200
            R = np.diag([1 / (max_yank ** 2), 1 / (0.1 ** 2), 1 / (0.1 ** 2), 1 / (0.1 ** 2)])
        if Q is None:
            max_thrust = env.MAX_THRUST - env.M * env.G
            qflat = [1 / ((np.pi / 20) ** 2)] * 2 + [1 / ((np.pi / 40) ** 2)] + [1 / (max_thrust ** 2)] + [1 / (.15 ** 2)] * 3 \
                + [1 / (.05 ** 2)] * 3
            Q = np.diag(qflat)
        self.lin_model = lin_model
        self.Q, self.R = Q, R
        self.debug = debug
        self.use_noisy_model = use_noisy_model
        self.A = lin_model.Ahat if use_noisy_model else lin_model.A
        self.B = lin_model.Bhat if use_noisy_model else lin_model.B
        self.desired_pos = self.desired_vel = self.desired_yaw = None
        self.compute_gain_matrix()

    def compute_gain_matrix(self):
        self.P = la.solve_continuous_are(self.A, self.B, self.Q, self.R, e=None, s=None, balanced=True)
        self.K = la.solve(self.R, self.B.T @ self.P)
        K = np.ascontiguousarray(self.K, dtype=np.float64)
        capi.check(self.env._lib.mds_set_lqr_yank_omega_gain(self.env._h, capi.as_double_ptr(K)), "mds_set_lqr_yank_omega_gain")

    def set_desired_trajectory(self, robot_idx, desired_pos, desired_vel, desired_acc, desired_yaw, desired_omega):
        self.desired_pos = desired_pos
        self.desired_vel = desired_vel
        self.desired_yaw = desired_yaw

    def step_cost(self, x, u):
        return x.T @ self.Q @ x + u.T @ self.R @ u

    def compute_batched(self, obs, des):
        """obs [E,D,20], des [E,D,11] -> u [E,D,4] = (yank, wx, wy, wz)."""
        env = self.env
        o = to_device(obs, env.device, env.dtype).reshape(env.n, capi.OBS_DIM)
        d = to_device(des, env.device, env.dtype).reshape(env.n, capi.DES_DIM)
        u = torch.empty((env.NUM_ENVS, env.NUM_DRONES, 4), dtype=env.dtype, device=env.device)
        capi.check(env._lib.mds_lqr_yank_omega_compute(env._h, C.c_void_p(o.data_ptr()), C.c_void_p(d.data_ptr()),
                                                       C.c_void_p(u.data_ptr()), C.c_void_p(stream_ptr(env.device))),
                   "mds_lqr_yank_omega_compute")
        return u

    def compute(self, obs=None, x=None, skip_low_level=False):
        """Reference signature (single drone, slot 0 of the env's batch): -> (action | None, u)."""
        if obs is None:
            raise NotImplementedError("compute(x=...) without an obs is not part of the batched path")
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
        return self.yo_controller.compute_low_level_batched(U, O_).reshape(-1, 4)[0].double().cpu().numpy()

    def cap_u(self, u):
        pass
