"""control/lqr/lqr_controller.py of the reference: ``LQRController(env, lin_model, ..., use_noisy_model)`` -- the controller
simulations/EnvGeometric.py runs by default ('lqr', :32, :425-427) on the 12-state LinearizedModel.

The gain is the reference's host-side continuous ARE (``compute_gain_matrix`` :53-57, Bryson weights :18-37); the per-step
``u = -K e + [M G,0,0,0]`` -> ``input_to_action`` (:83-113) runs in HIP kernels (mds_lqr_compute; fused with the trajectory
sample and env.step in mds_step_lqr / ``env.step_lqr(t)``)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.linalg as la
import torch

from ... import _capi as capi
from ..._device import stream_ptr, to_device
from ..base_controller import BaseController


class LQRController(BaseController):
    def __init__(self, env, lin_model, Q=None, R=None, debug=False, use_noisy_model=False):
        super().__init__(env)
        max_thrust = env.MAX_THRUST
        rflat = [1 / (max_thrust ** 2), 1 / (0.001 ** 2), 1 / (0.001 ** 2), 1 / (0.001 ** 2)]
        qflat = [1 / ((np.pi / 40) ** 2)] * 3 + [1 / (.25 ** 2)] * 3 + [1 / (.15 ** 2)] * 3 + [1 / (.05 ** 2)] * 3
        self.R = np.diag(rflat)                    # the reference overwrites its Q / R arguments with these (:23-37)
        self.Q = np.diag(qflat)
        self.lin_model = lin_model
        self.debug = debug
        self.A = lin_model.Ahat if use_noisy_model else lin_model.A
        self.B = lin_model.Bhat if use_noisy_model else lin_model.B
        self.desired_pos = self.desired_vel = self.desired_omega = self.desired_yaw = None
        self.compute_gain_matrix()

    def compute_gain_matrix(self):
        self.P = la.solve_continuous_are(self.A, self.B, self.Q, self.R, e=None, s=None, balanced=True)
        self.K = la.solve(self.R, self.B.T @ self.P)
        K = np.ascontiguousarray(self.K, dtype=np.float64)
        capi.check(self.env._lib.mds_set_lqr_gain(self.env._h, capi.as_double_ptr(K)), "mds_set_lqr_gain")

    def set_desired_trajectory(self, robot_idx, desired_pos, desired_vel, desired_acc, desired_yaw, desired_omega):
        self.desired_pos = desired_pos
        self.desired_vel = desired_vel
        self.desired_omega = desired_omega
        self.desired_yaw = desired_yaw

    def step_cost(self, x, u):
        return x.T @ self.Q @ x + u.T @ self.R @ u

    def compute_batched(self, obs, des):
        """obs [E,D,20], des [E,D,11] (pos, vel, -, yaw, omega) -> (action [E,D,4] RPM, u [E,D,4])."""
        env = self.env
        o = to_device(obs, env.device, env.dtype).reshape(env.n, capi.OBS_DIM)
        d = to_device(des, env.device, env.dtype).reshape(env.n, capi.DES_DIM)
        u = torch.empty((env.NUM_ENVS, env.NUM_DRONES, 4), dtype=env.dtype, device=env.device)
        act = torch.empty_like(u)
        capi.check(env._lib.mds_lqr_compute(env._h, C.c_void_p(o.data_ptr()), C.c_void_p(d.data_ptr()), C.c_void_p(u.data_ptr()),
                                            C.c_void_p(act.data_ptr()), C.c_void_p(stream_ptr(env.device))), "mds_lqr_compute")
        return act, u

    def compute(self, obs, skip_low_level=False):
        """Reference signature (single drone, slot 0 of the env's batch): -> (action, u)."""
        env = self.env
        O_ = np.zeros((env.n, capi.OBS_DIM))
        O_[:, 6] = 1.0
        O_[0] = np.asarray(obs, dtype=np.float64)
        Dd = np.zeros((env.n, capi.DES_DIM))
        Dd[0, 0:3], Dd[0, 3:6], Dd[0, 9], Dd[0, 10] = self.desired_pos, self.desired_vel, self.desired_yaw, self.desired_omega
        act, u = self.compute_batched(O_, Dd)
        return act.reshape(-1, 4)[0].double().cpu().numpy(), u.reshape(-1, 4)[0].double().cpu().numpy()
