"""control/geometric.py of the reference: ``GeometricControl(env)``,
``set_desired_trajectory(...)``, ``compute(obs, return_omegas=False)``.

The arithmetic of ``compute`` (geometric.py:59-115, quirks included) runs in the HIP kernel
behind ``mds_geometric_compute``; batched use goes through ``compute_batched`` or, fused with
the physics step, ``env.step_geometric``."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _capi as capi
from .._device import stream_ptr, to_device
from .base_controller import BaseController


class GeometricControl(BaseController):
    def __init__(self, env):
        super().__init__(env)
        self.m = env.M
        self.J = env.J
        self.Kp = np.array([2.25, 2.25, 2.25])
        self.Kv = np.array([3.5, 3.5, 3.5])
        self.KR = np.array([125, 125, 125])
        self.Kw = np.array([10, 10, 10])
        self.e3 = np.array([0., 0., 1.])
        self.g = 9.81
        self.max_force = 4 * self.m * self.g
        self.max_torque = 0.1
        self.max_tilt_angle = 40 * np.pi / 180
        self.desired_position = None
        self.desired_rotation = None
        self.desired_velocity = None
        self.desired_acceleration = None
        self.desired_yaw = None
        self.desired_omega = None

    def set_desired_trajectory(self, robot_idx, desired_pos, desired_vel, desired_acc, desired_yaw, desired_omega):
        self.desired_position = desired_pos
        self.desired_velocity = desired_vel
        self.desired_acceleration = desired_acc
        self.desired_yaw = desired_yaw
        self.desired_omega = desired_omega

    def _push_gains(self):
        self.env.set_geometric_gains(Kp=self.Kp, Kv=self.Kv, KR=self.KR, Kw=self.Kw, g=self.g,
                                     max_tilt_angle=self.max_tilt_angle)

    def compute_batched(self, obs, des, return_omegas=False):
        """obs [E,D,20], des [E,D,11] (pos3|vel3|acc3|yaw|yaw_rate) on the env's device ->
        rpm [E,D,4]; with ``return_omegas`` also (force [E,D], w_des [E,D,3], R_des [E,D,3,3])."""
        env = self.env
        self._push_gains()
        obs_t = to_device(obs, env.device, env.dtype).reshape(env.n, capi.OBS_DIM)
        des_t = to_device(des, env.device, env.dtype).reshape(env.n, capi.DES_DIM)
        rpm = torch.empty((env.NUM_ENVS, env.NUM_DRONES, 4), dtype=env.dtype, device=env.device)
        aux = torch.empty((env.NUM_ENVS, env.NUM_DRONES, capi.GEO_AUX_DIM), dtype=env.dtype, device=env.device) if return_omegas else None
        capi.check(env._lib.mds_geometric_compute(env._h, C.c_void_p(obs_t.data_ptr()), C.c_void_p(des_t.data_ptr()),
                                                  C.c_void_p(rpm.data_ptr()),
                                                  C.c_void_p(aux.data_ptr() if aux is not None else None),
                                                  C.c_void_p(stream_ptr(env.device))), "mds_geometric_compute")
        if return_omegas:
            return rpm, aux[..., 0], aux[..., 1:4], aux[..., 4:13].reshape(env.NUM_ENVS, env.NUM_DRONES, 3, 3)
        return rpm

    def compute(self, obs, return_omegas=False):
        """Single-drone call with the reference's signature: obs (20,) -> 4 RPM, or
        ``(force, w_des, R_des)`` when ``return_omegas``.  The same desired trajectory is applied
        to slot 0 of the env's batch; only that row is returned."""
        env = self.env
        full_obs = np.zeros((env.n, capi.OBS_DIM))
        full_obs[:, 6] = 1.0
        full_obs[0] = np.asarray(obs, dtype=np.float64)
        des = np.zeros((env.n, capi.DES_DIM))
        des[0] = np.hstack([self.desired_position, self.desired_velocity, self.desired_acceleration,
                            self.desired_yaw, self.desired_omega])
        out = self.compute_batched(full_obs, des, return_omegas=return_omegas)
        if return_omegas:
            rpm, force, w_des, R_des = out
            return (float(force.reshape(-1)[0].cpu()), w_des.reshape(-1, 3)[0].double().cpu().numpy(),
                    R_des.reshape(-1, 3, 3)[0].double().cpu().numpy())
        return out.reshape(-1, 4)[0].double().cpu().numpy()
