"""model/linearized.py of the reference: ``LinearizedModel(env)`` -- hover linearisation with the 12-state
x = [r, p, y, r_dot, p_dot, y_dot, vx, vy, vz, px, py, pz] and input u = [F, tau_x, tau_y, tau_z] (:50-75), plus the
deliberately wrong (Ahat, Bhat) the scripts use as a 'noisy model' (inertia and mass off by 0.75).  The constant
matrices parameterise the LQR gain of control/lqr/lqr_controller.py; ``calc_xdot_from_obs`` / ``calc_xdot`` (:83-104, the
linear side of simulations/CompareModels.py) run batched on the GPU (mds_compare_models / mds_linear_xdot)."""
import ctypes as C

import numpy as np
import torch

from .. import _capi as capi
from .._device import stream_ptr, to_device
from ._eval import eval_handle


class LinearizedModel:
    def __init__(self, env, debug=False):
        self.mass = env.M
        self.Ixx, self.Iyy, self.Izz = env.J[0, 0], env.J[1, 1], env.J[2, 2]
        self.g = env.G
        self.env = env
        self.A = np.zeros((12, 12))
        self.B = np.zeros((12, 4))
        self.C = np.eye(12)
        self.D = np.zeros((12, 6))
        self.init_matrices()

    def init_matrices(self):
        self.A[0:3, 3:6] = np.eye(3)
        self.A[9:, 6:9] = np.eye(3)
        self.A[6, 1] = self.g
        self.A[7, 0] = -self.g
        self.B[8, 0] = 1.0 / self.mass
        self.B[3:6, 1:] = np.diag([1 / self.Ixx, 1 / self.Iyy, 1 / self.Izz])
        self.D[:, 2:] = self.B.copy()
        self.D[7, 1] = 1.0 / self.mass
        self.D[6, 0] = 1.0 / self.mass
        self.Ahat = self.A.copy()
        self.Bhat = self.B.copy()
        self.Bhat[3:6, 1:] = np.diag([1 / self.Ixx, 1 / self.Iyy, 1 / self.Izz]) * 0.75
        self.Bhat[8, 0] = 1.0 / (self.mass * .75)

    # ---- x_dot of the linear model (:83-104), batched over leading axes; NumPy in -> NumPy (float64) out, device tensor in -> tensor out
    def _mats(self):
        A = np.ascontiguousarray(self.A, dtype=np.float64)
        B = np.ascontiguousarray(self.B, dtype=np.float64)
        if A.shape != (12, 12) or B.shape != (12, 4):
            raise ValueError(f"matmul: A {A.shape} / B {B.shape} do not fit the 12-state x and the 4 inputs")
        return A, B

    def calc_xdot_from_obs(self, obs):
        """:param obs: observation(s) [..., 20] from the environment (includes the clipped action)
        :return: x_dot of the linear model at the observation's state and action (:83-90)"""
        numpy_in = not isinstance(obs, torch.Tensor)
        shape = tuple(np.shape(obs))
        if shape[-1] != capi.OBS_DIM:
            raise ValueError(f"obs must end in {capi.OBS_DIM} components, got shape {shape}")
        ev = eval_handle(self.env, torch.float64 if numpy_in else obs.dtype)
        ot = to_device(obs, ev.dev, ev.dtype).reshape(-1, capi.OBS_DIM)
        out = torch.empty((ot.shape[0], 12), dtype=ev.dtype, device=ev.dev)
        A, B = self._mats()
        J = (C.c_double * 3)(1.0, 1.0, 1.0)
        capi.check(ev.lib.mds_compare_models(ev.h, C.c_int(ot.shape[0]), C.c_void_p(ot.data_ptr()), capi.as_double_ptr(A), capi.as_double_ptr(B),
                                             C.c_double(self.mass * self.g), C.c_double(1.0), J, C.c_double(0.0), C.c_void_p(out.data_ptr()),
                                             None, None, C.c_void_p(stream_ptr(ev.dev))), "mds_compare_models")
        out = out.reshape(shape[:-1] + (12,))
        return out.cpu().numpy() if numpy_in else out

    def calc_xdot(self, x, action):
        """A (x - x_eq) + B (u - u_eq), u = action_to_input(env, action), x_eq = (0 .. 0, the position of x), u_eq = (m g, 0, 0, 0) (:92-104)."""
        numpy_in = not isinstance(x, torch.Tensor)
        shape = tuple(np.shape(x))
        if shape[-1] != 12:
            raise ValueError(f"x must end in 12 components, got shape {shape}")
        ev = eval_handle(self.env, torch.float64 if numpy_in else x.dtype)
        xt = to_device(x, ev.dev, ev.dtype).reshape(-1, 12)
        at = to_device(action, ev.dev, ev.dtype).reshape(-1, 4)
        if at.shape[0] != xt.shape[0]:
            raise ValueError(f"{xt.shape[0]} states but {at.shape[0]} actions")
        out = torch.empty_like(xt)
        A, B = self._mats()
        capi.check(ev.lib.mds_linear_xdot(ev.h, C.c_int(xt.shape[0]), C.c_void_p(xt.data_ptr()), C.c_void_p(at.data_ptr()), capi.as_double_ptr(A),
                                          capi.as_double_ptr(B), C.c_double(self.mass * self.g), C.c_void_p(out.data_ptr()),
                                          C.c_void_p(stream_ptr(ev.dev))), "mds_linear_xdot")
        out = out.reshape(shape)
        return out.cpu().numpy() if numpy_in else out
