"""model/dynamics.py of the reference: ``QuadrotorDynamics.dynamics(t, state, u)`` (:83-106),
batched on the GPU through mds_quadrotor_dynamics.  ``step()`` raises ValueError in the
reference (12-long derivative vs 24-long state, :105,:110); it does so here too."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _capi as capi
from .._device import require_gpu, stream_ptr, to_device


class QuadrotorDynamics:
    def __init__(self, sim_freq, init_position=None, init_rpys=None):
        self.m, self.Jxx, self.Jyy, self.Jzz, self.g = 6.77, 1.05, 1.05, 2.05, 9.81   # Hummingbird (:24-28)
        self.sim_freq = int(sim_freq)
        self.dt = 1.0 / sim_freq
        self.kf, self.km = 3.16e-10, 7.94e-12
        self.J = np.diag([self.Jxx, self.Jyy, self.Jzz])
        self.J_inv = np.linalg.inv(self.J)

    def load_env_params(self, env):
        self.m, self.g, self.kf = env.M, env.G, env.KF
        self.Ixx, self.Iyy, self.Izz = env.J[0, 0], env.J[1, 1], env.J[2, 2]
        self.sim_freq = env.PYB_FREQ
        self.dt = 1.0 / self.sim_freq
        self.J = np.diag([self.Jxx, self.Jyy, self.Jzz])   # stale Hummingbird inertia, as the reference (:18)
        self.J_inv = np.linalg.inv(self.J)

    def dynamics(self, t, state, u, device=0):
        """state [..., 18] (p, R row-major, v, w), u [..., 4] (thrust, torques) -> [..., 12]."""
        lib = capi.load_library()
        dev = require_gpu(device)
        numpy_in = not isinstance(state, torch.Tensor)
        dt = torch.float64 if numpy_in else state.dtype
        code = {torch.float64: capi.MDS_F64, torch.float32: capi.MDS_F32, torch.float16: capi.MDS_F16}[dt]
        s = to_device(state, dev, dt)
        lead = s.shape[:-1]
        s = s.reshape(-1, 18)
        uu = to_device(u, dev, dt).reshape(-1, 4)
        out = torch.empty((s.shape[0], 12), dtype=dt, device=dev)
        J = (C.c_double * 3)(self.J[0, 0], self.J[1, 1], self.J[2, 2])
        capi.check(lib.mds_quadrotor_dynamics(code, s.shape[0], C.c_void_p(s.data_ptr()), C.c_void_p(uu.data_ptr()),
                                              C.c_double(self.m), J, C.c_double(self.g), C.c_void_p(out.data_ptr()),
                                              C.c_void_p(stream_ptr(dev))), "mds_quadrotor_dynamics")
        out = out.reshape(lead + (12,))
        return out.cpu().numpy() if numpy_in else out

    def step(self, action):
        raise ValueError("operands could not be broadcast together with shapes (12,) (24,)  "
                         "[QuadrotorDynamics.step is broken in the reference, model/dynamics.py:105-110; "
                         "use CtrlAviary(integrator='rk4').step for an integrated step]")
