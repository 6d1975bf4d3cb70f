"""cbf/qptracker.py of the reference: ``DroneQPTracker.compute_control`` -- minimally change
the nominal input so that the ECBF rows hold (min 1/2|u - u_hat|^2 s.t. G u <= h, :86-114).

The QP runs on the GPU, one wavefront per env (``mds_cbf_filter``) and returns the exact, unique
minimiser (status 0).  An env whose rows are infeasible keeps its nominal control with status 1:
a MODELLED fallback.  The reference takes its ``return u_nominal`` exit (:30-34) only when
``cvxopt.solvers.qp`` raises (:103-112), and cvxopt's ``coneqp`` answers an infeasible QP with
``status 'unknown'`` and its last iterate rather than an exception, so on such envs the reference most
likely applies that iterate; cvxopt is not available here, parity on status-1 envs is unpinned
(include/mds.h, ``mds_cbf_filter``)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _capi as capi
from .._device import stream_ptr, to_device


class DroneQPTracker(object):
    def __init__(self, cbf, order=2, num_robots=1, xdim=9, env=None):
        self.cbf = cbf
        self.order = order
        self.num_robots = num_robots
        self.xdim = xdim
        self.env = env if env is not None else cbf.env
        self.last_status = None

    def compute_control_batched(self, obs, xdes, u_nominal, x_obs=None, obs_r_list=None):
        """obs [E,D,20], xdes [E,D,xdim], u_nominal [E,D,4] -> (u_safe [E,D,4], status [E] int32)."""
        env = self.env
        self.cbf.configure(x_obs, obs_r_list)
        o = to_device(obs, env.device, env.dtype).reshape(env.n, capi.OBS_DIM)
        xd = to_device(xdes, env.device, env.dtype).reshape(env.n, self.xdim)
        un = to_device(u_nominal, env.device, env.dtype).reshape(env.n, 4)
        us = torch.empty((env.NUM_ENVS, env.NUM_DRONES, 4), dtype=env.dtype, device=env.device)
        st = torch.empty((env.NUM_ENVS,), dtype=torch.int32, device=env.device)
        capi.check(env._lib.mds_cbf_filter(env._h, C.c_void_p(o.data_ptr()), C.c_void_p(xd.data_ptr()), C.c_void_p(un.data_ptr()),
                                           C.c_void_p(us.data_ptr()), C.c_void_p(st.data_ptr()), C.c_void_p(stream_ptr(env.device))),
                   "mds_cbf_filter")
        self.last_status = st
        return us, st

    def compute_control(self, obs, xdes, u_nominal, ignore_zmin=False, x_obs=None, obs_r_list=None):
        """Reference signature (qptracker.py:22): obs [N,20], xdes [N,xdim], u_nominal [N,4] -> u [N,4]."""
        env = self.env
        self.cbf.set_xdes(np.asarray(xdes))
        E = env.NUM_ENVS
        bc = lambda a: np.broadcast_to(np.asarray(a, dtype=np.float64), (E,) + np.asarray(a).shape)
        us, st = self.compute_control_batched(bc(obs), bc(xdes), bc(u_nominal), x_obs, obs_r_list)
        if int(st[0].item()) != 0:
            print("QPTracker cannot find cbf-qp controller. Using nominal control.")
            return np.asarray(u_nominal)
        return us[0].double().cpu().numpy().reshape((self.num_robots, -1))
