"""cbf/cbf.py of the reference: ``DroneCBF`` -- exponential control-barrier rows
``G u <= h`` for inter-agent and obstacle avoidance.

Same constructor and ``_build_ineq_const`` / ``set_xdes`` surface as the reference
(:540-580, :308-367, :127-133); the rows are produced by the HIP kernel behind ``mds_cbf_rows``
(closed form of :135-303 for the two hover linearisations, pinned by golden vectors minted from
the reference).  Batched use goes through ``DroneQPTracker.compute_control``."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _capi as capi
from .._device import stream_ptr, to_device
from ..model.linear_omega import LinearizedOmegaModel
from ..model.linear_yank_omega import LinearizedYankOmegaModel


def place_poles_gains(poles):
    """Kcbf of cbf/cbf.py:115-124: place_poles on a chain of integrators = ascending
    coefficients of prod(s - p_i) without the leading one."""
    co = np.poly(np.asarray(poles, dtype=np.float64))
    return co[1:][::-1].copy()


class DroneCBF:
    def __init__(self, env, lin_models, zscale=2.0, safety_radius=1, cbf_poles=np.array([-2.2, -2.4]), room_bounds=None,
                 omega_max=np.array([10, 10, 10]), order=2):
        self.env = env
        self.lin_models = lin_models
        self.num_agents = len(lin_models)
        self.xdim = lin_models[0].A.shape[0]
        if self.num_agents != env.NUM_DRONES:
            raise ValueError("one linear model per drone of the env is required")
        if order not in (2, 3) or self.xdim != {2: 9, 3: 10}[order]:
            raise NotImplementedError("order 2 (LinearizedOmegaModel) and order 3 (LinearizedYankOmegaModel) are built")
        ref = (LinearizedOmegaModel if order == 2 else LinearizedYankOmegaModel)(env)
        for m in lin_models:      # the closed-form rows are those of the hover linearisation
            if not (np.array_equal(np.asarray(m.A), ref.A) and np.array_equal(np.asarray(m.B), ref.B)):
                raise NotImplementedError("DroneCBF rows are built for the reference's hover (A, B) only")
        assert len(cbf_poles) == order, "Number of specified CBF poles ({})does not match order ({})".format(len(cbf_poles), order)
        self.order = order
        self.zscale = zscale
        self.safety_radius = safety_radius
        self.cbf_poles = np.asarray(cbf_poles, dtype=np.float64)
        self.Kcbf = place_poles_gains(cbf_poles).reshape(1, -1)
        self.Fmin = -env.M * env.G
        self.Fmax = env.MAX_THRUST
        Ymax = (env.MAX_THRUST / env.CTRL_TIMESTEP) / 100
        self.umax = np.array([env.MAX_THRUST if order == 2 else Ymax, omega_max[0], omega_max[1], omega_max[2]], dtype=np.float64)
        self.xdes = np.zeros((self.num_agents, self.xdim))
        self.max_iter = 0
        self.tol = 0.0
        self._configured = None

    # ------------------------------------------------------------------
    def set_xdes(self, xdes):
        xdes = np.asarray(xdes)
        if xdes.shape != self.xdes.shape:
            if xdes.shape[0] == self.xdes.shape[0] * self.xdes.shape[1]:
                xdes = xdes.reshape(self.xdes.T.shape).T
            else:
                raise ValueError("xdes shape {} does not match expected shape {}".format(xdes.shape, self.xdes.shape))
        self.xdes = xdes

    def update_zscale(self, zscale):
        ok = zscale > 0
        if ok:
            self.zscale = zscale
        return ok

    def update_safety_radius(self, safety_radius):
        ok = safety_radius > 0
        if ok:
            self.safety_radius = safety_radius
        return ok

    # ------------------------------------------------------------------
    def _obstacle_array(self, x_obs, obs_r_list):
        if x_obs is None or obs_r_list is None or len(obs_r_list) == 0:
            return np.zeros((0, 4))
        assert len(x_obs) == len(obs_r_list), \
            "The lists for Obstacle positions and radii must have the same length. Right now {} & {}".format(len(x_obs), len(obs_r_list))
        xo = np.asarray(x_obs, dtype=np.float64).reshape(len(obs_r_list), -1, 3)
        assert xo.shape[1] == self.order, "Each obstacle state must match robot state of shape (order={}, 3))".format(self.order)
        return np.concatenate([xo[:, 0, :], np.asarray(obs_r_list, dtype=np.float64).reshape(-1, 1)], axis=1)

    def configure(self, x_obs=None, obs_r_list=None):
        """Push parameters + obstacles to the env's handle (mds_cbf_configure); cached."""
        obst = np.ascontiguousarray(self._obstacle_array(x_obs, obs_r_list))
        key = (self.order, self.zscale, self.safety_radius, tuple(self.Kcbf.reshape(-1)), tuple(self.umax), obst.tobytes(),
               self.max_iter, self.tol)
        if key == self._configured:
            return obst.shape[0]
        p = capi.MdsCbfParams()
        p.order, p.n_obs, p.max_iter = self.order, obst.shape[0], int(self.max_iter)
        kc = np.zeros(3)
        kc[:self.order] = self.Kcbf.reshape(-1)
        p.Kcbf = (C.c_double * 3)(*kc)
        p.umax = (C.c_double * 4)(*self.umax)
        p.safety_radius, p.zscale, p.Fmin, p.Fmax, p.tol = float(self.safety_radius), float(self.zscale), float(self.Fmin), float(self.Fmax), float(self.tol)
        env = self.env
        capi.check(env._lib.mds_cbf_configure(env._h, C.byref(p), capi.as_double_ptr(obst) if obst.shape[0] else None),
                   "mds_cbf_configure")
        self._configured = key
        return obst.shape[0]

    def last_iterations(self):
        """[E] int32 device tensor: active-set iterations each env's QP took in the most recent filter launch."""
        env = self.env
        out = torch.empty((env.NUM_ENVS,), dtype=torch.int32, device=env.device)
        capi.check(env._lib.mds_cbf_last_iterations(env._h, C.c_void_p(out.data_ptr()), C.c_void_p(stream_ptr(env.device))),
                   "mds_cbf_last_iterations")
        return out

    def build_ineq_const_batched(self, x, xdes, x_obs=None, obs_r_list=None):
        """x, xdes [E,D,xdim] -> (G [E,m,4D], h [E,m]) device tensors."""
        env = self.env
        self.configure(x_obs, obs_r_list)
        m = env._lib.mds_cbf_num_rows(env._h)
        xt = to_device(x, env.device, env.dtype).reshape(env.n, self.xdim)
        xd = to_device(xdes, env.device, env.dtype).reshape(env.n, self.xdim)
        G = torch.empty((env.NUM_ENVS, m, 4 * env.NUM_DRONES), dtype=env.dtype, device=env.device)
        h = torch.empty((env.NUM_ENVS, m), dtype=env.dtype, device=env.device)
        capi.check(env._lib.mds_cbf_rows(env._h, C.c_void_p(xt.data_ptr()), C.c_void_p(xd.data_ptr()), C.c_void_p(G.data_ptr()),
                                         C.c_void_p(h.data_ptr()), C.c_void_p(stream_ptr(env.device))), "mds_cbf_rows")
        return G, h

    def _build_ineq_const(self, x, ignore_pos_zmin=False, x_obs=None, obs_r_list=None):
        """Reference signature (cbf/cbf.py:308): x [N,xdim] -> (G [m,4N], h [m]) NumPy, env 0."""
        x = np.asarray(x, dtype=np.float64)
        if x.ndim == 1:
            x = x.reshape(self.xdes.T.shape).T
        E = self.env.NUM_ENVS
        G, h = self.build_ineq_const_batched(np.broadcast_to(x, (E,) + x.shape), np.broadcast_to(self.xdes, (E,) + self.xdes.shape),
                                             x_obs, obs_r_list)
        return G[0].double().cpu().numpy(), h[0].double().cpu().numpy()
