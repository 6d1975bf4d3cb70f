"""simulations/CompareModels.py of the reference: the geometric rigid-body model against the hover linearisation on a logged
rollout.  The reference walks the observation history in Python (:48-56) -- per row ``linear.calc_xdot_from_obs(obs)``,
``geo_x_dot_to_linear(geo_dynamics.dynamics(None, obs_to_geo_model(obs), action_to_input(env, obs[16:])))`` and
``obs_to_lin_model(obs)`` --; here that loop body is ONE kernel launch over all T x D rows (``mds_compare_models``), and
``roll_out_linear_system`` (:84-98) integrates the linear model with the same scipy ``solve_ivp`` whose right-hand side is
``LinearizedModel.calc_xdot`` on the GPU.  The matplotlib figures (:59-80) are outside the path: ``main`` returns the arrays."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _capi as capi
from .._device import stream_ptr, to_device
from ..model._eval import eval_handle
from ..model.dynamics import QuadrotorDynamics
from ..model.linearized import LinearizedModel
from .EnvGeometric import GeometricEnv, parse_args


def compare_models(linear: LinearizedModel, geo_dynamics: QuadrotorDynamics, observations):
    """The loop of :48-56 over ``observations`` [..., 20] -> (x_dot_linear, x_dot_geometric, x_lin_obs), each [..., 12].
    NumPy in -> float64 on the GPU -> NumPy out; a device tensor (e.g. the observation log of a fused rollout) stays on the device
    in its own element type."""
    numpy_in = not isinstance(observations, torch.Tensor)
    shape = tuple(np.shape(observations))
    if shape[-1] != capi.OBS_DIM:
        raise ValueError(f"observations must end in {capi.OBS_DIM} components, got shape {shape}")
    ev = eval_handle(linear.env, torch.float64 if numpy_in else observations.dtype)
    ot = to_device(observations, ev.dev, ev.dtype).reshape(-1, capi.OBS_DIM)
    n = ot.shape[0]
    outs = [torch.empty((n, 12), dtype=ev.dtype, device=ev.dev) for _ in range(3)]
    A, B = linear._mats()
    J = (C.c_double * 3)(geo_dynamics.J[0, 0], geo_dynamics.J[1, 1], geo_dynamics.J[2, 2])
    capi.check(ev.lib.mds_compare_models(ev.h, C.c_int(n), C.c_void_p(ot.data_ptr()), capi.as_double_ptr(A), capi.as_double_ptr(B),
                                         C.c_double(linear.mass * linear.g), C.c_double(geo_dynamics.m), J, C.c_double(geo_dynamics.g),
                                         C.c_void_p(outs[0].data_ptr()), C.c_void_p(outs[1].data_ptr()), C.c_void_p(outs[2].data_ptr()),
                                         C.c_void_p(stream_ptr(ev.dev))), "mds_compare_models")
    outs = [o.reshape(shape[:-1] + (12,)) for o in outs]
    return tuple(o.cpu().numpy() for o in outs) if numpy_in else tuple(outs)


def roll_out_linear_system(linear, observations, obs_ts):
    """:84-98: the linear model integrated from the first observation, driven by the logged actions (zero-order hold on the
    closest observation in the past); ``observations`` [T, 20] of one drone."""
    from scipy.integrate import solve_ivp
    observations = np.asarray(observations)
    obs_ts = np.asarray(obs_ts)

    def f(t, x):
        closest_idx = int(np.argmin(np.abs(obs_ts - t)))
        if obs_ts[closest_idx] > t:
            closest_idx -= 1
        return linear.calc_xdot(x, observations[closest_idx][16:])

    o0 = observations[0]
    x0 = np.concatenate([o0[7:10], o0[13:16], o0[10:13], o0[0:3]])                 # obs_to_lin_model(obs), dim 12: a re-ordering of one row
    return solve_ivp(f, [0, obs_ts[-1]], x0, t_eval=obs_ts)


def main(argv=None, roll_out=False):
    args = parse_args(argv)
    geometric = GeometricEnv(args, circle_init=True)
    geometric.TARGET_POSITIONS[0, :] = np.array([3, 3, 1.5])
    geometric.TARGET_RPYS[0, :] = np.array([0, 0, 0])
    env = geometric.create_env(gui=False)
    geometric.do_control()
    observations = np.array(geometric.observations).squeeze()                      # [T, D, 20]
    obs_ts = np.array(geometric.obs_ts).squeeze()
    linear = LinearizedModel(env)
    geo_dynamics = QuadrotorDynamics(env.PYB_FREQ)
    geo_dynamics.load_env_params(env)
    x_dot_linear, x_dot_geometric, x_lin_obs = compare_models(linear, geo_dynamics, observations)
    res = roll_out_linear_system(linear, observations[:, 0] if observations.ndim == 3 else observations, obs_ts) if roll_out else None
    return dict(observations=observations, obs_ts=obs_ts, x_dot_linear=x_dot_linear, x_dot_geometric=x_dot_geometric, x_lin_obs=x_lin_obs,
                roll_out=res)


if __name__ == "__main__":
    out = main()
    d = np.abs(out["x_dot_linear"] - out["x_dot_geometric"]).max(axis=tuple(range(out["x_dot_linear"].ndim - 1)))
    print("max |x_dot_linear - x_dot_geometric| per component:", np.array2string(d, precision=3))
    print("done.")
