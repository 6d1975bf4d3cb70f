"""utils/model_conversions.py of the reference (:69-103): RPM <-> (thrust, torques) for the
CF2P "+" frame, evaluated by the HIP kernels behind mds_input_to_action /
mds_action_to_input.  ``env`` is a multidronesim_amd CtrlAviary; arrays are [E,D,4] (or [D,4] /
(4,) for the single-env reference shapes)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _capi as capi
from .._device import default_device_index, require_gpu, stream_ptr, to_device


def _run(env, fn_name, x, *extra):
    numpy_in = not isinstance(x, torch.Tensor)
    shape = tuple(np.shape(x)) if numpy_in else tuple(x.shape)
    xt = to_device(x, env.device, env.dtype).reshape(-1, 4)
    if xt.shape[0] != env.n:                      # single row / partial batch: pad to the handle's n
        pad = torch.zeros((env.n, 4), dtype=env.dtype, device=env.device)
        pad[: xt.shape[0]] = xt
        rows, xt = xt.shape[0], pad
    else:
        rows = env.n
    out = torch.empty_like(xt)
    fn = getattr(env._lib, fn_name)
    capi.check(fn(env._h, C.c_void_p(xt.data_ptr()), *extra, C.c_void_p(out.data_ptr()), C.c_void_p(stream_ptr(env.device))), fn_name)
    out = out[:rows].reshape(shape)
    return out.double().cpu().numpy() if numpy_in else out


def input_to_action(env, u):
    """u = (thrust, tau_x, tau_y, tau_z) -> 4 RPM (model_conversions.py:85-103)."""
    return _run(env, "mds_input_to_action", u)


def action_to_input(env, action, cap_rpm=True):
    """4 RPM -> (thrust, tau_x, tau_y, tau_z) (model_conversions.py:69-83)."""
    return _run(env, "mds_action_to_input", action, C.c_int(1 if cap_rpm else 0))


# ---- observation -> model-state adapters (model_conversions.py:20-58, :105-114, :137-143) -------------------------
# Re-orderings of the 20-float obs (plus one quaternion -> matrix expansion) for code written against the reference's
# helpers.  The fused kernels do these in registers and never call this file; here they run as one small kernel
# (mds_obs_to_model) on the env's device -- like everything else in the package there is no host implementation.

def _obs_to_model(env, obs, dim):
    numpy_in = not isinstance(obs, torch.Tensor)
    shape = tuple(np.shape(obs)) if numpy_in else tuple(obs.shape)
    if shape[-1] != capi.OBS_DIM:
        raise ValueError(f"obs must end in {capi.OBS_DIM} components, got shape {shape}")
    ot = to_device(obs, env.device, env.dtype).reshape(-1, capi.OBS_DIM)
    rows = ot.shape[0]
    if rows > env.n:
        raise ValueError(f"at most {env.n} observations per call (the env's batch), got {rows}")
    if rows != env.n:                              # single row / partial batch: pad to the handle's n
        pad = torch.zeros((env.n, capi.OBS_DIM), dtype=env.dtype, device=env.device)
        pad[:, 6] = 1.0
        pad[:rows] = ot
        ot = pad
    out = torch.empty((env.n, dim), dtype=env.dtype, device=env.device)
    capi.check(env._lib.mds_obs_to_model(env._h, C.c_void_p(ot.data_ptr()), C.c_int(dim), C.c_void_p(out.data_ptr()),
                                         C.c_void_p(stream_ptr(env.device))), "mds_obs_to_model")
    out = out[:rows].reshape(shape[:-1] + (dim,))
    return out.double().cpu().numpy() if numpy_in else out


def obs_to_lin_model(obs, dim=12, env=None):
    """[rpy, (ang_v | F |), vel, pos] for dim 12 / 10 / 9 (:20-58), batched over leading axes.  ``env`` (a multidronesim_amd
    CtrlAviary) is where it runs; the reference needs it only for dim 10 -- here it is always required."""
    if dim not in (9, 10, 12):
        raise ValueError("Invalid dim for linear model")
    if env is None:
        raise ValueError("env must be provided: the conversion runs on the env's GPU (the reference needs it for dim 10 only)")
    return _obs_to_model(env, obs, dim)


def obs_to_geo_model(obs, env=None):
    """x18 = [pos, R(quat) row-major (normalising, as scipy's Rotation.from_quat), vel, ang_v] (:105-114)."""
    if env is None:
        raise ValueError("env must be provided: the conversion runs on the env's GPU")
    return _obs_to_model(env, obs, 18)


def calc_z_thrust(env, obs):
    """KF * sum(rpm^2) of the last clipped action in obs[..., -4:] (:137-143)."""
    return _obs_to_model(env, obs, 10)[..., 3]


# ---- the helpers around the call site of QuadrotorDynamics.dynamics (simulations/CompareModels.py) ------------------
_CODE = {torch.float64: capi.MDS_F64, torch.float32: capi.MDS_F32, torch.float16: capi.MDS_F16}


def _rows(fn_name, x, in_dim, out_dim, out_shape):
    """One stateless row kernel (no handle): NumPy in -> float64 on this rank's GPU -> NumPy out; device tensor in -> tensor out."""
    lib = capi.load_library()
    numpy_in = not isinstance(x, torch.Tensor)
    dev = require_gpu(default_device_index()) if numpy_in or not x.is_cuda else x.device
    dt = torch.float64 if numpy_in else x.dtype
    xt = to_device(x, dev, dt).reshape(-1, in_dim)
    out = torch.empty((xt.shape[0], out_dim), dtype=dt, device=dev)
    capi.check(getattr(lib, fn_name)(C.c_int(_CODE[dt]), C.c_int(xt.shape[0]), C.c_void_p(xt.data_ptr()), C.c_void_p(out.data_ptr()),
                                     C.c_void_p(stream_ptr(dev))), fn_name)
    out = out.reshape(out_shape)
    return out.cpu().numpy() if numpy_in else out


def rpy_to_rot(rpy):
    """R = Rz(yaw) Ry(pitch) Rx(roll) (:4-19); [..., 3] -> [..., 3, 3]."""
    shape = tuple(np.shape(rpy))
    if shape[-1] != 3:
        raise ValueError(f"rpy must end in 3 components, got shape {shape}")
    return _rows("mds_rpy_to_rot", rpy, 3, 9, shape[:-1] + (3, 3))


def geo_model_to_obs(x):
    """[pos, R row-major, vel, ang_v] -> the first 16 observation values: pos, quaternion (xyzw, as scipy's
    Rotation.from_matrix(R).as_quat() gives it), three zeros in the rpy slots, vel, ang_v (:116-122); [..., 18] -> [..., 16]."""
    shape = tuple(np.shape(x))
    if shape[-1] != 18:
        raise ValueError(f"x must end in 18 components, got shape {shape}")
    return _rows("mds_geo_model_to_obs", x, 18, 16, shape[:-1] + (16,))


def geo_x_dot_to_linear(geo_xdot):
    """(v, w, v_dot, w_dot) of QuadrotorDynamics.dynamics -> the linear model's order (w, w_dot, v_dot, v) (:124-135).  A
    re-ordering of twelve numbers with no arithmetic: done on whatever holds them (NumPy array or device tensor)."""
    idx = [3, 4, 5, 9, 10, 11, 6, 7, 8, 0, 1, 2]
    if isinstance(geo_xdot, torch.Tensor):
        return geo_xdot[..., idx]
    return np.asarray(geo_xdot)[..., idx]
