"""utils/model_conversions.py of the reference (:69-103): RPM <-> (thrust, torques) for the
CF2P "+" frame, evaluated by the HIP kernels behind mds_input_to_action /
mds_action_to_input.  ``env`` is a multidronesim_amd CtrlAviary; arrays are [E,D,4] (or [D,4] /
(4,) for the single-env reference shapes)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _capi as capi
from .._device import stream_ptr, to_device


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
# Pure re-orderings of the 20-float obs (plus one quaternion -> matrix expansion).  The fused kernels do these in
# registers (csrc/mds_cbf.hpp obs_to_lin, csrc/mds_math.hpp quat_to_rot) and never call this file; these adapters
# exist so that code written against the reference's helpers keeps working on [..., 20] batches, on whichever
# device the obs tensor lives (NumPy in -> NumPy out, as the reference).

def _xp(a):
    return torch if isinstance(a, torch.Tensor) else np


def calc_z_thrust(env, obs):
    """KF * sum(rpm^2) of the last clipped action in obs[..., -4:] (:137-143)."""
    rpms = obs[..., -4:]
    return (env.KF * rpms ** 2).sum(-1)


def obs_to_lin_model(obs, dim=12, env=None):
    """[rpy, (ang_v | F |), vel, pos] for dim 12 / 10 / 9 (:20-58), batched over leading axes."""
    xp = _xp(obs)
    rpy, vel, pos = obs[..., 7:10], obs[..., 10:13], obs[..., 0:3]
    cat = (lambda parts: torch.cat(parts, dim=-1)) if xp is torch else (lambda parts: np.concatenate(parts, axis=-1))
    if dim == 12:
        return cat([rpy, obs[..., 13:16], vel, pos])
    if dim == 9:
        return cat([rpy, vel, pos])
    if dim == 10:
        assert env is not None, "env must be provided for 10 dim model to calculate the thrust"
        return cat([rpy, calc_z_thrust(env, obs)[..., None], vel, pos])
    raise ValueError("Invalid dim for linear model")


def obs_to_geo_model(obs):
    """x18 = [pos, R(quat) row-major (normalising, as scipy's Rotation.from_quat), vel, ang_v] (:105-114)."""
    xp = _xp(obs)
    q = obs[..., 3:7]
    q = q / ((q * q).sum(-1, keepdims=True) if xp is np else (q * q).sum(-1, keepdim=True)) ** 0.5
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    rows = [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
            2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
            2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]
    R = xp.stack(rows, -1) if xp is np else torch.stack(rows, dim=-1)
    cat = (lambda parts: torch.cat(parts, dim=-1)) if xp is torch else (lambda parts: np.concatenate(parts, axis=-1))
    return cat([obs[..., 0:3], R, obs[..., 10:13], obs[..., 13:16]])
