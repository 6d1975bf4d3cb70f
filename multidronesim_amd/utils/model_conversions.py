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
