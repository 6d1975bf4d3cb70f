"""Private evaluation handles for the model helpers (LinearizedModel.calc_xdot*, the CompareModels loop body).

The reference builds its models from an env and keeps using them after ``env.close()`` (simulations/CompareModels.py:17-31: the
rollout is done, the env closed, then the models are compared on the logged observations).  Here the env's own handle is gone by
then, so the helpers run on a one-drone handle of their own that carries the env's constants (what action_to_input reads) in the
element type of the data they are given -- created on first use, on the env's GPU, destroyed with the env object."""
from __future__ import annotations

import ctypes as C

import torch

from .. import _capi as capi
from .._device import TORCH_DTYPE, require_gpu

_CODE = {torch.float64: capi.MDS_F64, torch.float32: capi.MDS_F32, torch.float16: capi.MDS_F16}


class _EvalHandle:
    def __init__(self, cfg, code, device_index):
        self.lib = capi.load_library()
        self.dev = require_gpu(device_index)
        c = capi.MdsConfig()
        C.memmove(C.byref(c), C.byref(cfg), C.sizeof(capi.MdsConfig))
        c.num_envs, c.num_drones, c.dtype, c.device, c.track_last_rpm = 1, 1, code, self.dev.index, 0
        c.physics = capi.MDS_PHYSICS_DYN
        self.h = C.c_void_p()
        capi.check(self.lib.mds_create(C.byref(c), C.byref(self.h)), "mds_create")
        self.dtype = TORCH_DTYPE[code]

    def __del__(self):
        try:
            if self.h is not None:
                self.lib.mds_destroy(self.h)
                self.h = None
        except Exception:
            pass


def eval_handle(env, dtype) -> _EvalHandle:
    """The evaluation handle of ``env`` (a multidronesim_amd CtrlAviary, open or closed) for element type ``dtype``."""
    cache = env.__dict__.setdefault("_eval_handles", {})
    code = _CODE[dtype]
    if code not in cache:
        cache[code] = _EvalHandle(env._cfg, code, env.device.index)
    return cache[code]
