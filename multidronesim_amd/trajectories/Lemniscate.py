"""trajectories/Lemniscate.py of the reference: same constructor, same ``__call__(t)``
5-tuple.  The arithmetic (Lemniscate.py:32-63) runs in the HIP kernels: attached to an env
(``env.set_trajectories``) it is sampled inside the fused step; ``__call__`` evaluates it
through ``mds_lemniscate_eval`` on a private 1-drone handle."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _capi as capi
from .._device import require_gpu, stream_ptr


class TrajectoryBase:
    def __call__(self, t):
        raise NotImplementedError

    def get_total_time(self):
        raise NotImplementedError


class Lemniscate(TrajectoryBase):
    def __init__(self, a=1, omega=.5, center=np.array([0, 0, 0]), yaw_rate=0, revolutions=None, duration=None, phase_shift=0):
        self.a = a
        self.omega = omega
        self.center = center
        self.yaw_rate = yaw_rate
        self.phase_shift = phase_shift
        if revolutions is not None:
            self.total_time = 2 * np.pi * revolutions / omega
        elif duration is not None:
            self.total_time = duration
        else:
            self.total_time = 2 * np.pi / omega
        self._h = None

    def get_total_time(self):
        return self.total_time

    def params(self):
        """(a, omega, centre3, yaw_rate, phase_shift) -- the MDS_LEM_DIM row of include/mds.h."""
        c = np.asarray(self.center, dtype=np.float64)
        return np.array([self.a, self.omega, c[0], c[1], c[2], self.yaw_rate, self.phase_shift], dtype=np.float64)

    def _handle(self):
        if self._h is None:
            lib = capi.load_library()
            dev = require_gpu(0)
            cfg = capi.MdsConfig()
            capi.check(lib.mds_default_config(capi.MDS_CF2P, C.byref(cfg)), "mds_default_config")
            cfg.num_envs, cfg.num_drones, cfg.dtype = 1, 1, capi.MDS_F64
            h = C.c_void_p()
            capi.check(lib.mds_create(C.byref(cfg), C.byref(h)), "mds_create")
            self._h, self._lib, self._dev = h, lib, dev
            self._des = torch.zeros(capi.DES_DIM, dtype=torch.float64, device=dev)
            self._sent = None
        p = self.params()
        if self._sent is None or not np.array_equal(p, self._sent):
            capi.check(self._lib.mds_set_lemniscate(self._h, capi.as_double_ptr(p), C.c_void_p(stream_ptr(self._dev))),
                       "mds_set_lemniscate")
            self._sent = p
        return self._h

    def __call__(self, t):
        h = self._handle()
        capi.check(self._lib.mds_lemniscate_eval(h, C.c_double(float(t)), C.c_void_p(self._des.data_ptr()),
                                                 C.c_void_p(stream_ptr(self._dev))), "mds_lemniscate_eval")
        d = self._des.cpu().numpy()
        return d[0:3].copy(), d[3:6].copy(), d[6:9].copy(), float(d[9]), float(d[10])

    def __del__(self):
        try:
            if self._h is not None:
                self._lib.mds_destroy(self._h)
        except Exception:
            pass
