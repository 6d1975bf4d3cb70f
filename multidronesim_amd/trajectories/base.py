"""Shared machinery of the trajectory classes: every trajectory flattens into the segment rows of
``csrc/mds_traj.hpp`` (what ``env.set_trajectories`` uploads) and evaluates ``__call__(t)`` on the GPU
through ``mds_traj_eval`` on a private one-drone float64 handle."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _capi as capi
from .._device import default_device_index, require_gpu, stream_ptr

KIND_LEMNISCATE, KIND_CIRCLE, KIND_LINE, KIND_WAIT = 0, 1, 2, 3


def segment_row(kind, duration, params):
    row = np.zeros(capi.SEG_DIM)
    row[0], row[1], row[2] = kind, 0.0, duration
    row[3:3 + len(params)] = params
    row[27:36] = np.eye(3).reshape(-1)
    return row


class TrajectoryBase:
    def __call__(self, t):
        """-> (pos, vel, acc, yaw, yaw_rate), evaluated by the HIP kernel."""
        h = self._handle()
        capi.check(self._lib.mds_traj_eval(h, C.c_double(float(t)), C.c_void_p(self._des.data_ptr()), C.c_void_p(stream_ptr(self._dev))),
                   "mds_traj_eval")
        d = self._des.cpu().numpy()
        return d[0:3].copy(), d[3:6].copy(), d[6:9].copy(), float(d[9]), float(d[10])

    def get_total_time(self):
        raise NotImplementedError

    # ---- flattening ------------------------------------------------------------------
    def _segments(self):
        """-> (rows [k, SEG_DIM] with cumulative t_start/t_end, compound flag)."""
        raise NotImplementedError

    def anchor(self):
        """A point near the trajectory (used as the drone's local-frame origin)."""
        rows, _ = self._segments()
        r = rows[0]
        p = r[3:27]
        kind = int(r[0])
        base = {KIND_LEMNISCATE: p[2:5], KIND_CIRCLE: p[2:5], KIND_LINE: p[0:3], KIND_WAIT: p[0:3]}[kind]
        return r[27:36].reshape(3, 3) @ base + r[36:39]

    # ---- private evaluation handle -----------------------------------------------------
    def _handle(self):
        rows, compound = self._segments()
        key = rows.tobytes() + bytes([compound])
        if getattr(self, "_h", None) is None:
            lib = capi.load_library()
            dev = require_gpu(default_device_index())        # this rank's GPU, never a hard-coded device 0
            cfg = capi.MdsConfig()
            capi.check(lib.mds_default_config(capi.MDS_CF2P, C.byref(cfg)), "mds_default_config")
            cfg.num_envs, cfg.num_drones, cfg.dtype = 1, 1, capi.MDS_F64
            cfg.device = dev.index
            h = C.c_void_p()
            capi.check(lib.mds_create(C.byref(cfg), C.byref(h)), "mds_create")
            self._h, self._lib, self._dev = h, lib, dev
            self._des = torch.zeros(capi.DES_DIM, dtype=torch.float64, device=dev)
            self._sent = None
        if self._sent != key:
            upload_segments(self._lib, self._h, [self], self._dev)
            self._sent = key
        return self._h

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None:
                self._lib.mds_destroy(self._h)
        except Exception:
            pass


def upload_segments(lib, handle, trajs, device):
    """Flatten one trajectory per drone and hand the tables to mds_set_trajectory_segments."""
    rows, offsets, compound, anchors = [], [0], [], []
    for tr in trajs:
        r, c = tr._segments()
        rows.append(r)
        offsets.append(offsets[-1] + r.shape[0])
        compound.append(1 if c else 0)
        anchors.append(tr.anchor())
    segs = np.ascontiguousarray(np.concatenate(rows, axis=0), dtype=np.float64)
    off = np.ascontiguousarray(offsets, dtype=np.int32)
    comp = np.ascontiguousarray(compound, dtype=np.int32)
    anc = np.ascontiguousarray(anchors, dtype=np.float64)
    capi.check(lib.mds_set_trajectory_segments(handle, capi.as_double_ptr(segs), off.ctypes.data_as(C.POINTER(C.c_int32)),
                                               comp.ctypes.data_as(C.POINTER(C.c_int32)), capi.as_double_ptr(anc),
                                               C.c_int32(segs.shape[0]), C.c_void_p(stream_ptr(device))),
               "mds_set_trajectory_segments")
