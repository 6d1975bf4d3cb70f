"""trajectories/Lemniscate.py of the reference: same constructor, same ``__call__(t)`` 5-tuple.
The arithmetic (Lemniscate.py:32-63) runs in the HIP kernels: attached to an env
(``env.set_trajectories``) a batch of Lemniscates is sampled inside the fused fp32 step."""
from __future__ import annotations

import numpy as np

from .base import KIND_LEMNISCATE, TrajectoryBase, segment_row


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

    def get_total_time(self):
        return self.total_time

    def params(self):
        """(a, omega, centre3, yaw_rate, phase_shift) -- the MDS_LEM_DIM row of include/mds.h."""
        c = np.asarray(self.center, dtype=np.float64)
        return np.array([self.a, self.omega, c[0], c[1], c[2], self.yaw_rate, self.phase_shift], dtype=np.float64)

    def _segments(self):
        return segment_row(KIND_LEMNISCATE, self.total_time, self.params())[None, :], False
