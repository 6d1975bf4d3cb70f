"""trajectories/Circle.py of the reference (:24-45): constant-speed circle, yaw ramp wrapped to [pi, 3 pi)."""
import numpy as np

from .base import KIND_CIRCLE, TrajectoryBase, segment_row


class CircleTrajectory(TrajectoryBase):
    def __init__(self, r=1.0, v=.5, center=np.array([0, 0, 0]), yaw_rate=0, revolutions=None, duration=None):
        self.r = float(r)
        self.v = float(v)
        self.center = center
        self.yaw_rate = float(yaw_rate)
        if revolutions is not None:
            self.total_time = 2 * r * np.pi * revolutions / self.v
        elif duration is not None:
            self.total_time = duration
        else:
            self.total_time = 2 * np.pi * self.r / self.v

    def get_total_time(self):
        return self.total_time

    def _segments(self):
        c = np.asarray(self.center, dtype=np.float64)
        return segment_row(KIND_CIRCLE, self.total_time, [self.r, self.v, c[0], c[1], c[2], self.yaw_rate])[None, :], False
