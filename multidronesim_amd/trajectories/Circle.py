"""trajectories/Circle.py of the reference (:24-45): constant-speed circle, yaw ramp wrapped to [pi, 3 pi)."""
import numpy as np

from .base import KIND_CIRCLE, TrajectoryBase, segment_row


class CircleTrajectory(TrajectoryBase):
    def __init__(self, r=1.0, v=.5, center=np.array([0, 0, 0]), yaw_rate=0, revolutions=None, duration=None):
        self.r, self.v, self.yaw_rate = float(r), float(v), float(yaw_rate)
        self.center = center
        lap = 2 * np.pi * self.r / self.v                  # one revolution at speed v
        self.total_time = lap * revolutions if revolutions is not None else (duration if duration is not None else lap)

    def get_total_time(self):
        return self.total_time

    def _segments(self):
        c = np.asarray(self.center, dtype=np.float64)
        return segment_row(KIND_CIRCLE, self.total_time, [self.r, self.v, c[0], c[1], c[2], self.yaw_rate])[None, :], False
