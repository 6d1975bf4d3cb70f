"""trajectories/RotateTrajectory.py of the reference (:19-25): pos' = R (pos - c) + c, vel' = R vel,
acc' = R acc, yaw unchanged -- folded into each segment's affine map (A, b)."""
import numpy as np

from .base import TrajectoryBase


class RotateTrajectory(TrajectoryBase):
    def __init__(self, trajectory: TrajectoryBase, R: np.ndarray, center: np.ndarray):
        self.trajectory, self.R, self.center = trajectory, R, center

    @property
    def total_time(self):
        return self.trajectory.get_total_time()

    def get_total_time(self):
        return self.total_time

    def _segments(self):
        rows, compound = self.trajectory._segments()
        rows = rows.copy()
        R, c = np.asarray(self.R, dtype=np.float64), np.asarray(self.center, dtype=np.float64)
        for r in rows:
            A, b = r[27:36].reshape(3, 3), r[36:39].copy()
            r[27:36] = (R @ A).reshape(-1)
            r[36:39] = R @ (b - c) + c
        return rows, compound
