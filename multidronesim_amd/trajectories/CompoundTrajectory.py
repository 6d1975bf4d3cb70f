"""trajectories/CompoundTrajectory.py of the reference (:26-40): pieces played back to back; past the
end the last piece's final pose.  Flattened into cumulative-time segments and looked up on the GPU."""
import numpy as np

from .base import TrajectoryBase


class CompoundTrajectory(TrajectoryBase):
    def __init__(self, trajectories: list):
        self.trajectories = list(trajectories)
        durations = [float(tr.get_total_time()) for tr in self.trajectories]
        self.times = np.cumsum(durations)                 # end time of every piece
        self.total_time = float(self.times[-1])

    def get_total_time(self):
        return self.total_time

    def reset(self):
        """The reference keeps a playback cursor and rewinds it here; the GPU lookup is stateless."""
        return None

    def _segments(self):
        rows, t0 = [], 0.0
        for tr in self.trajectories:
            r, _ = tr._segments()
            r = r.copy()
            if r.shape[0] == 1:                      # a leaf: occupies [t0, t0 + its total time]
                r[0, 1], r[0, 2] = t0, t0 + tr.get_total_time()
            else:                                    # a nested compound: shift its own cumulative times
                r[:, 1] += t0
                r[:, 2] += t0
            rows.append(r)
            t0 += tr.get_total_time()
        return np.concatenate(rows, axis=0), True
