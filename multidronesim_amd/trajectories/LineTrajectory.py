"""trajectories/LineTrajectory.py of the reference: ``WaitTrajectory`` (:4-15) and ``LineTrajectory``
(:17-104: 1 m/s^2 ramp to `speed`, cruise, ramp to the final speed).  The constructor derives the
phase times exactly as the reference does (including its short-distance formula, :52-62, and
``dist_end`` built from ``|v0|``, :49); the piecewise evaluation runs on the GPU."""
import numpy as np

from .base import KIND_LINE, KIND_WAIT, TrajectoryBase, segment_row


class WaitTrajectory(TrajectoryBase):
    def __init__(self, position: np.ndarray, duration: float, yaw=0):
        self.duration = duration
        self.position = position
        self.yaw = yaw

    def get_total_time(self):
        return self.duration

    def _segments(self):
        p = np.asarray(self.position, dtype=np.float64)
        return segment_row(KIND_WAIT, self.duration, [p[0], p[1], p[2], self.yaw])[None, :], False


class LineTrajectory(TrajectoryBase):
    def __init__(self, start: np.ndarray, end: np.ndarray, speed: float = None, duration: float = None, s0=0, sf=0):
        assert speed > 0, "Speed must be positive"
        if duration is not None:
            assert duration > 0, "Duration must be positive"
        self.start = np.asarray(start, dtype=np.float64)
        self.end = np.asarray(end, dtype=np.float64)
        self.delta = self.end - self.start
        self.max_acc = 1.0
        length = np.linalg.norm(self.delta)
        self.speed = length / duration if speed is None else speed
        self.dir = self.delta / length
        self.v0 = s0 * self.dir
        self.vf = sf * self.dir
        self._ramps()
        if self.dist_init + self.dist_end > length:      # too short to reach `speed`: the reference's own peak-speed formula
            self.time_middle = 0
            self.dist_middle = 0
            self.speed = sf + np.sqrt(length * self.max_acc) + 0.5 * s0 ** 2 - 0.5 * sf ** 2
            self._ramps()
        else:
            self.dist_middle = length - self.dist_init - self.dist_end
            self.time_middle = self.dist_middle / self.speed
        self.total_time = self.time_init + self.time_middle + self.time_end

    def _ramps(self):
        self.delta_v_init = self.speed * self.dir - self.v0
        self.delta_v_end = self.vf - self.speed * self.dir
        self.time_init = np.linalg.norm(self.delta_v_init) / self.max_acc
        self.time_end = np.linalg.norm(self.delta_v_end) / self.max_acc
        self.dist_init = np.linalg.norm(self.v0) * self.time_init + 0.5 * self.max_acc * self.time_init ** 2
        self.dist_end = np.linalg.norm(self.v0) * self.time_end + 0.5 * self.max_acc * self.time_end ** 2

    def get_total_time(self):
        return self.total_time

    def _segments(self):
        vmid = self.speed * self.delta / np.linalg.norm(self.delta)
        p = np.concatenate([self.start, self.end, self.v0, self.vf, np.sign(self.delta_v_init) * self.max_acc,
                            np.sign(self.delta_v_end) * self.max_acc, vmid, [self.time_init, self.time_middle, self.total_time]])
        return segment_row(KIND_LINE, self.total_time, p)[None, :], False
