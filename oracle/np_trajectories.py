"""TEST INFRASTRUCTURE ONLY -- float64 restatement of the reference's trajectory family
(trajectories/Circle.py, LineTrajectory.py, CompoundTrajectory.py, RotateTrajectory.py), pinned by
tests/golden/trajectories.npz (minted from the reference classes).  Each class returns the
reference's 5-tuple (pos, vel, acc, yaw, yaw_rate) from ``__call__(t)``."""
import numpy as np

from . import np_oracle as O


class Lemniscate:
    def __init__(self, a=1, omega=.5, center=(0, 0, 0), yaw_rate=0, revolutions=None, duration=None, phase_shift=0):
        self.p = (a, omega, np.asarray(center, dtype=np.float64), yaw_rate, phase_shift)
        self.total_time = 2 * np.pi * revolutions / omega if revolutions is not None else (duration if duration is not None else 2 * np.pi / omega)

    def get_total_time(self):
        return self.total_time

    def __call__(self, t):
        a, om, c, yr, ph = self.p
        return O.lemniscate(t, a, om, c, yr, ph)


class Circle:
    """trajectories/Circle.py:24-45."""

    def __init__(self, r=1.0, v=.5, center=(0, 0, 0), yaw_rate=0, revolutions=None, duration=None):
        self.r, self.v, self.c, self.yr = float(r), float(v), np.asarray(center, dtype=np.float64), float(yaw_rate)
        self.total_time = 2 * r * np.pi * revolutions / self.v if revolutions is not None else (duration if duration is not None else 2 * np.pi * self.r / self.v)

    def get_total_time(self):
        return self.total_time

    def __call__(self, t):
        w = self.v / self.r
        s, c = np.sin(w * t), np.cos(w * t)
        pos = np.array([self.c[0] + self.r * c, self.c[1] + self.r * s, self.c[2]])
        vel = np.array([-self.v * s, self.v * c, 0.0])
        acc = np.array([-(self.v ** 2) / self.r * c, -(self.v ** 2) / self.r * s, 0.0])
        yaw = (self.yr * t - np.pi) % (2 * np.pi) + np.pi          # :28 (Python modulo: result in [pi, 3 pi))
        return pos, vel, acc, yaw, self.yr


class Wait:
    """LineTrajectory.py:4-15 WaitTrajectory."""

    def __init__(self, position, duration, yaw=0):
        self.position, self.duration, self.yaw = np.asarray(position, dtype=np.float64), duration, yaw

    def get_total_time(self):
        return self.duration

    def __call__(self, t):
        return self.position, np.zeros(3), np.zeros(3), self.yaw, 0


class Line:
    """LineTrajectory.py:17-104: accelerate at 1 m/s^2 to `speed`, cruise, decelerate (with the
    reference's own formulas for the short-distance case and its dist_end using |v0|)."""

    def __init__(self, start, end, speed=None, duration=None, s0=0, sf=0):
        start, end = np.asarray(start, dtype=np.float64), np.asarray(end, dtype=np.float64)
        delta = end - start
        L = np.linalg.norm(delta)
        amax = 1.0
        spd = L / duration if speed is None else speed
        d = delta / L
        v0, vf = s0 * d, sf * d

        def phases(spd):
            dvi, dve = spd * d - v0, vf - spd * d
            ti, te = np.linalg.norm(dvi) / amax, np.linalg.norm(dve) / amax
            di = np.linalg.norm(v0) * ti + 0.5 * amax * ti ** 2
            de = np.linalg.norm(v0) * te + 0.5 * amax * te ** 2       # sic: |v0| (:49)
            return dvi, dve, ti, te, di, de
        dvi, dve, ti, te, di, de = phases(spd)
        if di + de > L:
            spd = sf + np.sqrt(L * amax) + 0.5 * s0 ** 2 - 0.5 * sf ** 2   # :56
            dvi, dve, ti, te, di, de = phases(spd)
            tm = 0
        else:
            tm = (L - di - de) / spd
        self.start, self.end, self.v0, self.vf = start, end, v0, vf
        self.sgi, self.sge = np.sign(dvi), np.sign(dve)
        self.vmid = spd * delta / L
        self.ti, self.tm, self.te, self.amax = ti, tm, te, amax
        self.total_time = ti + tm + te

    def get_total_time(self):
        return self.total_time

    def __call__(self, t):
        a = self.amax
        if t > self.total_time:
            return self.end, self.vf, np.zeros(3), 0, 0
        if t < self.ti:
            return self.start + self.v0 * t + 0.5 * self.sgi * a * t ** 2, self.v0 + self.sgi * a * t, self.sgi * a, 0, 0
        dpi = self.v0 * self.ti + 0.5 * self.sgi * a * self.ti ** 2
        if t < self.tm + self.ti:
            tl = t - self.ti
            return self.start + dpi + self.vmid * tl, self.vmid, np.zeros(3), 0, 0
        tl = t - self.tm - self.ti
        dpm = dpi + self.vmid * self.tm
        return self.start + dpm + self.vmid * tl + 0.5 * self.sge * a * tl ** 2, self.vmid + self.sge * a * tl, self.sge * a, 0, 0


class Compound:
    """CompoundTrajectory.py:26-40, evaluated statelessly: past the end -> last piece at its own end time;
    otherwise the first piece whose cumulative end time is >= t, at t minus the previous end time."""

    def __init__(self, trajectories):
        self.trajs = list(trajectories)
        self.times = np.cumsum([tr.get_total_time() for tr in self.trajs])
        self.total_time = float(self.times[-1])

    def get_total_time(self):
        return self.total_time

    def __call__(self, t):
        if t >= self.total_time:
            return self.trajs[-1](self.trajs[-1].get_total_time())
        k = 0
        while t > self.times[k]:
            k += 1
        return self.trajs[k](t - (self.times[k - 1] if k > 0 else 0.0))


class Rotate:
    """RotateTrajectory.py:19-25."""

    def __init__(self, trajectory, R, center):
        self.tr, self.R, self.c = trajectory, np.asarray(R, dtype=np.float64), np.asarray(center, dtype=np.float64)

    def get_total_time(self):
        return self.tr.get_total_time()

    def __call__(self, t):
        pos, vel, acc, yaw, om = self.tr(t)
        return self.R @ (pos - self.c) + self.c, self.R @ vel, self.R @ np.asarray(acc, dtype=np.float64) * np.ones(3), yaw, om
