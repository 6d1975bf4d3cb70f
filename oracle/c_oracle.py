"""TEST INFRASTRUCTURE ONLY -- ctypes driver of oracle/c_oracle.c (the plain-C, float64 restatement of the reference's control loop).
Used by tests/ as a second checker beside np_oracle.py and by bench.py's cpu_baseline leg (kind "port"); the product never loads it."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libc_oracle.so")
_PD = C.POINTER(C.c_double)


class Consts(C.Structure):
    _fields_ = [("M", C.c_double), ("L", C.c_double), ("KF", C.c_double), ("KM", C.c_double), ("J", C.c_double * 3), ("G", C.c_double),
                ("MAX_RPM", C.c_double), ("MAX_THRUST", C.c_double), ("Kp", C.c_double * 3), ("Kv", C.c_double * 3), ("KR", C.c_double * 3),
                ("Kw", C.c_double * 3), ("g_ctrl", C.c_double), ("max_tilt", C.c_double), ("substeps", C.c_int), ("pyb_dt", C.c_double),
                ("ctrl_dt", C.c_double)]


def build(force=False):
    src = os.path.join(HERE, "c_oracle.c")
    if force or not os.path.exists(SO) or os.path.getmtime(src) > os.path.getmtime(SO):
        subprocess.check_call(["make", "-C", HERE, "-B", "libc_oracle.so"], stdout=subprocess.DEVNULL)
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        assert _lib.co_sizeof_consts() == C.sizeof(Consts), "co_consts layout"
        _lib.co_geometric_loop.restype = C.c_int
    return _lib


def _dp(a):
    return a.ctypes.data_as(_PD)


def consts(pyb_freq=100, ctrl_freq=100):
    c = Consts()
    lib().co_default_consts(C.byref(c), C.c_int(pyb_freq), C.c_int(ctrl_freq))
    return c


class AviaryC:
    """[UPSTREAM] CtrlAviary state machine (Physics.DYN, explicit Euler, CF2P) on the C restatement: reset / step / the whole loop."""

    def __init__(self, init_xyzs, init_rpys, pyb_freq=100, ctrl_freq=100):
        self.c = consts(pyb_freq, ctrl_freq)
        self.xyz = np.ascontiguousarray(np.asarray(init_xyzs, dtype=np.float64).reshape(-1, 3))
        self.rpy = np.ascontiguousarray(np.asarray(init_rpys, dtype=np.float64).reshape(-1, 3))
        self.n = self.xyz.shape[0]
        self.st = np.zeros((self.n, 20))
        self.CTRL_TIMESTEP = 1.0 / ctrl_freq
        lib().co_reset(C.c_int(self.n), _dp(self.xyz), _dp(self.rpy), _dp(self.st))

    def step(self, action):
        a = np.ascontiguousarray(np.asarray(action, dtype=np.float64).reshape(-1, 4))
        obs = np.zeros((self.n, 20))
        lib().co_step(C.byref(self.c), C.c_int(self.n), _dp(self.st), _dp(a), _dp(obs))
        return obs

    def geometric_loop(self, P, steps, t0=0.0, first_zero_step=True, threads=1):
        """-> (last observation [n,20], threads used)"""
        P = np.ascontiguousarray(np.asarray(P, dtype=np.float64).reshape(-1, 7))
        obs = np.zeros((self.n, 20))
        used = lib().co_geometric_loop(C.byref(self.c), C.c_int(self.n), C.c_int(steps), C.c_double(t0), C.c_int(1 if first_zero_step else 0), _dp(P),
                                       _dp(self.st), _dp(obs), C.c_int(threads))
        return obs, used


def lemniscate(t, P):
    P = np.ascontiguousarray(np.asarray(P, dtype=np.float64).reshape(-1, 7))
    des = np.zeros((P.shape[0], 11))
    lib().co_lemniscate(C.c_int(P.shape[0]), C.c_double(t), _dp(P), _dp(des))
    return des


def geometric_compute(obs, des, c=None):
    obs = np.ascontiguousarray(np.asarray(obs, dtype=np.float64).reshape(-1, 20))
    des = np.ascontiguousarray(np.asarray(des, dtype=np.float64).reshape(-1, 11))
    rpm = np.zeros((obs.shape[0], 4))
    c = c or consts()
    lib().co_geometric_compute(C.byref(c), C.c_int(obs.shape[0]), _dp(obs), _dp(des), _dp(rpm))
    return rpm
