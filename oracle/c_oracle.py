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
        r = subprocess.run(["make", "-C", HERE, "-B", "libc_oracle.so"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"make -C oracle failed ({r.returncode}):\n{r.stdout}")
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


def lqr_loop(av, P, K, steps, wind=None, t0=0.0, first_zero_step=True, threads=1):
    """The 'lqr' do_control loop of simulations/EnvGeometric.py on the AviaryC `av`: -> (last observation [n,20], threads used)."""
    P = np.ascontiguousarray(np.asarray(P, dtype=np.float64).reshape(-1, 7))
    K = np.ascontiguousarray(np.asarray(K, dtype=np.float64).reshape(4, 12))
    w = None if wind is None else np.ascontiguousarray(np.asarray(wind, dtype=np.float64).reshape(3))
    obs = np.zeros((av.n, 20))
    lib().co_lqr_loop.restype = C.c_int
    used = lib().co_lqr_loop(C.byref(av.c), C.c_int(av.n), C.c_int(steps), C.c_double(t0), C.c_int(1 if first_zero_step else 0), _dp(P), _dp(K),
                             None if w is None else _dp(w), _dp(av.st), _dp(obs), C.c_int(threads))
    return obs, used


def lqr12_compute(obs, des, K, c=None):
    obs = np.ascontiguousarray(np.asarray(obs, dtype=np.float64).reshape(-1, 20))
    des = np.ascontiguousarray(np.asarray(des, dtype=np.float64).reshape(-1, 11))
    K = np.ascontiguousarray(np.asarray(K, dtype=np.float64).reshape(4, 12))
    rpm = np.zeros((obs.shape[0], 4))
    c = c or consts()
    lib().co_lqr12_compute(C.byref(c), C.c_int(obs.shape[0]), _dp(K), _dp(obs), _dp(des), _dp(rpm))
    return rpm


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


# ---- the CBF-filtered loop (BASELINE config 4) ----------------------------------------------------------------------------
CO_MAXOBS = 8


class Cbf(C.Structure):
    _fields_ = [("Kcbf", C.c_double * 3), ("umax", C.c_double * 4), ("safety_radius", C.c_double), ("zscale", C.c_double), ("n_obs", C.c_int),
                ("order", C.c_int), ("obs_xyz", (C.c_double * 3) * CO_MAXOBS), ("obs_r", C.c_double * CO_MAXOBS), ("Fmin", C.c_double),
                ("Fmax", C.c_double)]


def cbf_params(Kcbf, umax, safety_radius, zscale, x_obs=None, obs_r=None, order=2, Fmin=None, Fmax=None):
    """x_obs as the reference passes it (simulations/CBFTest.py:421-425): one (order, 3) state per sphere, position in row 0."""
    assert lib().co_sizeof_cbf() == C.sizeof(Cbf), "co_cbf layout"
    b = Cbf()
    kk = [float(k) for k in np.asarray(Kcbf).reshape(-1)[:3]]
    b.Kcbf[:] = kk + [0.0] * (3 - len(kk))
    b.order = int(order)
    cc = consts()
    b.Fmin = -cc.M * cc.G if Fmin is None else float(Fmin)
    b.Fmax = cc.MAX_THRUST if Fmax is None else float(Fmax)
    b.umax[:] = [float(k) for k in np.asarray(umax).reshape(-1)[:4]]
    b.safety_radius, b.zscale = float(safety_radius), float(zscale)
    n = 0 if obs_r is None else len(obs_r)
    assert n <= CO_MAXOBS
    b.n_obs = n
    for j in range(n):
        xo = np.asarray(x_obs[j], dtype=np.float64).reshape(-1, 3)[0]
        b.obs_xyz[j][:] = [float(v) for v in xo]
        b.obs_r[j] = float(obs_r[j])
    return b


def cbf_rows(x, xdes, b, c=None):
    """CBF._build_ineq_const for one env: x, xdes [D,9] (order 2) or [D,10] (order 3) -> (G [m,4D], h [m]) in the reference's row order."""
    xd = 9 if b.order == 2 else 10
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1, xd))
    xdes = np.ascontiguousarray(np.asarray(xdes, dtype=np.float64).reshape(-1, xd))
    D = x.shape[0]
    mmax = D * (D - 1) // 2 + 8 * D + D * b.n_obs + (2 * D if b.order == 3 else 0)
    G, h = np.zeros((mmax, 4 * D)), np.zeros(mmax)
    c = c or consts()
    fn = lib().co_cbf_rows if b.order == 2 else lib().co_cbf_rows3
    m = fn(C.byref(c), C.byref(b), C.c_int(D), _dp(x), _dp(xdes), _dp(G), _dp(h))
    assert m == mmax
    return G, h


def qp_project(uhat, G, h):
    """-> (solved, u, iterations)"""
    G = np.ascontiguousarray(np.asarray(G, dtype=np.float64))
    h = np.ascontiguousarray(np.asarray(h, dtype=np.float64))
    u = np.ascontiguousarray(np.asarray(uhat, dtype=np.float64).reshape(-1).copy())
    it = C.c_int(0)
    ok = lib().co_qp_project(C.c_int(u.size), C.c_int(h.size), _dp(G), _dp(h), _dp(u), C.byref(it))
    if ok < 0:
        raise ValueError("QP larger than the C oracle's static bounds")
    return bool(ok), u, it.value


class CbfLoopC:
    """simulations/CBFTest.py:303-350 for E envs of D drones on the C restatement (geometric nominal, order-2 filter, ThrustOmega low level)."""

    def __init__(self, xyz, rpy, b, pyb_freq=100, ctrl_freq=100, first_rpm=0.0):
        xyz = np.asarray(xyz, dtype=np.float64)
        self.E, self.D = xyz.shape[0], xyz.shape[1]
        self.av = AviaryC(xyz.reshape(-1, 3), np.asarray(rpy, dtype=np.float64).reshape(-1, 3), pyb_freq, ctrl_freq)
        self.b = b
        self.pid = np.zeros((self.E * self.D, 6))
        self.av.step(np.full((self.E * self.D, 4), float(first_rpm)))       # env.step(zeros) before the loop (:296-300); order 3: hover RPM

    def run3(self, P, steps, K_yank_omega, t0=0.0, threads=1):
        """The order-3 loop (simulations/CBFTestOrd3.py:306-352) -> (obs [E,D,20], statuses [steps,E], solver iterations, threads used)."""
        P = np.ascontiguousarray(np.asarray(P, dtype=np.float64).reshape(-1, 7))
        K = np.ascontiguousarray(np.asarray(K_yank_omega, dtype=np.float64).reshape(4, 10))
        obs = np.zeros((self.E * self.D, 20))
        st = np.zeros((steps, self.E), dtype=np.int32)
        tot = C.c_longlong(0)
        lib().co_cbf3_loop.restype = C.c_int
        used = lib().co_cbf3_loop(C.byref(self.av.c), C.byref(self.b), C.c_int(self.E), C.c_int(self.D), C.c_int(steps), C.c_double(t0), _dp(P),
                                  _dp(self.av.st), _dp(self.pid), _dp(obs), st.ctypes.data_as(C.POINTER(C.c_int)), C.byref(tot), C.c_int(threads), _dp(K))
        if used < 0:
            raise ValueError("env larger than the C oracle's static bounds")
        return obs.reshape(self.E, self.D, 20), st, tot.value, used

    def run(self, P, steps, t0=0.0, threads=1, K_lqr_omega=None):
        """-> (obs [E,D,20], statuses [steps,E], solver iterations in all, threads used).  K_lqr_omega [4,9]: the LQR-omega nominal
        controller of simulations/CBFTest.py:290-293 instead of the geometric one."""
        P = np.ascontiguousarray(np.asarray(P, dtype=np.float64).reshape(-1, 7))
        obs = np.zeros((self.E * self.D, 20))
        st = np.zeros((steps, self.E), dtype=np.int32)
        tot = C.c_longlong(0)
        lib().co_cbf_loop.restype = C.c_int
        used = lib().co_cbf_loop(C.byref(self.av.c), C.byref(self.b), C.c_int(self.E), C.c_int(self.D), C.c_int(steps), C.c_double(t0), _dp(P),
                                 _dp(self.av.st), _dp(self.pid), _dp(obs), st.ctypes.data_as(C.POINTER(C.c_int)), C.byref(tot), C.c_int(threads),
                                 None if K_lqr_omega is None else _dp(np.ascontiguousarray(np.asarray(K_lqr_omega, dtype=np.float64).reshape(4, 9))))
        if used < 0:
            raise ValueError("env larger than the C oracle's static bounds")
        return obs.reshape(self.E, self.D, 20), st, tot.value, used
