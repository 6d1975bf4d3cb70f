"""TEST TOOLING ONLY: ctypes driver for the g++ build of the device arithmetic
(tests/emul/mds_emul.cpp).  Used by CPU tests to study fp32 precision against the oracle
before spending GPU time; the product never loads this."""
import ctypes as C
import os
import subprocess

import numpy as np

from multidronesim_amd._capi import MdsConfig, MdsGeometricGains, MDS_CF2P

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libmds_emul.so")
SRC = os.path.join(HERE, "mds_emul.cpp")
_PD = C.POINTER(C.c_double)


def build(force=False):
    deps = [SRC] + [os.path.join(HERE, "..", "..", "multidronesim_amd", "csrc", f) for f in ("mds_math.hpp", "mds_consts.hpp")]
    if force or not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-mfma", "-ffp-contract=fast", "-o", SO, SRC])
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        for sfx in ("f32", "f64"):
            getattr(_lib, f"emul_create_{sfx}").restype = C.c_void_p
    return _lib


def dp(a):
    return a.ctypes.data_as(_PD)


class Emul:
    def __init__(self, dtype="f32", num_envs=1, num_drones=1, pyb_freq=100, ctrl_freq=100, physics=0, integrator=0,
                 model=MDS_CF2P):
        L = lib()
        comp = dtype.startswith("f32c")  # fp32 arithmetic, compensated accumulation: "f32c" = MDS_F32C as the device stores it (residuals
        mask = {"f32c": 8, "f32c13": 15}.get(dtype, 8) if comp else 0      # of the body rates only); "f32c13" = all 13, "f32c:<mask>" = study
        if comp and ":" in dtype:
            mask = int(dtype.split(":")[1])
        dtype = "f32" if comp else dtype
        self.sfx = dtype
        self.cfg = MdsConfig()
        self.gains = MdsGeometricGains()
        L.emul_default_config(model, C.byref(self.cfg), C.byref(self.gains))
        self.cfg.num_envs, self.cfg.num_drones = num_envs, num_drones
        self.cfg.pyb_freq, self.cfg.ctrl_freq = pyb_freq, ctrl_freq
        self.cfg.physics, self.cfg.integrator = physics, integrator
        self.n = num_envs * num_drones
        self.h = C.c_void_p(getattr(L, f"emul_create_{dtype}")(C.byref(self.cfg), C.byref(self.gains)))
        if comp:
            self._f("emul_set_comp")(self.h, C.c_int(1))
            L.emul_set_comp_mask_f32(self.h, C.c_int(mask))

    def _f(self, name):
        return getattr(lib(), f"{name}_{self.sfx}")

    def set_state(self, s):
        s = np.ascontiguousarray(s, dtype=np.float64)
        self._f("emul_set_state")(self.h, dp(s))

    def get_state(self):
        out = np.zeros((self.n, 13))
        self._f("emul_get_state")(self.h, dp(out))
        return out

    def set_lemniscate(self, p):
        p = np.ascontiguousarray(p, dtype=np.float64)
        self._f("emul_set_lem")(self.h, dp(p))

    def step(self, action):
        a = np.ascontiguousarray(action, dtype=np.float64)
        obs = np.zeros((self.n, 20))
        self._f("emul_step")(self.h, dp(a), dp(obs))
        return obs

    def step_geometric(self, t):
        obs = np.zeros((self.n, 20))
        act = np.zeros((self.n, 4))
        self._f("emul_step_geo")(self.h, C.c_double(t), dp(obs), dp(act))
        return obs, act

    def geometric_compute(self, obs, des):
        obs = np.ascontiguousarray(obs, dtype=np.float64)
        des = np.ascontiguousarray(des, dtype=np.float64)
        n = obs.shape[0]
        rpm = np.zeros((n, 4))
        aux = np.zeros((n, 13))
        self._f("emul_geo_compute")(self.h, C.c_int(n), dp(obs), dp(des), dp(rpm), dp(aux))
        return rpm, aux

    def compare_models(self, obs, A, B, ueq0, dyn_m, dyn_J, dyn_g):
        obs = np.ascontiguousarray(obs, dtype=np.float64)
        A, B = np.ascontiguousarray(A, dtype=np.float64), np.ascontiguousarray(B, dtype=np.float64)
        J = np.ascontiguousarray(dyn_J, dtype=np.float64)
        n = obs.shape[0]
        outs = [np.zeros((n, 12)) for _ in range(3)]
        self._f("emul_compare_models")(self.h, C.c_int(n), dp(obs), dp(A), dp(B), C.c_double(ueq0), C.c_double(dyn_m), dp(J), C.c_double(dyn_g),
                                       dp(outs[0]), dp(outs[1]), dp(outs[2]))
        return outs                                                      # xdot_lin, xdot_geo, x_lin

    def lemniscate(self, t):
        des = np.zeros((self.n, 11))
        self._f("emul_lem")(self.h, C.c_double(t), dp(des))
        return des


def sincos_f32(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    s = np.zeros_like(x)
    c = np.zeros_like(x)
    PF = C.POINTER(C.c_float)
    lib().emul_sincos_f32(C.c_int(x.size), x.ctypes.data_as(PF), s.ctypes.data_as(PF), c.ctypes.data_as(PF))
    return s, c


def rpy_to_rot(rpy, dtype="f64"):
    rpy = np.ascontiguousarray(rpy, dtype=np.float64)
    R = np.zeros((rpy.shape[0], 9))
    getattr(lib(), f"emul_rpy_to_rot_{dtype}")(C.c_int(rpy.shape[0]), dp(rpy), dp(R))
    return R.reshape(-1, 3, 3)


def rot_to_quat(R, dtype="f64"):
    R = np.ascontiguousarray(np.asarray(R, dtype=np.float64).reshape(-1, 9))
    q = np.zeros((R.shape[0], 4))
    getattr(lib(), f"emul_rot_to_quat_{dtype}")(C.c_int(R.shape[0]), dp(R), dp(q))
    return q
