"""[UPSTREAM] gym_pybullet_drones.control.DSLPIDControl with the surface the reference uses
(PIDEnv.py:124-134,166-169; MultiDroneExample.py): ``DSLPIDControl(drone_model=...)``, mutable
``P/I/D_COEFF_FOR/TOR`` arrays and ``computeControlFromState(control_timestep, state, target_pos,
target_rpy)``.  The class is not in the reference tree; its arithmetic is restated from the published
upstream source (parity unpinned) and runs in the HIP kernels (mds_dslpid_compute / mds_step_dslpid).

Stand-alone objects own a private one-drone float64 handle (PID memory lives on the GPU);
``MultiDroneEnv`` instead pushes the gains to its env and uses the fused controller+physics kernel."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _capi as capi
from .._device import default_device_index, require_gpu, stream_ptr
from ..utils.enums import DroneModel


def gains_struct(ctrl) -> capi.MdsDslPidGains:
    g = capi.MdsDslPidGains()
    for name in ("P_COEFF_FOR", "I_COEFF_FOR", "D_COEFF_FOR", "P_COEFF_TOR", "I_COEFF_TOR", "D_COEFF_TOR"):
        setattr(g, name, (C.c_double * 3)(*np.asarray(getattr(ctrl, name), dtype=np.float64)))
    return g


class DSLPIDControl:
    def __init__(self, drone_model: DroneModel, g: float = 9.8):
        if isinstance(drone_model, str):
            drone_model = DroneModel(drone_model)
        if drone_model not in (DroneModel.CF2X, DroneModel.CF2P):
            raise NotImplementedError("DSLPIDControl requires DroneModel.CF2X or DroneModel.CF2P")
        self.DRONE_MODEL = drone_model
        self.GRAVITY = g * 0.027
        self.KF, self.KM = 3.16e-10, 7.94e-12
        self.P_COEFF_FOR = np.array([.4, .4, 1.25])
        self.I_COEFF_FOR = np.array([.05, .05, .05])
        self.D_COEFF_FOR = np.array([.2, .2, .5])
        self.P_COEFF_TOR = np.array([70000., 70000., 60000.])
        self.I_COEFF_TOR = np.array([.0, .0, 500.])
        self.D_COEFF_TOR = np.array([20000., 20000., 12000.])
        self.PWM2RPM_SCALE, self.PWM2RPM_CONST = 0.2685, 4070.3
        self.MIN_PWM, self.MAX_PWM = 20000, 65535
        self.control_counter = 0
        self._h = None

    def _handle(self, control_timestep):
        if self._h is None:
            lib = capi.load_library()
            dev = require_gpu(default_device_index())
            cfg = capi.MdsConfig()
            capi.check(lib.mds_default_config(capi.MDS_CF2P if self.DRONE_MODEL == DroneModel.CF2P else capi.MDS_CF2X, C.byref(cfg)),
                       "mds_default_config")
            f = int(round(1.0 / control_timestep))
            cfg.num_envs, cfg.num_drones, cfg.dtype, cfg.pyb_freq, cfg.ctrl_freq = 1, 1, capi.MDS_F64, f, f
            cfg.device = dev.index
            h = C.c_void_p()
            capi.check(lib.mds_create(C.byref(cfg), C.byref(h)), "mds_create")
            self._h, self._lib, self._dev, self._dt = h, lib, dev, 1.0 / f
            self._buf = torch.zeros(20 + 3 + 3 + 4, dtype=torch.float64, device=dev)
        elif abs(control_timestep - self._dt) > 1e-12:
            raise ValueError("control_timestep changed between calls")
        capi.check(self._lib.mds_set_dslpid_gains(self._h, C.byref(gains_struct(self))), "mds_set_dslpid_gains")
        return self._h

    def reset(self):
        self.control_counter = 0
        if self._h is not None:
            capi.check(self._lib.mds_dslpid_reset(self._h, C.c_void_p(stream_ptr(self._dev))), "mds_dslpid_reset")

    def computeControlFromState(self, control_timestep, state, target_pos, target_rpy=np.zeros(3), target_vel=np.zeros(3),
                                target_rpy_rates=np.zeros(3)):
        """-> (rpm[4], pos_e[3], yaw_e) like upstream; target_vel / target_rpy_rates must be zero (all the reference passes)."""
        if np.any(np.asarray(target_vel) != 0) or np.any(np.asarray(target_rpy_rates) != 0):
            raise NotImplementedError("non-zero target_vel / target_rpy_rates")
        h = self._handle(control_timestep)
        state = np.asarray(state, dtype=np.float64)
        host = np.concatenate([state[:20], np.asarray(target_pos, dtype=np.float64), np.asarray(target_rpy, dtype=np.float64), np.zeros(4)])
        self._buf.copy_(torch.as_tensor(host))
        b = self._buf
        # 16-byte alignment of the rpm slot: 26 doubles = 208 bytes
        capi.check(self._lib.mds_dslpid_compute(h, C.c_void_p(b.data_ptr()), C.c_void_p(b[20:].data_ptr()), C.c_void_p(b[23:].data_ptr()),
                                                C.c_void_p(b[26:].data_ptr()), C.c_void_p(stream_ptr(self._dev))), "mds_dslpid_compute")
        self.control_counter += 1
        rpm = b[26:30].cpu().numpy()
        return rpm, np.asarray(target_pos, dtype=np.float64) - state[0:3], 0.0

    def __del__(self):
        try:
            if self._h is not None:
                self._lib.mds_destroy(self._h)
        except Exception:
            pass
