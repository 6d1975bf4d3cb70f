"""ctypes binding of libmds.so (include/mds.h).  Thin: argument marshalling and status
checks only -- all arithmetic lives in the HIP kernels.  There is NO fallback: if the
shared library is missing or a call fails, this raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MDS_LIB_PATH selects another build of the same library (kernel-tuning experiments)
LIB_PATH = os.environ.get("MDS_LIB_PATH") or os.path.join(_HERE, "libmds.so")

MDS_OK = 0
MDS_F32, MDS_F64, MDS_F16, MDS_F32C = 0, 1, 2, 3
MDS_PHYSICS_DYN, MDS_PHYSICS_DYN_DRAG, MDS_PHYSICS_DYN_GND, MDS_PHYSICS_DYN_DW, MDS_PHYSICS_DYN_GND_DRAG_DW = 0, 1, 2, 3, 4
MDS_INTEGRATOR_EULER, MDS_INTEGRATOR_RK4 = 0, 1
MDS_CF2X, MDS_CF2P = 0, 1
OBS_DIM, ACT_DIM, STATE_DIM, DES_DIM, LEM_DIM, GEO_AUX_DIM, SEG_DIM = 20, 4, 13, 11, 7, 13, 40


class MdsConfig(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("num_drones", C.c_int32), ("dtype", C.c_int32), ("physics", C.c_int32),
                ("integrator", C.c_int32), ("drone_model", C.c_int32), ("pyb_freq", C.c_int32),
                ("ctrl_freq", C.c_int32), ("device", C.c_int32), ("track_last_rpm", C.c_int32),
                ("M", C.c_double), ("L", C.c_double), ("KF", C.c_double), ("KM", C.c_double), ("J", C.c_double * 3),
                ("G", C.c_double), ("thrust2weight", C.c_double), ("drag_coeff", C.c_double * 3)]


class MdsGeometricGains(C.Structure):
    _fields_ = [("Kp", C.c_double * 3), ("Kv", C.c_double * 3), ("KR", C.c_double * 3), ("Kw", C.c_double * 3),
                ("g", C.c_double), ("max_tilt_angle", C.c_double)]


class MdsCbfParams(C.Structure):
    _fields_ = [("order", C.c_int32), ("n_obs", C.c_int32), ("max_iter", C.c_int32), ("reserved", C.c_int32),
                ("Kcbf", C.c_double * 3), ("umax", C.c_double * 4), ("safety_radius", C.c_double), ("zscale", C.c_double),
                ("Fmin", C.c_double), ("Fmax", C.c_double), ("tol", C.c_double)]


class MdsDslPidGains(C.Structure):
    _fields_ = [(k, C.c_double * 3) for k in ("P_COEFF_FOR", "I_COEFF_FOR", "D_COEFF_FOR", "P_COEFF_TOR", "I_COEFF_TOR", "D_COEFF_TOR")]


class MdsError(RuntimeError):
    def __init__(self, status, where, detail):
        super().__init__(f"{where}: {detail} (mds_status {status})")
        self.status = status


_P = C.c_void_p
_PD = C.POINTER(C.c_double)

# name -> (restype, argtypes); mirrors include/mds.h one to one
PROTOTYPES = {
    "mds_version": (C.c_int, []),
    "mds_strerror": (C.c_char_p, [C.c_int]),
    "mds_last_error": (C.c_char_p, []),
    "mds_default_config": (C.c_int, [C.c_int, C.POINTER(MdsConfig)]),
    "mds_default_geometric_gains": (C.c_int, [C.POINTER(MdsGeometricGains)]),
    "mds_create": (C.c_int, [C.POINTER(MdsConfig), C.POINTER(_P)]),
    "mds_destroy": (C.c_int, [_P]),
    "mds_get_derived": (C.c_int, [_P, _PD]),
    "mds_reset": (C.c_int, [_P, _PD, _PD, _P]),
    "mds_state_ptrs": (C.c_int, [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p)]),
    "mds_get_state": (C.c_int, [_P, _PD, _P]),
    "mds_set_state": (C.c_int, [_P, _PD, _P]),
    "mds_set_origin": (C.c_int, [_P, _PD, _P]),
    "mds_get_obs": (C.c_int, [_P, _P, _P]),
    "mds_step": (C.c_int, [_P, _P, _P, _P]),
    "mds_set_wind": (C.c_int, [_P, _PD]),
    "mds_set_lemniscate": (C.c_int, [_P, _PD, _P]),
    "mds_set_trajectory_segments": (C.c_int, [_P, _PD, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _PD, C.c_int32, _P]),
    "mds_traj_eval": (C.c_int, [_P, C.c_double, _P, _P]),
    "mds_set_geometric_gains": (C.c_int, [_P, C.POINTER(MdsGeometricGains)]),
    "mds_step_geometric": (C.c_int, [_P, C.c_double, _P, _P, _P]),
    "mds_rollout_geometric": (C.c_int, [_P, C.c_double, C.c_int, _P, C.c_int, _P]),
    "mds_set_rollout_streams": (C.c_int, [_P, C.c_int]),
    "mds_get_last_rollout_streams": (C.c_int, [_P]),
    "mds_rollout_streams_for": (C.c_int, [_P, C.c_int, C.c_int]),
    "mds_set_rollout_form": (C.c_int, [_P, C.c_int, C.c_int]),
    "mds_rollout_form_for": (C.c_int, [_P, C.c_int]),
    "mds_get_last_rollout_form": (C.c_int, [_P]),
    "mds_rollout_step": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P]),
    "mds_rollout_step_fused": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P]),
    "mds_reset_async": (C.c_int, [_P, _P]),
    "mds_rollout_geometric_fused": (C.c_int, [_P, C.c_double, C.c_int, _P, _P, _P]),
    "mds_lemniscate_eval": (C.c_int, [_P, C.c_double, _P, _P]),
    "mds_geometric_compute": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "mds_input_to_action": (C.c_int, [_P, _P, _P, _P]),
    "mds_action_to_input": (C.c_int, [_P, _P, C.c_int, _P, _P]),
    "mds_obs_to_model": (C.c_int, [_P, _P, C.c_int, _P, _P]),
    "mds_quadrotor_dynamics": (C.c_int, [C.c_int, C.c_int, _P, _P, C.c_double, _PD, C.c_double, _P, _P]),
    "mds_compare_models": (C.c_int, [_P, C.c_int, _P, _PD, _PD, C.c_double, C.c_double, _PD, C.c_double, _P, _P, _P, _P]),
    "mds_linear_xdot": (C.c_int, [_P, C.c_int, _P, _P, _PD, _PD, C.c_double, _P, _P]),
    "mds_rpy_to_rot": (C.c_int, [C.c_int, C.c_int, _P, _P, _P]),
    "mds_geo_model_to_obs": (C.c_int, [C.c_int, C.c_int, _P, _P, _P]),
    "mds_cbf_configure": (C.c_int, [_P, C.POINTER(MdsCbfParams), _PD]),
    "mds_cbf_num_rows": (C.c_int, [_P]),
    "mds_cbf_rows": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "mds_cbf_filter": (C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "mds_cbf_last_iterations": (C.c_int, [_P, _P, _P]),
    "mds_lowlevel_reset": (C.c_int, [_P, _P]),
    "mds_thrust_omega_compute": (C.c_int, [_P, _P, _P, _P, _P]),
    "mds_thrust_omega_from_rates": (C.c_int, [_P, _P, _P, _P, _P]),
    "mds_default_dslpid_gains": (C.c_int, [C.POINTER(MdsDslPidGains)]),
    "mds_set_dslpid_gains": (C.c_int, [_P, C.POINTER(MdsDslPidGains)]),
    "mds_dslpid_reset": (C.c_int, [_P, _P]),
    "mds_dslpid_compute": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "mds_step_dslpid": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "mds_rollout_dslpid": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P]),
    "mds_set_lqr_omega_gain": (C.c_int, [_P, _PD]),
    "mds_lqr_omega_compute": (C.c_int, [_P, _P, _P, _P, _P]),
    "mds_set_lqr_gain": (C.c_int, [_P, _PD]),
    "mds_lqr_compute": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "mds_step_lqr": (C.c_int, [_P, C.c_double, _P, _P, _P]),
    "mds_rollout_lqr_fused": (C.c_int, [_P, C.c_double, C.c_int, _P, _P, _P]),
    "mds_set_lqr_yank_omega_gain": (C.c_int, [_P, _PD]),
    "mds_lqr_yank_omega_compute": (C.c_int, [_P, _P, _P, _P, _P]),
    "mds_yank_omega_compute": (C.c_int, [_P, _P, _P, _P, _P]),
    "mds_cbf_set_nominal": (C.c_int, [_P, C.c_int]),
    "mds_cbf_set_step_kernel": (C.c_int, [_P, C.c_int]),
    "mds_cbf_last_step_kernel": (C.c_int, [_P]),
    "mds_step_cbf_geometric": (C.c_int, [_P, C.c_double, _P, _P, _P, _P]),
    "mds_rollout_cbf_geometric": (C.c_int, [_P, C.c_double, C.c_int, _P, _P, _P]),
    "mds_rollout_cbf_geometric_fused": (C.c_int, [_P, C.c_double, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, _P, _P, _P]),
    "mds_step_nominal": (C.c_int, [_P, C.c_double, _P, _P, _P]),
    "mds_rollout_nominal_fused": (C.c_int, [_P, C.c_double, C.c_int, _P, _P, _P]),
}

_lib = None


def load_library(path: str | None = None):
    """dlopen libmds.so and bind every prototype.  Raises (never falls back) when the
    library has not been built -- run ``python -c 'import __graft_entry__ as g; g.build()'``."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise MdsError(-1, "load_library", f"{p} not found: the HIP extension is not built (no CPU fallback exists)")
    try:
        import torch  # noqa: F401  -- loads torch's bundled libamdhip64 first so both share one HIP runtime
    except Exception:  # pragma: no cover - torch is only needed for device tensors
        pass
    lib = C.CDLL(p)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def check(status: int, where: str):
    if status != MDS_OK:
        lib = load_library()
        detail = lib.mds_strerror(status).decode()
        extra = lib.mds_last_error().decode()
        raise MdsError(status, where, f"{detail}; {extra}" if extra else detail)


def as_double_ptr(arr):
    return arr.ctypes.data_as(_PD)
