"""CPU-side checks of the C-ABI boundary: libmds.so loads, exports every symbol that
include/mds.h declares, validates arguments before touching the device, and fails LOUDLY
(never silently falls back) where no GPU exists."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import multidronesim_amd
from multidronesim_amd import _capi as capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return capi.load_library()


def header_symbols():
    src = open(os.path.join(ROOT, "include", "mds.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mds_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = header_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"libmds.so does not export {n}"
    assert sorted(capi.PROTOTYPES) == names, "ctypes prototypes and include/mds.h disagree"


def test_version_and_strerror(lib):
    assert lib.mds_version() == 202
    assert lib.mds_strerror(0) == b"ok"
    assert b"aligned" in lib.mds_strerror(-4)
    assert lib.mds_strerror(-99) == b"unknown status"


def test_default_config_matches_urdf_constants(lib):
    cfg = capi.MdsConfig()
    assert lib.mds_default_config(capi.MDS_CF2P, C.byref(cfg)) == 0
    assert (cfg.M, cfg.L, cfg.KF, cfg.KM, cfg.G, cfg.thrust2weight) == (0.027, 0.0397, 3.16e-10, 7.94e-12, 9.8, 2.25)
    assert list(cfg.J) == [2.3951e-5, 2.3951e-5, 3.2347e-5]
    assert lib.mds_default_config(capi.MDS_CF2X, C.byref(cfg)) == 0
    assert list(cfg.J) == [1.4e-5, 1.4e-5, 2.17e-5]
    assert lib.mds_default_config(7, C.byref(cfg)) == capi.MDS_OK - 1
    g = capi.MdsGeometricGains()
    assert lib.mds_default_geometric_gains(C.byref(g)) == 0
    assert list(g.Kp) == [2.25] * 3 and list(g.Kv) == [3.5] * 3 and list(g.KR) == [125.0] * 3 and list(g.Kw) == [10.0] * 3
    assert g.g == 9.81 and g.max_tilt_angle == pytest.approx(40 * np.pi / 180)


def test_struct_layout_matches_header(lib):
    # 10 int32 + (4 + 3 + 2 + 3) doubles ; 12 + 2 doubles
    assert C.sizeof(capi.MdsConfig) == 10 * 4 + 12 * 8
    assert C.sizeof(capi.MdsGeometricGains) == 14 * 8


@pytest.mark.parametrize("mut", [dict(num_envs=0), dict(num_drones=-1), dict(dtype=9), dict(dtype=4), dict(physics=5), dict(integrator=2),
                                 dict(drone_model=3), dict(pyb_freq=240, ctrl_freq=100), dict(ctrl_freq=0), dict(M=0.0),
                                 dict(KF=-1.0)])
def test_create_rejects_bad_config_before_touching_the_device(lib, mut):
    cfg = capi.MdsConfig()
    lib.mds_default_config(capi.MDS_CF2P, C.byref(cfg))
    for k, v in mut.items():
        setattr(cfg, k, v)
    h = C.c_void_p()
    assert lib.mds_create(C.byref(cfg), C.byref(h)) == -1
    assert not h.value
    assert lib.mds_last_error() != b""
    assert lib.mds_create(None, C.byref(h)) == -1


def test_null_handle_calls_return_einval(lib):
    assert lib.mds_step(None, None, None, None) == -1
    assert lib.mds_step_geometric(None, 0.0, None, None, None) == -1
    assert lib.mds_get_obs(None, None, None) == -1
    assert lib.mds_step_lqr(None, 0.0, None, None, None) == -1
    assert lib.mds_step_nominal(None, 0.0, None, None, None) == -1
    assert lib.mds_step_cbf_geometric(None, 0.0, None, None, None, None) == -1
    assert lib.mds_obs_to_model(None, None, 9, None, None) == -1
    assert lib.mds_lqr_compute(None, None, None, None, None, None) == -1
    assert lib.mds_yank_omega_compute(None, None, None, None, None) == -1
    assert lib.mds_set_lqr_gain(None, None) == -1
    assert lib.mds_destroy(None) == 0
    # round-2 entry points
    assert lib.mds_set_rollout_streams(None, 2) == -1
    assert lib.mds_get_last_rollout_streams(None) == -1
    assert lib.mds_rollout_streams_for(None, 0, 100) == -1
    assert lib.mds_set_rollout_form(None, 2, 50) == -1 and lib.mds_rollout_form_for(None, 100) == -1 and lib.mds_get_last_rollout_form(None) == -1
    assert lib.mds_cbf_last_iterations(None, None, None) == -1
    assert lib.mds_rollout_geometric(None, 0.0, 5, None, 1, None) == -1
    assert lib.mds_rollout_cbf_geometric(None, 0.0, 5, None, None, None) == -1
    assert lib.mds_rollout_dslpid(None, None, None, 1, 0, 5, None, 0, None) == -1
    assert lib.mds_cbf_set_step_kernel(None, 1) == -1
    assert lib.mds_cbf_last_step_kernel(None) == -1
    # round-3 entry points
    assert lib.mds_rollout_cbf_geometric_fused(None, 0.0, 5, 5, None, 0, 0, None, None, None, None) == -1


def test_compensated_dtype_is_a_valid_config_and_env_effects_reject_it(lib):
    """MDS_F32C passes mds_create's validation (it then fails on the missing device here, not on the dtype); with ground effect /
    downwash physics it is refused before the device is touched."""
    cfg = capi.MdsConfig()
    lib.mds_default_config(capi.MDS_CF2P, C.byref(cfg))
    cfg.dtype = capi.MDS_F32C
    h = C.c_void_p()
    rc = lib.mds_create(C.byref(cfg), C.byref(h))
    import torch
    if torch.cuda.is_available():
        assert rc == 0
        lib.mds_destroy(h)
    else:
        assert rc in (-3, -2) and not h.value          # HIP error: no device -- not MDS_EINVAL
    cfg.physics = capi.MDS_PHYSICS_DYN_GND
    assert lib.mds_create(C.byref(cfg), C.byref(h)) == -1
    cfg.dtype = 4
    cfg.physics = capi.MDS_PHYSICS_DYN
    assert lib.mds_create(C.byref(cfg), C.byref(h)) == -1


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from multidronesim_amd.envs.CtrlAviary import CtrlAviary
    with pytest.raises(multidronesim_amd.MdsError):
        CtrlAviary(num_drones=2)
    from multidronesim_amd.trajectories.Lemniscate import Lemniscate
    with pytest.raises(multidronesim_amd.MdsError):
        Lemniscate()(0.0)


def test_missing_library_is_an_error(tmp_path):
    with pytest.raises(multidronesim_amd.MdsError):
        capi.load_library(str(tmp_path / "libmds.so"))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "multidronesim_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "np_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f
                assert "c_oracle" not in txt and "libc_oracle" not in txt, f
                assert "tests.emul" not in txt and "libmds_emul" not in txt, f


def test_reference_api_names_present():
    """Names the reference imports / calls exist with the same spelling."""
    from multidronesim_amd.envs.BaseAviary import DroneModel, Physics
    from multidronesim_amd.envs.CtrlAviary import CtrlAviary
    from multidronesim_amd.control.geometric import GeometricControl
    from multidronesim_amd.control.base_controller import BaseController
    from multidronesim_amd.trajectories.Lemniscate import Lemniscate, TrajectoryBase
    from multidronesim_amd.utils.utils import sync, str2bool
    from multidronesim_amd.utils.model_conversions import input_to_action, action_to_input
    from multidronesim_amd.PIDEnv import MultiDroneEnv
    from multidronesim_amd import MultiDroneExample
    a = MultiDroneExample.parse_args(["--num_drones", "3", "--control_freq_hz", "48", "--simulation_freq_hz", "240"])
    assert (a.num_drones, a.control_freq_hz, a.simulation_freq_hz, a.init_rad, a.duration_sec) == (3, 48, 240, 1.0, 30)
    from multidronesim_amd.model.dynamics import QuadrotorDynamics
    # the reference's package-level imports (control/__init__.py, model/__init__.py, cbf/__init__.py, utils/__init__.py)
    from multidronesim_amd.control import (GeometricControl as G2, LQRController, LQROmegaController, LQRYankOmegaController,  # noqa: F401
                                           ThrustOmegaController, YankOmegaController)
    from multidronesim_amd.model import LinearizedModel, LinearizedOmegaModel, LinearizedYankOmegaModel  # noqa: F401
    from multidronesim_amd.cbf import DroneCBF, DroneQPTracker  # noqa: F401
    from multidronesim_amd.utils import obs_to_lin_model, obs_to_geo_model, calc_z_thrust  # noqa: F401
    assert G2 is GeometricControl
    assert DroneModel("cf2p") is DroneModel.CF2P and Physics("pyb") is Physics.PYB
    for m in ("step", "reset", "render", "close", "getPyBulletClient", "getDroneIds", "_showDroneLocalAxes"):
        assert callable(getattr(CtrlAviary, m))
    for m in ("set_desired_trajectory", "compute"):
        assert callable(getattr(GeometricControl, m)) and callable(getattr(BaseController, m))
    for m in ("threaded_sim", "run_sim", "sim_step", "stop_sim", "stop", "build_args"):
        assert callable(getattr(MultiDroneEnv, m))
    assert str2bool("yes") is True and str2bool("0") is False
    t = Lemniscate(a=2, omega=0.5, center=np.array([1, 2, 3]), yaw_rate=0.1, phase_shift=0.3)
    np.testing.assert_allclose(t.params(), [2, 0.5, 1, 2, 3, 0.1, 0.3])
    assert Lemniscate(omega=2.0, revolutions=3).get_total_time() == pytest.approx(3 * np.pi)
    m = MultiDroneEnv(num_drones=3, gui=False)
    assert m.INIT_XYZS.shape == (3, 3) and m.TARGET_POSITIONS[1, 2] == 1.0
    q = QuadrotorDynamics(100)
    with pytest.raises(ValueError):
        q.step(np.zeros(4))


def test_simulation_mirrors_set_up_on_cpu_and_need_the_gpu_to_run():
    """simulations/{EnvGeometric,EnvGeometricOmega,EnvGeometricYankOmega,CBFTest,CBFTestOrd3,CompareModels}.py mirrors: argument parsing and the initial-condition arithmetic are host
    code (as in the reference); creating the env without a GPU is a loud error, not a CPU simulation."""
    import numpy as np
    from multidronesim_amd import MdsError
    from multidronesim_amd.simulations import CBFTest, CBFTestOrd3, EnvGeometric
    a = EnvGeometric.parse_args([])
    assert (a.controller, a.num_drones, a.duration_sec, a.init_rad, a.num_envs) == ("lqr", 2, 30, 1.0, 1)       # EnvGeometric.py:26-32,61
    assert CBFTest.parse_args([]).init_rad == .2 and EnvGeometric.wind_force == .00025
    geo = EnvGeometric.GeometricEnv(EnvGeometric.parse_args(["--num_drones", "4"]), circle_init=True)
    np.testing.assert_allclose(geo.INIT_XYZS[1], [0.0, 1.0, 0.0], atol=1e-15)                                      # (i-1)/N * 2 pi, sin/cos (:507-510)
    np.testing.assert_allclose(geo.TARGET_POSITIONS[:, 2], 1.0)
    np.testing.assert_allclose(geo.TARGET_RPYS[:, 2], np.pi / 2)
    g3 = CBFTestOrd3.GeometricEnv(CBFTestOrd3.parse_args(["--num_drones", "3"]), init_type="circle", center=np.array([0, 0, 0.5]))
    np.testing.assert_allclose(g3.INIT_XYZS[1], [.2 * np.cos(2 * np.pi / 3), .2 * np.sin(2 * np.pi / 3), 0.5], atol=1e-15)   # cos/sin + centre (:418-425)
    # simulations/EnvGeometricOmega.py (:28, :57, :364-365: sin/cos at i/N) and EnvGeometricYankOmega.py (:28-30, :389-390: cos/sin at i/N)
    from multidronesim_amd.simulations import CompareModels, EnvGeometricOmega, EnvGeometricYankOmega
    ao, ay = EnvGeometricOmega.parse_args([]), EnvGeometricYankOmega.parse_args([])
    assert (ao.duration_sec, ao.num_drones, ao.init_rad, ao.controller) == (50, 2, .2, "lqr")
    assert (ay.duration_sec, ay.num_drones, ay.init_rad, ay.controller) == (5, 1, .2, "lqr")
    go = EnvGeometricOmega.GeometricEnv(EnvGeometricOmega.parse_args(["--num_drones", "3"]), circle_init=True)
    gy = EnvGeometricYankOmega.GeometricEnv(EnvGeometricYankOmega.parse_args(["--num_drones", "3"]), circle_init=True)
    np.testing.assert_allclose(go.INIT_XYZS[1], [.2 * np.sin(2 * np.pi / 3), .2 * np.cos(2 * np.pi / 3), 0.0], atol=1e-15)
    np.testing.assert_allclose(gy.INIT_XYZS[1], [.2 * np.cos(2 * np.pi / 3), .2 * np.sin(2 * np.pi / 3), 0.0], atol=1e-15)
    np.testing.assert_allclose(gy.TARGET_RPYS[:, 2], np.pi / 2)
    assert callable(CompareModels.compare_models) and callable(CompareModels.roll_out_linear_system) and CompareModels.parse_args is EnvGeometric.parse_args
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(MdsError):
            geo.create_env()
        with pytest.raises(MdsError):
            go.create_env()
        with pytest.raises(MdsError):
            CBFTestOrd3.GeometricEnv(CBFTestOrd3.parse_args([]), init_type="lemniscate")       # evaluates a trajectory: device work


def test_library_is_plain_c_abi_without_torch(lib):
    """The drop-in boundary is a C-ABI shared object: it links the HIP runtime and the C/C++ runtimes, nothing of PyTorch,
    and exports unmangled mds_* entry points only (plus compiler-generated HIP registration symbols)."""
    import subprocess
    out = subprocess.run(["ldd", capi.LIB_PATH], capture_output=True, text=True).stdout
    deps = " ".join(l.split()[0] for l in out.splitlines() if l.strip())          # library names only (addresses are hex noise)
    assert "libamdhip64" in deps
    assert "torch" not in deps and "c10" not in deps and "python" not in deps.lower()
    syms = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True).stdout
    exported = [l.split()[-1] for l in syms.splitlines() if " T " in l]
    api = [n for n in exported if n.startswith("mds_")]
    assert sorted(api) == header_symbols()


def test_create_rejects_more_than_2_30_drones(lib):
    cfg = capi.MdsConfig()
    lib.mds_default_config(capi.MDS_CF2P, C.byref(cfg))
    cfg.num_envs, cfg.num_drones = 1 << 20, 1 << 11
    h = C.c_void_p()
    assert lib.mds_create(C.byref(cfg), C.byref(h)) == -1 and not h.value
    assert b"too many" in lib.mds_last_error()


def test_default_dslpid_gains(lib):
    """[UPSTREAM] DSLPIDControl coefficients as mds_default_dslpid_gains hands them out (PIDEnv.py:124-134 halves these)."""
    g = capi.MdsDslPidGains()
    assert lib.mds_default_dslpid_gains(C.byref(g)) == 0
    assert list(g.P_COEFF_FOR) == [.4, .4, 1.25] and list(g.I_COEFF_FOR) == [.05, .05, .05] and list(g.D_COEFF_FOR) == [.2, .2, .5]
    assert list(g.P_COEFF_TOR) == [70000., 70000., 60000.] and list(g.I_COEFF_TOR) == [.0, .0, 500.] and list(g.D_COEFF_TOR) == [20000., 20000., 12000.]
    assert lib.mds_default_dslpid_gains(None) == -1


def test_out_of_scope_crazyflie_names_resolve_and_raise():
    """SURVEY section 2 #11 (OUT OF SCOPE): the reference's `from model import CrazyflieModel` / `from control import CrazyflieLQR` resolve,
    and constructing either says why it is not built."""
    from multidronesim_amd.control import CrazyflieLQR
    from multidronesim_amd.model import CrazyflieModel
    from multidronesim_amd.utils import Environment
    env = Environment(G=9.8, M=0.027, MAX_THRUST=0.6, CTRL_TIMESTEP=0.01)
    assert (env.G, env.M, env.CTRL_TIMESTEP) == (9.8, 0.027, 0.01) and env.DRONE_MODEL.value == "cf2x"
    for cls in (CrazyflieModel, CrazyflieLQR):
        with pytest.raises(NotImplementedError):
            cls(env)
