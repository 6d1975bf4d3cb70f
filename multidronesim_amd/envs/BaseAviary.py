"""Batched replacement of [UPSTREAM] gym_pybullet_drones.envs.BaseAviary.

The reference drives this class through ``CtrlAviary(...)`` (PIDEnv.py:106-116,
simulations/EnvGeometric.py:89-100), ``env.step(action)`` (EnvGeometric.py:469),
``env.render()`` (:475), ``env.close()`` (:481) and the attribute census of SURVEY.md
section 3.4.  Here every drone of every env is stepped by one HIP kernel launch
(``mds_step`` / ``mds_step_geometric`` in include/mds.h); Bullet is not involved.
"""
from __future__ import annotations

import ctypes as C
import time

import numpy as np
import torch

from .. import _capi as capi
from .._device import DTYPE_BY_NAME, TORCH_DTYPE, default_device_index, forget_env_device, note_env_device, require_gpu, stream_ptr, to_device
from ..utils.enums import DroneModel, Physics

__all__ = ["BaseAviary", "DroneModel", "Physics"]

_PHYSICS_MAP = {Physics.DYN: capi.MDS_PHYSICS_DYN, Physics.PYB: capi.MDS_PHYSICS_DYN,
                Physics.PYB_DRAG: capi.MDS_PHYSICS_DYN_DRAG, Physics.PYB_GND: capi.MDS_PHYSICS_DYN_GND,
                Physics.PYB_DW: capi.MDS_PHYSICS_DYN_DW, Physics.PYB_GND_DRAG_DW: capi.MDS_PHYSICS_DYN_GND_DRAG_DW}
_MODEL_MAP = {DroneModel.CF2X: capi.MDS_CF2X, DroneModel.CF2P: capi.MDS_CF2P}
_PYB_SUBSTITUTED = (Physics.PYB, Physics.PYB_DRAG, Physics.PYB_GND, Physics.PYB_DW, Physics.PYB_GND_DRAG_DW)


class BaseAviary:
    """Gym-style multi-drone env, batched over ``num_envs`` independent copies.

    Shapes follow the reference when ``num_envs == 1`` and NumPy goes in: ``step(action[D,4])
    -> obs[D,20]``.  With a torch tensor in (any ``num_envs``) everything stays on the GPU:
    ``step(action[E,D,4]) -> obs[E,D,20]``.  obs layout ([UPSTREAM] _getDroneStateVector):
    pos3 | quat4 xyzw | rpy3 | vel3 | ang_v3 (world) | last clipped RPM4.

    ``Physics.PYB`` (the reference's default, PIDEnv.py:19) is served by the explicit
    ``Physics.DYN`` rigid-body model -- there is no Bullet here; ``PYB_DRAG`` adds upstream's
    ``_drag`` term; ``PYB_GND`` / ``PYB_DW`` / ``PYB_GND_DRAG_DW`` add upstream's ``_groundEffect`` / ``_downwash`` as extra
    terms of that model (``env.step(action)`` only: explicit Euler, f32 / f64; spec-level).  The first env
    created with a ``PYB*`` mode says so once (RuntimeWarning).
    """

    _warned_pyb = False

    def __init__(self, drone_model: DroneModel = DroneModel.CF2X, num_drones: int = 1, neighbourhood_radius: float = np.inf,
                 initial_xyzs=None, initial_rpys=None, physics: Physics = Physics.PYB, pyb_freq: int = 240,
                 ctrl_freq: int = 240, gui=False, record=False, obstacles=False, user_debug_gui=True,
                 output_folder="results", *, num_envs: int = 1, dtype="float32", integrator: str = "euler",
                 device: int | None = None, track_last_rpm: bool | None = None):
        lib = capi.load_library()
        if isinstance(drone_model, str):
            drone_model = DroneModel(drone_model)
        if isinstance(physics, str):
            physics = Physics(physics)
        if drone_model not in _MODEL_MAP:
            raise NotImplementedError(f"drone model {drone_model} (only CF2X / CF2P urdf constants are built in)")
        if physics not in _PHYSICS_MAP:
            raise NotImplementedError(f"physics {physics}")
        if pyb_freq % ctrl_freq != 0:
            raise ValueError("pyb_freq is not divisible by env_freq.")  # [UPSTREAM] BaseAviary.__init__
        if physics in _PYB_SUBSTITUTED and not BaseAviary._warned_pyb:
            import warnings
            BaseAviary._warned_pyb = True
            warnings.warn(f"Physics.{physics.name}: there is no Bullet here -- served by the explicit-Euler Physics.DYN rigid-body model "
                          "(no ground plane, no btMultiBody semi-implicit step); trajectories differ from the reference's PyBullet "
                          "runs from the first step (the reference starts on the floor at z = 0). Pass Physics.DYN to select it knowingly.",
                          RuntimeWarning, stacklevel=3)
        if device is None:
            device = default_device_index()
        else:
            note_env_device(id(self), device.index if isinstance(device, torch.device) else int(device))
        self.device = require_gpu(device)
        self._lib = lib
        self.DRONE_MODEL, self.PHYSICS = drone_model, physics
        self.NUM_DRONES, self.NUM_ENVS = int(num_drones), int(num_envs)
        self.NEIGHBOURHOOD_RADIUS = neighbourhood_radius
        self.PYB_FREQ, self.CTRL_FREQ = int(pyb_freq), int(ctrl_freq)
        self.PYB_STEPS_PER_CTRL = self.PYB_FREQ // self.CTRL_FREQ
        self.CTRL_TIMESTEP, self.PYB_TIMESTEP = 1.0 / self.CTRL_FREQ, 1.0 / self.PYB_FREQ
        self.GUI, self.RECORD, self.OBSTACLES, self.USER_DEBUG = gui, record, obstacles, user_debug_gui
        self.OUTPUT_FOLDER = output_folder
        self._dtype_code = DTYPE_BY_NAME[dtype]
        self.dtype = TORCH_DTYPE[self._dtype_code]

        cfg = capi.MdsConfig()
        capi.check(lib.mds_default_config(_MODEL_MAP[drone_model], C.byref(cfg)), "mds_default_config")
        cfg.num_envs, cfg.num_drones = self.NUM_ENVS, self.NUM_DRONES
        cfg.dtype = self._dtype_code
        cfg.physics = _PHYSICS_MAP[physics]
        cfg.integrator = {"euler": capi.MDS_INTEGRATOR_EULER, "rk4": capi.MDS_INTEGRATOR_RK4}[integrator]
        cfg.pyb_freq, cfg.ctrl_freq = self.PYB_FREQ, self.CTRL_FREQ
        cfg.device = self.device.index
        # _computeObs()[..., 16:20] after a step: kept by the library when the env is used the reference's way (one env);
        # batched envs read the last clipped RPM from the obs step() returns and skip the extra 16 B per drone-step
        cfg.track_last_rpm = int(self.NUM_ENVS == 1 if track_last_rpm is None else track_last_rpm)
        self._cfg = cfg
        # urdf constants ([UPSTREAM] _parseURDFParameters) + derived attributes the reference reads off env
        self.M, self.L, self.KF, self.KM = cfg.M, cfg.L, cfg.KF, cfg.KM
        self.THRUST2WEIGHT_RATIO = cfg.thrust2weight
        self.J = np.diag([cfg.J[0], cfg.J[1], cfg.J[2]])
        self.J_INV = np.linalg.inv(self.J)
        self.G = cfg.G
        self.DRAG_COEFF = np.array([cfg.drag_coeff[0], cfg.drag_coeff[1], cfg.drag_coeff[2]])
        h = C.c_void_p()
        capi.check(lib.mds_create(C.byref(cfg), C.byref(h)), "mds_create")
        self._h = h
        d = (C.c_double * 8)()
        capi.check(lib.mds_get_derived(self._h, d), "mds_get_derived")
        (self.GRAVITY, self.HOVER_RPM, self.MAX_RPM, self.MAX_THRUST, self.MAX_XY_TORQUE, self.MAX_Z_TORQUE) = list(d)[:6]
        self.n = self.NUM_ENVS * self.NUM_DRONES
        self.DRONE_IDS = np.arange(1, self.NUM_DRONES + 1)
        self.CLIENT = -1
        # initial poses ([UPSTREAM] defaults: a line of drones 4L apart, just above the floor)
        if initial_xyzs is None:
            initial_xyzs = np.vstack([np.array([x * 4 * self.L for x in range(self.NUM_DRONES)]),
                                      np.array([y * 4 * self.L for y in range(self.NUM_DRONES)]),
                                      np.ones(self.NUM_DRONES) * 0.1125]).T
        if initial_rpys is None:
            initial_rpys = np.zeros((self.NUM_DRONES, 3))
        self.INIT_XYZS = self._broadcast_init(initial_xyzs)
        self.INIT_RPYS = self._broadcast_init(initial_rpys)
        self._obs = torch.zeros((self.NUM_ENVS, self.NUM_DRONES, capi.OBS_DIM), dtype=self.dtype, device=self.device)
        self._act = torch.zeros((self.NUM_ENVS, self.NUM_DRONES, capi.ACT_DIM), dtype=self.dtype, device=self.device)
        self._has_traj = False
        self.step_counter = 0
        self.RESET_TIME = time.time()
        self._housekeeping()

    # ------------------------------------------------------------------ helpers
    def _broadcast_init(self, a):
        a = np.asarray(a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a, dtype=np.float64)
        if a.shape == (self.NUM_DRONES, 3):
            a = np.broadcast_to(a, (self.NUM_ENVS, self.NUM_DRONES, 3))
        if a.shape != (self.NUM_ENVS, self.NUM_DRONES, 3):
            raise ValueError(f"initial_xyzs/initial_rpys must be ({self.NUM_DRONES},3) or "
                             f"({self.NUM_ENVS},{self.NUM_DRONES},3), got {a.shape}")
        return np.ascontiguousarray(a)

    def _stream(self):
        return C.c_void_p(stream_ptr(self.device))

    def _require_open(self):
        if self._h is None:
            raise capi.MdsError(capi.MDS_OK - 5, "BaseAviary", "environment is closed")

    def _housekeeping(self):
        """[UPSTREAM] _housekeeping: poses from INIT_XYZS / INIT_RPYS, zero velocity and action."""
        self._require_open()
        xyz = self.INIT_XYZS.reshape(-1, 3)
        rpy = self.INIT_RPYS.reshape(-1, 3)
        capi.check(self._lib.mds_reset(self._h, capi.as_double_ptr(xyz), capi.as_double_ptr(rpy), self._stream()), "mds_reset")
        self.step_counter = 0

    def _out(self, obs: torch.Tensor, numpy_out: bool):
        if not numpy_out:
            return obs
        o = obs.detach().to("cpu", torch.float64).numpy()
        return o[0] if self.NUM_ENVS == 1 else o

    # ------------------------------------------------------------------ gym API
    def reset(self, seed: int = None, options: dict = None):
        """gymnasium ``reset() -> (obs, info)``.  (The reference never calls it -- SURVEY.md 3.3.)"""
        self._housekeeping()
        obs = self._computeObs()
        return self._out(obs, True) if not getattr(self, "_tensor_mode", False) else obs, self._computeInfo()

    def step(self, action):
        """[UPSTREAM] BaseAviary.step: clip RPM, PYB_FREQ//CTRL_FREQ physics substeps, 20-float obs.
        Returns ``(obs, reward, terminated, truncated, info)`` exactly like CtrlAviary."""
        self._require_open()
        numpy_in = not isinstance(action, torch.Tensor)
        self._tensor_mode = not numpy_in
        act = to_device(action, self.device, self.dtype)
        if act.numel() != self.n * capi.ACT_DIM:
            raise ValueError(f"action must hold {self.NUM_ENVS}x{self.NUM_DRONES}x4 RPMs, got shape {tuple(act.shape)}")
        capi.check(self._lib.mds_step(self._h, C.c_void_p(act.data_ptr()), C.c_void_p(self._obs.data_ptr()), self._stream()),
                   "mds_step")
        self.step_counter += self.PYB_STEPS_PER_CTRL
        return (self._out(self._obs, numpy_in), self._computeReward(), self._computeTerminated(),
                self._computeTruncated(), self._computeInfo())

    def render(self, mode="human", close=False):
        """[UPSTREAM] BaseAviary.render: text dump of every drone of env 0 (EnvGeometric.py:475)."""
        o = self._computeObs().detach().to("cpu", torch.float64).numpy()[0]
        print("\n[INFO] BaseAviary.render() ——— it {:04d}".format(self.step_counter),
              "——— wall-clock time {:.1f}s,".format(time.time() - self.RESET_TIME),
              "simulation time {:.1f}s@{:d}Hz ({:.2f}x)".format(self.step_counter * self.PYB_TIMESTEP, self.PYB_FREQ,
                                                               (self.step_counter * self.PYB_TIMESTEP) / max(time.time() - self.RESET_TIME, 1e-9)))
        for i in range(self.NUM_DRONES):
            print("[INFO] BaseAviary.render() ——— drone {:d}".format(i),
                  "——— x {:+06.2f}, y {:+06.2f}, z {:+06.2f}".format(o[i, 0], o[i, 1], o[i, 2]),
                  "——— velocity {:+06.2f}, {:+06.2f}, {:+06.2f}".format(o[i, 10], o[i, 11], o[i, 12]),
                  "——— roll {:+06.2f}, pitch {:+06.2f}, yaw {:+06.2f}".format(*(o[i, 7:10] * 180 / np.pi)),
                  "——— angular velocity {:+06.4f}, {:+06.4f}, {:+06.4f} ——— ".format(o[i, 13], o[i, 14], o[i, 15]))

    def close(self):
        forget_env_device(id(self))
        if getattr(self, "_h", None) is not None:
            self._lib.mds_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def getPyBulletClient(self):
        return self.CLIENT

    def getDroneIds(self):
        return self.DRONE_IDS

    def _showDroneLocalAxes(self, nth_drone):
        pass  # Bullet GUI helper; nothing to draw here

    # ------------------------------------------------------------------ obs / state
    def _computeObs(self) -> torch.Tensor:
        self._require_open()
        capi.check(self._lib.mds_get_obs(self._h, C.c_void_p(self._obs.data_ptr()), self._stream()), "mds_get_obs")
        return self._obs

    def _getDroneStateVector(self, nth_drone, env: int = 0):
        return self._obs[env, nth_drone].detach().to("cpu", torch.float64).numpy()

    def get_state(self) -> np.ndarray:
        """World-frame 13-float state [E, D, 13] (pos3 | quat4 xyzw | vel3 | body rates3), float64 copy."""
        self._require_open()
        out = np.zeros((self.n, capi.STATE_DIM))
        capi.check(self._lib.mds_get_state(self._h, capi.as_double_ptr(out), self._stream()), "mds_get_state")
        return out.reshape(self.NUM_ENVS, self.NUM_DRONES, capi.STATE_DIM)

    def state_views(self):
        """Zero-copy torch views of the library-owned state (mds_state_ptrs): a dict with 13 strided ``[n]`` tensors
        ``comp[k]`` in the storage dtype (positions relative to ``origin``) and the three ``origin`` planes."""
        self._require_open()
        comp, stride, org = (C.c_void_p * 13)(), (C.c_size_t * 13)(), (C.c_void_p * 3)()
        capi.check(self._lib.mds_state_ptrs(self._h, comp, stride, org), "mds_state_ptrs")

        class _Raw:                      # the CUDA array interface is how torch adopts a foreign device pointer
            def __init__(self, ptr, n, stride, dt):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": dt, "data": (int(ptr), False), "version": 2,
                                                 "strides": (stride,)}

        ts = {torch.float16: "<f2", torch.float32: "<f4", torch.float64: "<f8"}
        es = {torch.float16: 2, torch.float32: 4, torch.float64: 8}[self.dtype]
        cdt = torch.float64 if self.dtype == torch.float64 else torch.float32
        return {"comp": [torch.as_tensor(_Raw(comp[k], self.n, int(stride[k]) * es, ts[self.dtype]), device=self.device) for k in range(13)],
                "origin": [torch.as_tensor(_Raw(org[k], self.n, es if cdt == self.dtype else 4, ts[cdt]), device=self.device) for k in range(3)]}

    def set_state(self, state):
        self._require_open()
        s = np.ascontiguousarray(np.asarray(state, dtype=np.float64).reshape(self.n, capi.STATE_DIM))
        capi.check(self._lib.mds_set_state(self._h, capi.as_double_ptr(s), self._stream()), "mds_set_state")

    def _kin(self, lo, hi):
        s = self.get_state()
        return s[0, :, lo:hi] if self.NUM_ENVS == 1 else s[:, :, lo:hi]

    pos = property(lambda self: self._kin(0, 3))
    quat = property(lambda self: self._kin(3, 7))
    vel = property(lambda self: self._kin(7, 10))
    rpy_rates = property(lambda self: self._kin(10, 13))

    @property
    def rpy(self):
        o = self._computeObs().detach().to("cpu", torch.float64).numpy()[..., 7:10]
        return o[0] if self.NUM_ENVS == 1 else o

    @property
    def ang_v(self):
        o = self._computeObs().detach().to("cpu", torch.float64).numpy()[..., 13:16]
        return o[0] if self.NUM_ENVS == 1 else o

    # ------------------------------------------------------------------ fused on-GPU control loop
    def set_trajectories(self, trajs):
        """Attach one trajectory per drone for ``step_geometric``: a list of D (shared by every env) or
        E*D trajectory objects, or an [E,D,7]/[D,7] Lemniscate parameter array (a, omega, centre3, yaw_rate,
        phase_shift).  All-Lemniscate sets use the fused fp32 kernel; anything else (Circle, Line, Wait,
        Compound, Rotate) is flattened into segment tables for the general kernel."""
        self._require_open()
        from ..trajectories.Lemniscate import Lemniscate
        from ..trajectories.base import upload_segments
        if isinstance(trajs, (list, tuple)):
            trajs = list(trajs)
            if len(trajs) == self.NUM_DRONES and self.NUM_ENVS > 1:
                trajs = trajs * self.NUM_ENVS
            if len(trajs) != self.n:
                raise ValueError(f"need {self.NUM_DRONES} or {self.n} trajectories, got {len(trajs)}")
            if not all(type(t) is Lemniscate for t in trajs):
                upload_segments(self._lib, self._h, trajs, self.device)
                self._has_traj = True
                return
            P = np.array([t.params() for t in trajs], dtype=np.float64)
        else:
            P = np.asarray(trajs, dtype=np.float64)
        if P.shape == (self.NUM_DRONES, capi.LEM_DIM):
            P = np.broadcast_to(P, (self.NUM_ENVS, self.NUM_DRONES, capi.LEM_DIM))
        P = np.ascontiguousarray(P.reshape(self.n, capi.LEM_DIM))
        capi.check(self._lib.mds_set_lemniscate(self._h, capi.as_double_ptr(P), self._stream()), "mds_set_lemniscate")
        self._has_traj = True

    def set_wind(self, force_world):
        """Constant world-frame force [N] on every drone each physics substep -- the reference's
        ``p.applyExternalForce(..., [wind_force, 0, 0], WORLD_FRAME)`` (EnvGeometric.py:463-467)."""
        f = np.ascontiguousarray(np.asarray(force_world, dtype=np.float64).reshape(3))
        capi.check(self._lib.mds_set_wind(self._h, capi.as_double_ptr(f)), "mds_set_wind")

    def set_geometric_gains(self, Kp=None, Kv=None, KR=None, Kw=None, g=None, max_tilt_angle=None):
        gains = capi.MdsGeometricGains()
        capi.check(self._lib.mds_default_geometric_gains(C.byref(gains)), "mds_default_geometric_gains")
        for name, val in (("Kp", Kp), ("Kv", Kv), ("KR", KR), ("Kw", Kw)):
            if val is not None:
                v = np.broadcast_to(np.asarray(val, dtype=np.float64), (3,))
                setattr(gains, name, (C.c_double * 3)(*v))
        if g is not None:
            gains.g = float(g)
        if max_tilt_angle is not None:
            gains.max_tilt_angle = float(max_tilt_angle)
        capi.check(self._lib.mds_set_geometric_gains(self._h, C.byref(gains)), "mds_set_geometric_gains")

    def step_geometric(self, t: float, return_action: bool = False):
        """One fused control step of simulations/EnvGeometric.py:434-469 for every drone:
        trajectory sample -> GeometricControl.compute -> input_to_action -> env.step."""
        self._require_open()
        act_ptr = C.c_void_p(self._act.data_ptr()) if return_action else C.c_void_p(None)
        capi.check(self._lib.mds_step_geometric(self._h, C.c_double(t), C.c_void_p(self._obs.data_ptr()), act_ptr,
                                                self._stream()), "mds_step_geometric")
        self.step_counter += self.PYB_STEPS_PER_CTRL
        return (self._obs, self._act) if return_action else self._obs

    def step_lqr(self, t: float, return_action: bool = False):
        """The same control step with the reference's 12-state LQRController (needs one constructed on this env): the default
        'lqr' branch of simulations/EnvGeometric.py do_control, fused for every drone."""
        self._require_open()
        act_ptr = C.c_void_p(self._act.data_ptr()) if return_action else C.c_void_p(None)
        capi.check(self._lib.mds_step_lqr(self._h, C.c_double(t), C.c_void_p(self._obs.data_ptr()), act_ptr, self._stream()), "mds_step_lqr")
        self.step_counter += self.PYB_STEPS_PER_CTRL
        return (self._obs, self._act) if return_action else self._obs

    def rollout_geometric(self, t0: float, n_steps: int, want_obs: bool = True, obs_every_step: bool = False):
        """``n_steps`` fused control steps enqueued from C; returns the last observation.
        ``obs_every_step`` materialises the [E,D,20] observation on every step (as env.step does)."""
        self._require_open()
        obs_ptr = C.c_void_p(self._obs.data_ptr()) if want_obs else C.c_void_p(None)
        capi.check(self._lib.mds_rollout_geometric(self._h, C.c_double(t0), C.c_int(n_steps), obs_ptr,
                                                   C.c_int(1 if obs_every_step else 0), self._stream()),
                   "mds_rollout_geometric")
        self.step_counter += n_steps * self.PYB_STEPS_PER_CTRL
        return self._obs if want_obs else None

    def rollout_step(self, actions: torch.Tensor, first_step: int, n_steps: int, obs_log: torch.Tensor | None = None,
                     episode_len: int = 0, steps_per_launch: int = 1):
        """``n_steps`` plain ``env.step`` calls issued from C with a replayed action table: step j applies ``actions[j % A]``
        ([A,E,D,4] device tensor of this env's dtype) and writes its observation into ``obs_log[j % T]`` ([T,E,D,20]).
        ``episode_len`` > 0 puts the drones back to their initial poses before every step j > 0 with j % episode_len == 0.
        ``steps_per_launch`` > 1 runs that many steps per kernel launch with the state in registers (mds_rollout_step_fused).
        Returns the log (or None)."""
        self._require_open()
        if actions.dtype != self.dtype or not actions.is_contiguous() or actions.numel() % (self.n * capi.ACT_DIM):
            raise ValueError("actions must be a contiguous [A,E,D,4] tensor of the env's dtype")
        if obs_log is not None and (obs_log.dtype != self.dtype or not obs_log.is_contiguous() or obs_log.numel() % (self.n * 20)):
            raise ValueError("obs_log must be a contiguous [T,E,D,20] tensor of the env's dtype")
        A = actions.numel() // (self.n * capi.ACT_DIM)
        T = obs_log.numel() // (self.n * 20) if obs_log is not None else 0
        if steps_per_launch > 1:
            capi.check(self._lib.mds_rollout_step_fused(self._h, C.c_void_p(actions.data_ptr()), C.c_int(A), C.c_int(first_step), C.c_int(n_steps),
                                                        C.c_void_p(obs_log.data_ptr() if obs_log is not None else None), C.c_int(T),
                                                        C.c_int(int(episode_len)), C.c_int(int(steps_per_launch)), self._stream()),
                       "mds_rollout_step_fused")
            self.step_counter += n_steps * self.PYB_STEPS_PER_CTRL
            return obs_log
        capi.check(self._lib.mds_rollout_step(self._h, C.c_void_p(actions.data_ptr()), C.c_int(A), C.c_int(first_step), C.c_int(n_steps),
                                              C.c_void_p(obs_log.data_ptr() if obs_log is not None else None), C.c_int(T), C.c_int(int(episode_len)),
                                              self._stream()),
                   "mds_rollout_step")
        self.step_counter += n_steps * self.PYB_STEPS_PER_CTRL
        return obs_log

    def set_rollout_streams(self, n_streams: int = 0):
        """How rollout_geometric issues its steps: 0 auto, 1 the current stream only, 2 the two halves of the shard on two
        internal streams (mds_set_rollout_streams).  Same results either way."""
        self._require_open()
        capi.check(self._lib.mds_set_rollout_streams(self._h, C.c_int(int(n_streams))), "mds_set_rollout_streams")

    def last_rollout_streams(self) -> int:
        """1 or 2: how the most recent rollout_* call of this env was issued (0: none yet)."""
        self._require_open()
        return int(self._lib.mds_get_last_rollout_streams(self._h))

    def rollout_streams_for(self, n_steps: int, cbf: bool = False) -> int:
        """1 or 2: how a rollout of ``n_steps`` would be issued under the current ``set_rollout_streams`` setting."""
        self._require_open()
        return int(self._lib.mds_rollout_streams_for(self._h, C.c_int(1 if cbf else 0), C.c_int(int(n_steps))))

    def set_rollout_form(self, form: int = 0, steps_per_launch: int = 0):
        """How rollout_geometric launches its steps (mds_set_rollout_form): 0 auto (form 2 from 8 192 drones and 8 steps on), 1 one launch per control step
        (bit-identical to step_geometric calls at every shard size), 2 the whole-rollout kernel in launches of
        ``steps_per_launch`` control steps (state in registers; results equal to rounding)."""
        self._require_open()
        capi.check(self._lib.mds_set_rollout_form(self._h, C.c_int(int(form)), C.c_int(int(steps_per_launch))), "mds_set_rollout_form")

    def rollout_form_for(self, n_steps: int) -> int:
        self._require_open()
        return int(self._lib.mds_rollout_form_for(self._h, C.c_int(int(n_steps))))

    def last_rollout_form(self) -> int:
        self._require_open()
        return int(self._lib.mds_get_last_rollout_form(self._h))

    def rollout_geometric_fused(self, t0: float, n_steps: int, log: bool = False, log_out: torch.Tensor | None = None,
                                controller: str = "geometric"):
        """``n_steps`` fused control steps in ONE kernel launch (state stays in registers).  With
        ``log`` every step's observation goes to a [T,E,D,20] tensor (the reference's
        ``observations.append(obs)``, EnvGeometric.py:471); returns (last obs, log or None).  ``controller="lqr"`` runs the
        12-state LQRController (constructed on this env) instead of GeometricControl."""
        self._require_open()
        if log and log_out is None:
            log_out = torch.empty((n_steps, self.NUM_ENVS, self.NUM_DRONES, capi.OBS_DIM), dtype=self.dtype, device=self.device)
        fn = {"lqr": self._lib.mds_rollout_lqr_fused, "geometric": self._lib.mds_rollout_geometric_fused,
              "nominal": self._lib.mds_rollout_nominal_fused}[controller]   # "nominal": the LQR + low level chosen with set_cbf_nominal
        capi.check(fn(self._h, C.c_double(t0), C.c_int(n_steps), C.c_void_p(log_out.data_ptr() if log_out is not None else None),
                      C.c_void_p(self._obs.data_ptr()), self._stream()), "mds_rollout_*_fused")
        self.step_counter += n_steps * self.PYB_STEPS_PER_CTRL
        return self._obs, log_out

    def set_dslpid_gains(self, ctrl):
        """Push the P/I/D_COEFF_FOR/TOR arrays of a DSLPIDControl object to this env (PIDEnv.py:124-134)."""
        from ..control.DSLPIDControl import gains_struct
        capi.check(self._lib.mds_set_dslpid_gains(self._h, C.byref(gains_struct(ctrl))), "mds_set_dslpid_gains")

    def step_dslpid(self, target_pos, target_rpy, return_action: bool = False):
        """MultiDroneEnv.sim_step (PIDEnv.py:161-176) for every drone: DSLPID towards target_pos / target_rpy
        ([E,D,3] or [D,3]), fused with env.step."""
        self._require_open()
        def prep(a):
            if not isinstance(a, torch.Tensor):
                a = np.asarray(a, dtype=np.float64)
                if a.shape == (self.NUM_DRONES, 3):
                    a = np.broadcast_to(a, (self.NUM_ENVS, self.NUM_DRONES, 3))
            return to_device(a, self.device, self.dtype).reshape(self.n, 3)
        tp, tr = prep(target_pos), prep(target_rpy)
        act_ptr = C.c_void_p(self._act.data_ptr()) if return_action else C.c_void_p(None)
        capi.check(self._lib.mds_step_dslpid(self._h, C.c_void_p(tp.data_ptr()), C.c_void_p(tr.data_ptr()), C.c_void_p(self._obs.data_ptr()),
                                             act_ptr, self._stream()), "mds_step_dslpid")
        self.step_counter += self.PYB_STEPS_PER_CTRL
        return (self._obs, self._act) if return_action else self._obs

    def rollout_dslpid(self, target_pos, target_rpy, n_steps: int, first_step: int = 0, obs_every_step: bool = False):
        """``for i in range(n_steps): env.sim_step()`` of PIDEnv.py:201-207 as one C call (``mds_rollout_dslpid``).  Targets:
        [D,3] / [E,D,3] (fixed, PIDEnv's TARGET_POSITIONS) or a waypoint table [W,E,D,3] used cyclically from ``first_step``.
        Returns the last observation."""
        self._require_open()
        def prep(a):
            if not isinstance(a, torch.Tensor):
                a = np.asarray(a, dtype=np.float64)
                if a.shape == (self.NUM_DRONES, 3):
                    a = np.broadcast_to(a, (self.NUM_ENVS, self.NUM_DRONES, 3))
            a = to_device(a, self.device, self.dtype)
            return a.reshape(-1, self.n, 3).contiguous()
        tp, tr = prep(target_pos), prep(target_rpy)
        if tp.shape != tr.shape:
            raise ValueError(f"target_pos {tuple(tp.shape)} and target_rpy {tuple(tr.shape)} must hold the same number of sets")
        capi.check(self._lib.mds_rollout_dslpid(self._h, C.c_void_p(tp.data_ptr()), C.c_void_p(tr.data_ptr()), C.c_int(tp.shape[0]),
                                                C.c_int(first_step), C.c_int(n_steps), C.c_void_p(self._obs.data_ptr()),
                                                C.c_int(1 if obs_every_step else 0), self._stream()), "mds_rollout_dslpid")
        self.step_counter += n_steps * self.PYB_STEPS_PER_CTRL
        return self._obs

    def set_cbf_nominal(self, which: str):
        """Nominal controller of ``step_cbf_geometric``: "geometric" (GeometricControl return_omegas),
        "lqr_omega" (LQROmegaController) or "lqr_yank_omega" (LQRYankOmegaController, the order-3 loop of
        simulations/CBFTestOrd3.py); the LQR ones need a controller constructed on this env first."""
        capi.check(self._lib.mds_cbf_set_nominal(self._h, {"geometric": 0, "lqr_omega": 1, "lqr_yank_omega": 2}[which]),
                   "mds_cbf_set_nominal")

    def set_cbf_step_kernel(self, one_launch):
        """How the CBF-filtered step is issued (mds_cbf_set_step_kernel): False / 0 (default) the QP launch + the low-level launch,
        True / 1 one launch per control step where it applies (the faster of the two when few envs need active-set iterations),
        2 or "persistent" one launch of the several-steps-per-launch kernel per step (the fastest per-step form where it applies;
        ``rollout_cbf_geometric`` then runs 25 steps per launch)."""
        mode = 2 if one_launch == "persistent" else int(one_launch)
        capi.check(self._lib.mds_cbf_set_step_kernel(self._h, C.c_int(mode)), "mds_cbf_set_step_kernel")

    def cbf_last_step_kernel(self) -> int:
        """2: the most recent CBF-filtered step ran in the several-steps-per-launch kernel, 1: as one launch, 0: as QP launch +
        low-level launch, -1: none yet."""
        return int(self._lib.mds_cbf_last_step_kernel(self._h))

    def step_cbf_geometric(self, t: float, tracker, x_obs=None, obs_r_list=None, return_action: bool = False):
        """One CBF-filtered control step of simulations/CBFTest.py:303-350 for every env: nominal
        (force - M G, w_des) -> ``tracker`` (DroneQPTracker) ECBF QP -> ThrustOmega low level -> env.step; with an
        order-3 ``tracker.cbf`` the loop of simulations/CBFTestOrd3.py:306-352 (yank-omega LQR nominal, YankOmega low
        level).  Uses and updates the env's current observation; returns (obs, status[E])."""
        self._require_open()
        tracker.cbf.configure(x_obs, obs_r_list)
        if getattr(self, "_cbf_status", None) is None:
            self._cbf_status = torch.zeros((self.NUM_ENVS,), dtype=torch.int32, device=self.device)
        act_ptr = C.c_void_p(self._act.data_ptr()) if return_action else C.c_void_p(None)
        capi.check(self._lib.mds_step_cbf_geometric(self._h, C.c_double(t), C.c_void_p(self._obs.data_ptr()),
                                                    C.c_void_p(self._cbf_status.data_ptr()), act_ptr, self._stream()),
                   "mds_step_cbf_geometric")
        self.step_counter += self.PYB_STEPS_PER_CTRL
        return (self._obs, self._cbf_status, self._act) if return_action else (self._obs, self._cbf_status)

    def rollout_cbf_geometric(self, t0: float, n_steps: int, tracker, x_obs=None, obs_r_list=None):
        """``n_steps`` of ``step_cbf_geometric`` enqueued from C (mds_rollout_cbf_geometric); returns (obs, status) of the last step.
        Large batches run as two env halves on two internal streams (``set_rollout_streams``)."""
        self._require_open()
        tracker.cbf.configure(x_obs, obs_r_list)
        if getattr(self, "_cbf_status", None) is None:
            self._cbf_status = torch.zeros((self.NUM_ENVS,), dtype=torch.int32, device=self.device)
        capi.check(self._lib.mds_rollout_cbf_geometric(self._h, C.c_double(t0), C.c_int(n_steps), C.c_void_p(self._obs.data_ptr()),
                                                       C.c_void_p(self._cbf_status.data_ptr()), self._stream()),
                   "mds_rollout_cbf_geometric")
        self.step_counter += n_steps * self.PYB_STEPS_PER_CTRL
        return self._obs, self._cbf_status

    def rollout_cbf_geometric_fused(self, t0: float, n_steps: int, tracker, x_obs=None, obs_r_list=None, steps_per_launch: int = 20,
                                    obs_log: torch.Tensor | None = None, first_slot: int = 0, status_log: torch.Tensor | None = None):
        """``n_steps`` of ``step_cbf_geometric`` with ``steps_per_launch`` control steps per launch (mds_rollout_cbf_geometric_fused: the
        persistent kernel -- state, nominal input and QP results stay on the chip between steps).  ``obs_log``: optional ring
        ``[slots, E, D, 20]``, step k writes slot ``(first_slot + k) % slots``; ``status_log``: optional int32 ``[n_steps, E]``.
        Returns (obs, status) of the last step."""
        self._require_open()
        tracker.cbf.configure(x_obs, obs_r_list)
        if getattr(self, "_cbf_status", None) is None:
            self._cbf_status = torch.zeros((self.NUM_ENVS,), dtype=torch.int32, device=self.device)
        slots = 0
        if obs_log is not None:
            if obs_log.dtype != self.dtype or obs_log.device != self.device or not obs_log.is_contiguous() or obs_log[0].numel() != self.n * capi.OBS_DIM:
                raise ValueError("obs_log must be a contiguous [slots, E, D, 20] tensor of the env's dtype on its device")
            slots = int(obs_log.shape[0])
        if status_log is not None and (status_log.dtype != torch.int32 or status_log.device != self.device or not status_log.is_contiguous()
                                       or status_log.numel() < n_steps * self.NUM_ENVS):
            raise ValueError("status_log must be a contiguous int32 [n_steps, E] tensor on the env's device")
        capi.check(self._lib.mds_rollout_cbf_geometric_fused(
            self._h, C.c_double(t0), C.c_int(n_steps), C.c_int(steps_per_launch),
            C.c_void_p(obs_log.data_ptr() if obs_log is not None else None), C.c_int(slots), C.c_int(first_slot),
            C.c_void_p(self._obs.data_ptr()), C.c_void_p(self._cbf_status.data_ptr()),
            C.c_void_p(status_log.data_ptr() if status_log is not None else None), self._stream()), "mds_rollout_cbf_geometric_fused")
        self.step_counter += n_steps * self.PYB_STEPS_PER_CTRL
        return self._obs, self._cbf_status

    def step_nominal(self, t: float, return_action: bool = False):
        """``ctrl[j].compute(obs[j])`` + ``env.step(action)`` of simulations/EnvGeometricOmega.py / EnvGeometricYankOmega.py for every
        drone: the LQR selected with ``set_cbf_nominal`` ("lqr_omega" | "lqr_yank_omega"), its low-level controller, the physics step.
        No safety filter.  Uses and updates the env's current observation."""
        self._require_open()
        act_ptr = C.c_void_p(self._act.data_ptr()) if return_action else C.c_void_p(None)
        capi.check(self._lib.mds_step_nominal(self._h, C.c_double(t), C.c_void_p(self._obs.data_ptr()), act_ptr, self._stream()),
                   "mds_step_nominal")
        self.step_counter += self.PYB_STEPS_PER_CTRL
        return (self._obs, self._act) if return_action else self._obs

    # ------------------------------------------------------------------ gym hooks (CtrlAviary fills them)
    def _computeReward(self):
        raise NotImplementedError

    def _computeTerminated(self):
        raise NotImplementedError

    def _computeTruncated(self):
        raise NotImplementedError

    def _computeInfo(self):
        raise NotImplementedError
