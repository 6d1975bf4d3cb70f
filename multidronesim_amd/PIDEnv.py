"""PIDEnv.py of the reference: ``MultiDroneEnv`` -- a threaded hover service around
CtrlAviary with mutable ``TARGET_POSITIONS`` (PIDEnv.py:31-187).

Same constructor, ``threaded_sim() / run_sim() / sim_step(i) / stop()`` and attributes.  Like the
reference it builds one ``DSLPIDControl`` per drone with every gain halved (PIDEnv.py:124-134);
``sim_step`` runs that controller towards ``TARGET_POSITIONS`` / ``TARGET_RPYS`` fused with the
physics step on the GPU (``env.step_dslpid``).  DSLPIDControl is [UPSTREAM] (not in the reference
tree): spec-level restatement, parity unpinned.  ``controller="geometric"`` selects the reference's
own GeometricControl regulating to the targets instead.  The reference's latent bugs (module-global
ARGS, 20-vector unpacked into 4 names, missing ``logging`` import) are not reproduced."""
from __future__ import annotations

import argparse
import logging
import threading
import time

import numpy as np

from .envs.CtrlAviary import CtrlAviary
from .utils.enums import DroneModel, Physics
from .utils.utils import sync

DEFAULT_DRONES = DroneModel("cf2p")
DEFAULT_PHYSICS = Physics("pyb")
DEFAULT_GUI = True
DEFAULT_PLOT = False
DEFAULT_RECORD = False
DEFAULT_USER_DEBUG_GUI = False
DEFAULT_SIMULATION_FREQ_HZ = 100
DEFAULT_CONTROL_FREQ_HZ = 100
DEFAULT_DURATION_SEC = None
DEFAULT_OUTPUT_FOLDER = 'results'
DEFAULT_NUM_DRONES = 2


class MultiDroneEnv(object):
    def build_args(self, kwargs):
        """Keyword arguments -> the argparse-style namespace the rest of the class reads (unknown keys are reported and
        dropped, as the reference does).  ``num_envs`` / ``realtime`` / ``controller`` are additions of this package."""
        defaults = dict(drone=DEFAULT_DRONES, num_drones=DEFAULT_NUM_DRONES, physics=DEFAULT_PHYSICS, gui=DEFAULT_GUI, plot=DEFAULT_PLOT,
                        user_debug_gui=DEFAULT_USER_DEBUG_GUI, simulation_freq_hz=DEFAULT_SIMULATION_FREQ_HZ,
                        control_freq_hz=DEFAULT_CONTROL_FREQ_HZ, duration_sec=DEFAULT_DURATION_SEC, output_folder=DEFAULT_OUTPUT_FOLDER,
                        init_rad=1.0,
                        num_envs=1,            # batch axis
                        realtime=True,         # sync() to the wall clock like the reference (PIDEnv.py:179)
                        controller='dslpid')   # 'dslpid' (PIDEnv.py:124-134) or 'geometric'
        unknown = sorted(set(kwargs) - set(defaults))
        if unknown:
            logging.warning("MultiDroneEnv: ignoring unknown argument(s) %s; known: %s", unknown, sorted(defaults))
        defaults.update({k: v for k, v in kwargs.items() if k in defaults})
        return argparse.Namespace(**defaults)

    def __init__(self, INIT_XYZS=None, INIT_RPYS=None, TARGET_POSITIONS=None, TARGET_RPYS=None, args=None, **kwargs):
        self.args = self.build_args(kwargs) if args is None else args
        a = self.args
        for k, v in (("num_envs", 1), ("realtime", True), ("init_rad", 1.0), ("duration_sec", None), ("controller", "dslpid")):
            if not hasattr(a, k):
                setattr(a, k, v)
        starting_target_offset = 1
        if INIT_XYZS is None:
            INIT_XYZS = np.zeros((a.num_drones, 3))
            for i in range(1, a.num_drones):
                INIT_XYZS[i, 0] = a.init_rad * np.cos((i / a.num_drones) * 2 * np.pi)
                INIT_XYZS[i, 1] = a.init_rad * np.sin((i / a.num_drones) * 2 * np.pi)
                INIT_XYZS[i, 2] = 0.0
        if INIT_RPYS is None:
            INIT_RPYS = np.zeros((a.num_drones, 3))
        if TARGET_POSITIONS is None:
            TARGET_POSITIONS = np.array(INIT_XYZS, dtype=np.float64).copy()
            TARGET_POSITIONS[..., 2] += starting_target_offset
        if TARGET_RPYS is None:
            TARGET_RPYS = np.zeros((a.num_drones, 3))
        self.INIT_XYZS = INIT_XYZS
        self.INIT_RPYS = INIT_RPYS
        self.TARGET_POSITIONS = TARGET_POSITIONS
        self.TARGET_RPYS = TARGET_RPYS
        self.ctrl = []
        self.env = None
        self.action = None
        self.obs = None
        self.stop_cmd = False
        self._sent_targets = None

    def threaded_sim(self):
        self.stop_cmd = False
        t = threading.Thread(target=self.run_sim)
        t.start()
        return t

    def _push_targets(self):
        """Targets are copied to the device at a step boundary (the reference's REPL thread
        writes TARGET_POSITIONS unsynchronised, PIDEnv.py:206; here a change takes effect at the
        next step)."""
        tp = np.array(self.TARGET_POSITIONS, dtype=np.float64)
        if self._sent_targets is not None and np.array_equal(tp, self._sent_targets):
            return
        env = self.env
        P = np.zeros((env.NUM_ENVS, env.NUM_DRONES, 7))
        P[..., 1] = 1.0
        P[..., 2:5] = tp
        env.set_trajectories(P)
        self._sent_targets = tp.copy()

    def run_sim(self):
        args = self.args
        env = CtrlAviary(drone_model=args.drone, num_drones=args.num_drones, initial_xyzs=self.INIT_XYZS,
                         initial_rpys=self.INIT_RPYS, physics=args.physics, pyb_freq=args.simulation_freq_hz,
                         ctrl_freq=args.control_freq_hz, gui=args.gui, user_debug_gui=args.user_debug_gui,
                         output_folder=args.output_folder, num_envs=args.num_envs)
        self.env = env
        self.PYB_CLIENT = self.env.getPyBulletClient()
        self.DRONE_IDS = self.env.getDroneIds()
        self.env._showDroneLocalAxes(0)
        # PID control for set point regulation (PIDEnv.py:124-134): every gain halved
        ctrl = []
        if args.drone in [DroneModel.CF2X, DroneModel.CF2P]:
            from .control.DSLPIDControl import DSLPIDControl
            for i in range(args.num_drones):
                ctrl.append(DSLPIDControl(drone_model=args.drone))
                ctrl[i].P_COEFF_FOR = 0.5 * np.array([.4, .4, 1.25])
                ctrl[i].I_COEFF_FOR = 0.5 * np.array([.05, .05, .05])
                ctrl[i].D_COEFF_FOR = 0.5 * np.array([.2, .2, .5])
                ctrl[i].P_COEFF_TOR = 0.5 * np.array([70000., 70000., 60000.])
                ctrl[i].I_COEFF_TOR = 0.5 * np.array([.0, .0, 500.])
                ctrl[i].D_COEFF_TOR = 0.5 * np.array([20000., 20000., 12000.])
        self.ctrl = ctrl
        if ctrl:
            env.set_dslpid_gains(ctrl[0])      # the reference gives every drone the same gains
        self.START = time.time()
        self.action = np.zeros((args.num_drones, 4)) if args.num_envs == 1 else np.zeros((args.num_envs, args.num_drones, 4))
        self.obs, _, _, _, _ = self.env.step(self.action)
        if args.duration_sec is not None:
            CTRL_STEPS = int(args.duration_sec * self.env.CTRL_FREQ)
            for i in range(CTRL_STEPS):
                self.sim_step(i)
        else:
            i = 0
            while not self.stop_sim(i):
                self.sim_step(i)
                i += 1
        self.env.close()

    def sim_step(self, i):
        if getattr(self.args, "controller", "dslpid") == "dslpid":
            # targets are read at the step boundary (the reference's REPL thread writes them unsynchronised, PIDEnv.py:206)
            obs, act = self.env.step_dslpid(np.array(self.TARGET_POSITIONS, dtype=np.float64), np.array(self.TARGET_RPYS, dtype=np.float64),
                                            return_action=True)
        else:
            self._push_targets()
            obs, act = self.env.step_geometric(i * self.env.CTRL_TIMESTEP, return_action=True)
        self.obs = obs
        self.action = act
        if self.args.realtime:
            sync(i, self.START, self.env.CTRL_TIMESTEP)

    def stop_sim(self, i):
        if i * self.env.CTRL_TIMESTEP > 1000:
            return True
        return self.stop_cmd

    def stop(self):
        self.stop_cmd = True
