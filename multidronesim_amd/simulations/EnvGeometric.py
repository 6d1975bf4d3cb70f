"""simulations/EnvGeometric.py of the reference: ``parse_args``, ``GeometricEnv(args, circle_init)`` with
``create_env`` (:84-111), ``do_control(trajs, render, ...)`` (:404-481), ``circle_initialize`` (:502-524),
``geometric_xdot`` (:483-500).

``do_control`` is the reference's loop -- trajs[j](t) -> GeometricControl.compute -> wind -> env.step ->
observations.append(obs) -- run as fused kernels for every drone of every env.  Extra argument: ``args.num_envs``
(default 1 = the reference).  ``self.observations`` ends up as the reference leaves it: one [D,20] array per
control step (``np.save(path, geo.observations)`` -> [T,D,20], :553; with num_envs > 1: [T,E,D,20]).

Controllers: 'lqr' (LQRController on the 12-state LinearizedModel, the script's default, incl. ``use_noisy_model``) and
'geometric'.  Out of scope here: 'dlqr' and ``fedCE*`` (system identification)."""
from __future__ import annotations

import argparse
import time

import numpy as np
import torch

from ..envs.CtrlAviary import CtrlAviary, DroneModel, Physics
from ..trajectories import *  # noqa: F401,F403  (the reference star-imports them too)
from ..utils.utils import str2bool, sync

DEFAULT_DRONES = DroneModel("cf2p")
DEFAULT_PHYSICS = Physics("pyb")
DEFAULT_GUI = False                      # the reference defaults to True: there is no Bullet GUI here
DEFAULT_PLOT = False
DEFAULT_RECORD = False
DEFAULT_USER_DEBUG_GUI = False
DEFAULT_SIMULATION_FREQ_HZ = 100
DEFAULT_CONTROL_FREQ_HZ = 100
DEFAULT_DURATION_SEC = 30
DEFAULT_OUTPUT_FOLDER = 'results'
DEFAULT_NUM_DRONES = 2
controllers = ['lqr', 'geometric']       # whichever is first is the default (:32); 'dlqr' (FedCE) is not on this path
wind_force = .00025


def parse_args(argv=None, init_rad=1.0):
    parser = argparse.ArgumentParser(description='Geometric trajectory tracking, batched on the GPU')
    parser.add_argument('--drone', default=DEFAULT_DRONES, type=DroneModel, choices=DroneModel, metavar='')
    parser.add_argument('--num_drones', default=DEFAULT_NUM_DRONES, type=int, metavar='')
    parser.add_argument('--physics', default=DEFAULT_PHYSICS, type=Physics, choices=Physics, metavar='')
    parser.add_argument('--gui', default=DEFAULT_GUI, type=str2bool, metavar='')
    parser.add_argument('--record', default=DEFAULT_RECORD, type=str2bool, metavar='')
    parser.add_argument('--plot', default=DEFAULT_PLOT, type=str2bool, metavar='')
    parser.add_argument('--user_debug_gui', default=DEFAULT_USER_DEBUG_GUI, type=str2bool, metavar='')
    parser.add_argument('--simulation_freq_hz', default=DEFAULT_SIMULATION_FREQ_HZ, type=int, metavar='')
    parser.add_argument('--control_freq_hz', default=DEFAULT_CONTROL_FREQ_HZ, type=int, metavar='')
    parser.add_argument('--duration_sec', default=DEFAULT_DURATION_SEC, type=int, metavar='')
    parser.add_argument('--output_folder', default=DEFAULT_OUTPUT_FOLDER, type=str, metavar='')
    parser.add_argument('--init_rad', default=init_rad, type=float, metavar='')
    parser.add_argument('--controller', default=controllers[0], type=str, metavar='')
    parser.add_argument('--num_envs', default=1, type=int, help='independent copies of the scene (batch axis)', metavar='')
    parser.add_argument('--dtype', default='float32', type=str, metavar='')
    return parser.parse_args(argv)


class GeometricEnv:
    def __init__(self, args, circle_init=True):
        self.env = None
        self.obs = None
        self.conversion_mat = None
        self.observations = []
        self.args = args
        self.INIT_XYZS = np.zeros((args.num_drones, 3))
        self.INIT_RPYS = np.zeros((args.num_drones, 3))
        self.TARGET_POSITIONS = np.zeros((args.num_drones, 3))
        self.TARGET_RPYS = np.zeros((args.num_drones, 3))
        self.obs_ts = []
        self.linear_models = None
        self.wind_force = wind_force
        self._use_noisy_model = False
        self._step = None
        if circle_init:
            self.starting_target_offset = 1
            self.circle_initialize()

    def create_env(self, gui=True, record=False):
        args = self.args
        env = CtrlAviary(drone_model=args.drone, num_drones=args.num_drones, initial_xyzs=self.INIT_XYZS, initial_rpys=self.INIT_RPYS,
                         physics=args.physics, pyb_freq=args.simulation_freq_hz, ctrl_freq=args.control_freq_hz,
                         gui=args.gui and gui, record=args.record or record, user_debug_gui=args.user_debug_gui,
                         output_folder=args.output_folder, num_envs=getattr(args, "num_envs", 1), dtype=getattr(args, "dtype", "float32"))
        self.env = env
        self.linear_models = self._make_linear_models(env)
        r = env.KM / env.KF
        self.conversion_mat = np.array([[1.0, 1.0, 1.0, 1.0], [0.0, env.L, 0.0, -env.L], [-env.L, 0.0, env.L, 0.0], [-r, r, -r, r]])
        return env

    def _make_linear_models(self, env):
        from ..model import LinearizedModel
        return [LinearizedModel(env) for _ in range(self.args.num_drones)]            # :103

    # ------------------------------------------------------------------ the loop
    def _start(self, trajs):
        env, args = self.env, self.args
        env.getPyBulletClient()
        env.getDroneIds()
        env._showDroneLocalAxes(0)
        if args.controller == 'lqr':      # one LQRController per drone in the reference (:425-427): same model, same gain -> one upload
            from ..control import LQRController
            LQRController(env, self.linear_models[0], use_noisy_model=self._use_noisy_model)
            self._step = env.step_lqr
        elif args.controller == 'geometric':
            self._step = env.step_geometric
        else:
            raise NotImplementedError(f"controller {args.controller!r}: the dLQR / FedCE parts of this script are outside the hot path")
        if trajs is None:                 # set-point regulation towards TARGET_POSITIONS / TARGET_RPYS[:, 2] (:449-455)
            trajs = [WaitTrajectory(duration=float(args.duration_sec), position=self.TARGET_POSITIONS[j], yaw=self.TARGET_RPYS[j, 2])  # noqa: F405
                     for j in range(args.num_drones)]
        env.set_trajectories(list(trajs))
        shape = (env.NUM_ENVS, env.NUM_DRONES, 4)
        env.step(torch.zeros(shape, dtype=env.dtype, device=env.device))              # :431
        return int(args.duration_sec * env.CTRL_FREQ)

    def _log(self, obs, t):
        o = obs.double().cpu().numpy()
        self.obs = o[0] if self.env.NUM_ENVS == 1 else o
        self.observations.append(self.obs)
        self.obs_ts.append(t)

    def do_control(self, trajs=None, render=False, use_noisy_model=False, wind=True):
        env = self.env
        self._use_noisy_model = use_noisy_model
        steps = self._start(trajs)
        if wind:
            env.set_wind([self.wind_force, 0.0, 0.0])                                  # :463-467, every step, every drone
        START = time.time()
        t = 0.0
        args_controller = "lqr" if self.args.controller == "lqr" else "geometric"
        if render:                        # step by step, real time, like the reference with its GUI
            for i in range(steps):
                obs = self._step(t)
                self._log(obs, t)
                t += env.CTRL_TIMESTEP
                env.render()
                sync(i, START, env.CTRL_TIMESTEP)
        else:                             # the same loop on the device, observations logged there
            log = torch.empty((steps, env.NUM_ENVS, env.NUM_DRONES, 20), dtype=env.dtype, device=env.device)
            env.rollout_geometric_fused(0.0, steps, log=True, log_out=log, controller=args_controller)   # one launch: state in registers
            for i in range(steps):
                self.obs_ts.append(t)
                t += env.CTRL_TIMESTEP
            o = log.double().cpu().numpy()
            self.observations.extend(list(o[:, 0] if env.NUM_ENVS == 1 else o))
            self.obs = self.observations[-1]
        env.close()

    def geometric_xdot(self, obs):
        """[v_world, w_body, R^T [0,0,F/m], 0] from one observation (:483-500)."""
        from scipy.spatial.transform import Rotation
        obs = np.array(obs)
        rpm = np.clip(obs[16:20], 0, self.env.MAX_RPM)                 # action_to_input(env, action)[0] (model_conversions.py:69-83),
        a = np.zeros((3,))                                               # from the env's constants: usable after env.close() like the reference
        a[2] = self.env.KF * np.sum(rpm ** 2) / self.env.M
        R = Rotation.from_quat(obs[3:7]).as_matrix()
        x_dot = np.zeros((12,))
        x_dot[0:3] = obs[10:13]
        x_dot[3:6] = np.matmul(R.T, obs[13:16])
        x_dot[6:9] = R.T @ a
        return x_dot

    def circle_initialize(self):
        args = self.args
        self.INIT_XYZS = np.zeros((args.num_drones, 3))
        for i in range(1, args.num_drones):                                            # first drone stays at the origin
            self.INIT_XYZS[i, 0] = args.init_rad * np.sin(((i - 1) / args.num_drones) * 2 * np.pi)
            self.INIT_XYZS[i, 1] = args.init_rad * np.cos(((i - 1) / args.num_drones) * 2 * np.pi)
        for i in range(args.num_drones):
            self.INIT_RPYS[i, 2] = 0
            self.TARGET_POSITIONS[i, 0:2] = self.INIT_XYZS[i, 0:2]
            self.TARGET_POSITIONS[i, 2] = self.INIT_XYZS[i, 2] + self.starting_target_offset
            self.TARGET_RPYS[i] = [0, 0, np.pi / 2]


if __name__ == "__main__":
    ARGS = parse_args()
    geo = GeometricEnv(ARGS, circle_init=True)
    env = geo.create_env(gui=True)
    trajs = [Lemniscate(center=np.array([0, 0, .5]), omega=1.5, yaw_rate=0.0, phase_shift=(-np.pi / 4) * (num - 1))  # noqa: F405
             for num in range(ARGS.num_drones)]                                        # :540
    geo.do_control(trajs=trajs)
    np.save("wind_observations_lem.npy", geo.observations)                            # :553-556
    print("Wrote observations to wind_observations_lem.npy", np.asarray(geo.observations).shape)
