"""simulations/CBFTestOrd3.py of the reference: the order-3 (yank-omega) CBF demo.  ``GeometricEnv(args, init_type,
lemniscate_a)`` with LinearizedYankOmegaModel per drone (:109), LQRYankOmegaController + YankOmegaController as the
nominal / low level (:294-297) and the loop of :306-352 (yank - M G kept as the reference has it)."""
from __future__ import annotations

import numpy as np

from ..cbf import DroneCBF, DroneQPTracker
from ..control import LQRYankOmegaController, YankOmegaController
from ..model import LinearizedYankOmegaModel
from ..trajectories import *  # noqa: F401,F403
from . import CBFTest as _cbf

parse_args = _cbf.parse_args


class GeometricEnv(_cbf.GeometricEnv):
    ORDER, XDIM = 3, 10

    def __init__(self, args, init_type='circle', lemniscate_a=1, center=np.array([0, 0, 0])):
        super().__init__(args, circle_init=False)
        self.init_type, self.lemniscate_a, self.center = init_type, lemniscate_a, np.asarray(center, dtype=np.float64)
        self.starting_target_offset = 1
        if init_type == 'circle':
            self.circle_initialize()
        elif init_type == 'lemniscate':
            self.lemniscate_initialize()

    def lemniscate_initialize(self):
        """Every drone starts on its own point of the shared lemniscate (:388-416)."""
        D = self.args.num_drones
        self.INIT_XYZS = np.zeros((D, 3))
        for i in range(D):
            self.INIT_XYZS[i] = Lemniscate(center=self.center, phase_shift=(2 * np.pi / (D + 0.25)) * i)(0)[0]    # noqa: F405
        self._targets_above_start(0.0)

    def circle_initialize(self):
        """:418-436: cos/sin order and a centre, unlike EnvGeometric.py's."""
        args = self.args
        self.INIT_XYZS = np.zeros((args.num_drones, 3))
        for i in range(1, args.num_drones):
            self.INIT_XYZS[i, 0] = args.init_rad * np.cos((i / args.num_drones) * 2 * np.pi) + self.center[0]
            self.INIT_XYZS[i, 1] = args.init_rad * np.sin((i / args.num_drones) * 2 * np.pi) + self.center[1]
            self.INIT_XYZS[i, 2] = self.center[2]
        self._targets_above_start(0.0)

    def _targets_above_start(self, yaw):
        self.INIT_RPYS[:, 2] = 0
        self.TARGET_POSITIONS = self.INIT_XYZS + np.array([0, 0, self.starting_target_offset])
        self.TARGET_RPYS = np.zeros((self.args.num_drones, 3))
        self.TARGET_RPYS[:, 2] = yaw

    def _make_linear_models(self, env):
        return [LinearizedYankOmegaModel(env) for _ in range(self.args.num_drones)]

    def _nominal(self, env):
        if self.args.controller != 'lqr':
            raise NotImplementedError("the order-3 filter acts on (yank, w): only the yank-omega LQR produces that input (CBFTestOrd3.py:294-297)")
        LQRYankOmegaController(env, self.linear_models[0], YankOmegaController(env))
        return 'lqr_yank_omega'

    def _start(self, trajs):
        steps = super()._start(trajs)
        env = self.env
        import torch
        # the loop integrates thrust from the RPM echo of the obs; Bullet's ground holds the drones until the motors spin up,
        # the explicit DYN model has no ground: start from hover RPM instead of the reference's all-zero first action
        env.reset()
        env.step(torch.full((env.NUM_ENVS, env.NUM_DRONES, 4), float(env.HOVER_RPM), dtype=env.dtype, device=env.device))
        return steps


if __name__ == "__main__":
    ARGS = parse_args()
    geo = GeometricEnv(ARGS, init_type='lemniscate', lemniscate_a=1)
    env = geo.create_env()
    trajs = [Lemniscate(a=1, center=np.array([0, 0, 0.5]), omega=0.5, yaw_rate=0,                          # noqa: F405  (:450)
                        phase_shift=(2 * np.pi / (ARGS.num_drones + 0.25)) * num) for num in range(ARGS.num_drones)]
    droneCBF = DroneCBF(env, geo.linear_models, safety_radius=0.125, zscale=2, order=3, cbf_poles=np.array([-3.0, -3.6, -5.6]))
    droneTracker = DroneQPTracker(droneCBF, num_robots=ARGS.num_drones, xdim=10, env=env, order=3)
    geo.do_control(trajs=trajs, qpTracker=droneTracker, render=False, x_obs_list=None, obs_r_list=None)
    print("final positions (env 0):\n", np.asarray(geo.observations[-1]).reshape(-1, ARGS.num_drones, 20)[0, :, :3],
          "\nQP fallbacks:", int(geo.statuses.sum()), "of", geo.statuses.size)
