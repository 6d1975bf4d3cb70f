"""simulations/EnvGeometricOmega.py of the reference: trajectory tracking with the thrust / body-rate input model --
``GeometricEnv(args, circle_init)`` with a LinearizedOmegaModel per drone (:99), ``do_control(trajs, render, computed_K,
use_noisy_model)`` (:262-335) whose 'lqr' branch is ``LQROmegaController(env, model, ThrustOmegaController(env)).compute(obs[j])``
(nominal input + PID low level, :286-289, :314) followed by ``env.step(action)`` (:327), no wind, ``--init_rad 0.2`` and 50 s by
default (:28, :57).  The loop runs fused for every drone of every env (``mds_rollout_nominal_fused``: the whole run in one launch,
PID memory in registers; ``render=True``: ``mds_step_nominal`` step by step in real time).  Out of scope here: the
'dlqr' controller and ``fedCE*`` / ``warm_up_only`` (system identification around the hot path)."""
from __future__ import annotations

import numpy as np

from ..trajectories import *  # noqa: F401,F403
from . import CBFTest as _cbf
from . import EnvGeometric as _base

DEFAULT_DURATION_SEC = 50                 # :28
controllers = ['lqr', 'geometric']        # :32 without 'dlqr'


def parse_args(argv=None):
    args = _cbf.parse_args(argv)          # init_rad 0.2 (:57)
    if argv is None or '--duration_sec' not in argv:
        args.duration_sec = DEFAULT_DURATION_SEC
    return args


class GeometricEnv(_cbf.GeometricEnv):
    def do_control(self, trajs=None, render=False, computed_K=None, use_noisy_model=True):          # the reference's default (:265)
        if computed_K is not None or self.args.controller == 'dlqr':
            raise NotImplementedError("controller 'dlqr' (a gain identified by fedCE): the FedCE / decentralised-LQR loop is outside the hot path")
        # LQROmegaController(..., use_noisy_model=True) designs its gain on (Ahat, Bhat) (control/lqr/lqr_omega_controller.py:31-36)
        self._noisy = bool(use_noisy_model)
        return super().do_control(trajs=trajs, render=render, qpTracker=None)

    def _nominal(self, env):
        if self.args.controller == 'lqr' and getattr(self, "_noisy", False):
            from ..control import LQROmegaController, ThrustOmegaController
            LQROmegaController(env, self.linear_models[0], ThrustOmegaController(env), use_noisy_model=True)
            return 'lqr_omega'
        return super()._nominal(env)

    def circle_initialize(self):
        """:356-381: drone i > 0 at angle 2 pi i / N (EnvGeometric.py uses (i - 1) / N), targets one metre above, target yaw pi / 2."""
        args = self.args
        self.INIT_XYZS = np.zeros((args.num_drones, 3))
        for i in range(1, args.num_drones):
            self.INIT_XYZS[i, 0] = args.init_rad * np.sin((i / args.num_drones) * 2 * np.pi)
            self.INIT_XYZS[i, 1] = args.init_rad * np.cos((i / args.num_drones) * 2 * np.pi)
        for i in range(args.num_drones):
            self.INIT_RPYS[i, 2] = 0
            self.TARGET_POSITIONS[i, 0:2] = self.INIT_XYZS[i, 0:2]
            self.TARGET_POSITIONS[i, 2] = self.INIT_XYZS[i, 2] + self.starting_target_offset
            self.TARGET_RPYS[i] = [0, 0, np.pi / 2]


if __name__ == "__main__":
    ARGS = parse_args()
    geo = GeometricEnv(ARGS, circle_init=True)
    env = geo.create_env()
    delta = np.array([0, 5, 0])
    trajs = [CompoundTrajectory([LineTrajectory(start=geo.INIT_XYZS[idx], end=geo.TARGET_POSITIONS[idx], speed=.5),         # noqa: F405  (:396-403)
                                 WaitTrajectory(duration=1, position=geo.TARGET_POSITIONS[idx]),                             # noqa: F405
                                 LineTrajectory(start=geo.TARGET_POSITIONS[idx], end=geo.TARGET_POSITIONS[idx] + delta, speed=1),   # noqa: F405
                                 LineTrajectory(start=geo.TARGET_POSITIONS[idx] + delta, end=geo.TARGET_POSITIONS[idx], speed=1)])   # noqa: F405
             for idx in range(ARGS.num_drones)]
    geo.do_control(trajs=trajs, render=False, use_noisy_model=False)           # (the reference runs fedCE first and then 'dlqr': not on this path)
    np.save("observations_lem_bad_mass.npy", geo.observations)                 # :407
    print("Wrote observations to observations_lem_bad_mass.npy", np.asarray(geo.observations).shape)
