"""simulations/EnvGeometricYankOmega.py of the reference: trajectory tracking with the yank / body-rate input model --
``GeometricEnv(args, circle_init)`` with a LinearizedYankOmegaModel per drone (:99), ``do_control`` (:264-360) whose 'lqr' branch
is ``LQRYankOmegaController(env, model, YankOmegaController(env)).compute(obs[j])`` (:288-289, :319) followed by ``env.step``
(:332); one drone and 5 s by default (:28-30).  Fused like EnvGeometricOmega's (``mds_rollout_nominal_fused`` with the yank-omega
nominal: the thrust state is calc_z_thrust of the observation's RPM echo, the low level integrates the yank).  Out of scope here:
'dlqr' / ``fedCE*`` and the DataLogger figures (:346-359)."""
from __future__ import annotations

import numpy as np

from ..trajectories import *  # noqa: F401,F403
from . import CBFTestOrd3 as _ord3

DEFAULT_DURATION_SEC = 5                  # :28
DEFAULT_NUM_DRONES = 1                    # :30
controllers = ['lqr']


def parse_args(argv=None):
    args = _ord3.parse_args(argv)         # init_rad 0.2 (:57)
    if argv is None or '--duration_sec' not in argv:
        args.duration_sec = DEFAULT_DURATION_SEC
    if argv is None or '--num_drones' not in argv:
        args.num_drones = DEFAULT_NUM_DRONES
    return args


class GeometricEnv(_ord3.GeometricEnv):
    def __init__(self, args, circle_init=True):
        super().__init__(args, init_type='circle' if circle_init else None)

    def do_control(self, trajs=None, render=False, computed_K=None, use_noisy_model=True):      # the reference's default (:265)
        if computed_K is not None or self.args.controller == 'dlqr':
            raise NotImplementedError("controller 'dlqr' (a gain identified by fedCE): the FedCE / decentralised-LQR loop is outside the hot path")
        self._noisy = bool(use_noisy_model)
        return super().do_control(trajs=trajs, render=render, qpTracker=None)

    def _nominal(self, env):
        if getattr(self, "_noisy", False):
            from ..control import LQRYankOmegaController, YankOmegaController
            if self.args.controller != 'lqr':
                raise NotImplementedError("only the yank-omega LQR produces the (yank, w) input of this loop")
            LQRYankOmegaController(env, self.linear_models[0], YankOmegaController(env), use_noisy_model=True)
            return 'lqr_yank_omega'
        return super()._nominal(env)

    def circle_initialize(self):
        """:381-406: drone i > 0 at angle 2 pi i / N, cos on x and sin on y; targets one metre above, target yaw pi / 2."""
        args = self.args
        self.INIT_XYZS = np.zeros((args.num_drones, 3))
        for i in range(1, args.num_drones):
            self.INIT_XYZS[i, 0] = args.init_rad * np.cos((i / args.num_drones) * 2 * np.pi)
            self.INIT_XYZS[i, 1] = args.init_rad * np.sin((i / args.num_drones) * 2 * np.pi)
        self._targets_above_start(np.pi / 2)


if __name__ == "__main__":
    ARGS = parse_args()
    geo = GeometricEnv(ARGS, circle_init=True)
    env = geo.create_env()
    trajs = [Lemniscate(center=np.array([0, 0, .5]), omega=1.5, yaw_rate=0) for _ in range(ARGS.num_drones)]      # noqa: F405  (:413)
    geo.do_control(trajs=trajs, render=False, use_noisy_model=False)                                               # :432
    np.save("observations_omega.npy", geo.observations)                                                            # :415
    print("Wrote observations to observations_omega.npy", np.asarray(geo.observations).shape)
