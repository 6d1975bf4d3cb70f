"""simulations/CBFTest.py of the reference: the order-2 CBF demo.  Same ``GeometricEnv`` as EnvGeometric.py with
LinearizedOmegaModel per drone (:107), ``--init_rad 0.2`` (:61) and a ``do_control`` that runs nominal controller ->
DroneQPTracker -> + M G -> low level -> env.step (:303-350) when a ``qpTracker`` is given."""
from __future__ import annotations

import time

import numpy as np
import torch

from ..cbf import DroneCBF, DroneQPTracker
from ..control import LQROmegaController, ThrustOmegaController
from ..model import LinearizedOmegaModel
from ..trajectories import *  # noqa: F401,F403
from ..utils.utils import sync
from . import EnvGeometric as _base

controllers = ['lqr', 'geometric']        # :32


def parse_args(argv=None):
    args = _base.parse_args(argv, init_rad=.2)
    if args.controller == _base.controllers[0] and (argv is None or '--controller' not in argv):
        args.controller = controllers[0]
    return args


class GeometricEnv(_base.GeometricEnv):
    ORDER, XDIM = 2, 9

    def _make_linear_models(self, env):
        return [LinearizedOmegaModel(env) for _ in range(self.args.num_drones)]

    def _nominal(self, env):
        """The nominal controller objects the script builds (:283-293) and the name the fused step knows them by."""
        if self.args.controller == 'geometric':
            return 'geometric'
        if self.args.controller == 'lqr':
            LQROmegaController(env, self.linear_models[0], ThrustOmegaController(env))
            return 'lqr_omega'
        raise NotImplementedError(f"controller {self.args.controller!r}: dLQR / FedCE is outside the hot path")

    def do_control(self, trajs=None, render=False, qpTracker=None, computed_K=None, use_noisy_model=False, x_obs_list=None,
                   obs_r_list=None):
        env = self.env
        if self.args.controller == 'geometric' and qpTracker is None:        # plain geometric tracking: the EnvGeometric loop, no wind
            return super().do_control(trajs=trajs, render=render, wind=False)
        nominal = self._nominal(env)
        saved, self.args.controller = self.args.controller, 'geometric'      # _start only needs the trajectories and the first step
        try:
            steps = self._start(trajs)
        finally:
            self.args.controller = saved
        env.set_cbf_nominal(nominal)
        START = time.time()
        t = 0.0
        log = torch.empty((steps, env.NUM_ENVS, env.NUM_DRONES, 20), dtype=env.dtype, device=env.device)
        st_log = torch.zeros((steps, env.NUM_ENVS), dtype=torch.int32, device=env.device)
        if qpTracker is None and not render:   # ctrl[j].compute(obs[j]) = LQR + low level, then step (:296-300, :350): the whole run in one launch
            env.rollout_geometric_fused(0.0, steps, log=True, log_out=log, controller="nominal")
            for i in range(steps):
                self.obs_ts.append(t)
                t += env.CTRL_TIMESTEP
            steps = 0
        elif qpTracker is not None and not render:
            # nominal -> QP -> low level -> step (:303-350), the whole run through the persistent kernel where the library covers the
            # configuration (order 2, up to 16 drones per env, Lemniscates, Euler DYN at pyb == ctrl): 50 control steps per launch,
            # every step's observation into the log, every step's statuses into st_log.  MDS_EUNSUPPORTED: the step loop below.
            from .._capi import MdsError
            try:
                env.rollout_cbf_geometric_fused(0.0, steps, qpTracker, x_obs_list, obs_r_list, steps_per_launch=50, obs_log=log, status_log=st_log)
                for i in range(steps):
                    self.obs_ts.append(t)
                    t += env.CTRL_TIMESTEP
                steps = 0
            except MdsError as exc:
                if exc.status not in (-6, -5):          # unsupported combination / trajectories that are not Lemniscates
                    raise
        for i in range(steps):
            if qpTracker is not None:     # nominal -> QP -> low level -> step (:303-350)
                obs, st = env.step_cbf_geometric(t, qpTracker, x_obs_list, obs_r_list)
                st_log[i].copy_(st)
            else:
                obs = env.step_nominal(t)
            log[i].copy_(obs)
            self.obs_ts.append(t)
            t += env.CTRL_TIMESTEP
            if render:
                env.render()
                sync(i, START, env.CTRL_TIMESTEP)
        o = log.double().cpu().numpy()
        self.observations.extend(list(o[:, 0] if env.NUM_ENVS == 1 else o))
        self.obs = self.observations[-1]
        self.last_cbf_kernel = env.cbf_last_step_kernel()    # 2: the persistent rollout kernel, 1: one launch per step, 0: QP + low-level launches
        self.statuses = st_log.cpu().numpy()          # 1 where the QP was infeasible and the nominal control was kept (modelled fallback: cbf/qptracker.py docstring)
        env.close()


def add_env_obstacles(env, x_obs_list, obs_r_list):
    """The reference loads a sphere URDF per obstacle into Bullet for display (:407-412); nothing to draw here."""
    return None


if __name__ == "__main__":
    ARGS = parse_args()
    geo = GeometricEnv(ARGS, circle_init=True)
    env = geo.create_env()
    trajs = [Lemniscate(center=np.array([0, 0, 0.5]), omega=0.5, yaw_rate=0) for _ in range(ARGS.num_drones)]      # noqa: F405  (:418)
    droneCBF = DroneCBF(env, geo.linear_models, safety_radius=0.1, zscale=1)
    droneTracker = DroneQPTracker(droneCBF, num_robots=ARGS.num_drones)
    x_obs_list = np.array([np.array([[0, 0, .5], np.zeros(3)])])
    obs_r_list = [.1]
    add_env_obstacles(env, x_obs_list, obs_r_list)
    geo.do_control(trajs=trajs, qpTracker=droneTracker, render=False, x_obs_list=x_obs_list, obs_r_list=obs_r_list)
    print("final positions (env 0):\n", np.asarray(geo.observations[-1]).reshape(-1, ARGS.num_drones, 20)[0, :, :3],
          "\nQP fallbacks:", int(geo.statuses.sum()), "of", geo.statuses.size)
