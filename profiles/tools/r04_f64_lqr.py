"""us per control step of the persistent CBF kernel with the LQR-omega nominal (simulations/CBFTest.py's default), float64 and fp32, 16 384 envs x 16
drones, `under` spheres, steps 20..220: python3 profiles/tools/r04_f64_lqr.py   (MDS_LIB_PATH selects the library build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
from multidronesim_amd.cbf.cbf import DroneCBF
from multidronesim_amd.cbf.qptracker import DroneQPTracker
from multidronesim_amd.model.linear_omega import LinearizedOmegaModel
from multidronesim_amd.control.lqr.lqr_omega_controller import LQROmegaController

E, D = 16384, 16
for dtype in ("float64", "float32"):
    xyz, rpy, P = bench.c4_inputs(E, D, 1000)
    env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=Physics.DYN, pyb_freq=100, ctrl_freq=100,
                     num_envs=E, dtype=dtype, device=0)
    env.set_trajectories(P)
    LQROmegaController(env, LinearizedOmegaModel(env), None)
    env.set_cbf_nominal("lqr_omega")
    cbf = DroneCBF(env, [LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2, cbf_poles=np.array([-2.2, -2.4]))
    trk = DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
    x_obs, obs_r = bench.c4_spheres("under")
    env.step(torch.zeros((E, D, 4), dtype=env.dtype, device=env.device))
    ring = torch.empty((50, E, D, 20), dtype=env.dtype, device=env.device)
    env.rollout_cbf_geometric_fused(0.0, 20, trk, x_obs, obs_r, steps_per_launch=50, obs_log=ring)
    us = bench._timed_steps(env.device, lambda: env.rollout_cbf_geometric_fused(0.2, 200, trk, x_obs, obs_r, steps_per_launch=50, obs_log=ring), 200)
    print(dtype, "lqr_omega nominal: %.2f us per control step" % us, "fallback last", float((env._cbf_status != 0).float().mean()), flush=True)
    env.close()
