"""Per-step times of the order-3 step-by-step loop (fp32 / float64, 4 096 envs x 8): is the fp32 average (201 us) a tail?  python3 profiles/tools/r04_o3_step_times.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
from multidronesim_amd.cbf.cbf import DroneCBF
from multidronesim_amd.cbf.qptracker import DroneQPTracker
from multidronesim_amd.model.linear_yank_omega import LinearizedYankOmegaModel
from multidronesim_amd.control import LQRYankOmegaController, YankOmegaController
E, D = 4096, 8
for dtype in ("float32", "float64"):
    xyz, rpy, P = bench.c4_inputs(E, D, 1000)
    P[..., 1] = 0.5
    env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=Physics.DYN, pyb_freq=100, ctrl_freq=100, num_envs=E, dtype=dtype, device=0)
    env.set_trajectories(P)
    LQRYankOmegaController(env, LinearizedYankOmegaModel(env), YankOmegaController(env))
    cbf = DroneCBF(env, [LinearizedYankOmegaModel(env) for _ in range(D)], safety_radius=0.125, zscale=2.0, order=3, cbf_poles=np.array([-3.0, -3.6, -5.6]))
    trk = DroneQPTracker(cbf, order=3, num_robots=D, xdim=10, env=env)
    env.set_cbf_nominal("lqr_yank_omega")
    x_obs, obs_r = [np.array([[0.0, 0.0, -3.0], [0, 0, 0], [0, 0, 0]])], [0.1]
    env.step(torch.full((E, D, 4), float(env.HOVER_RPM), dtype=env.dtype, device=env.device))
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(221)]
    t = 0.0
    its = []
    for k in range(220):
        evs[k].record()
        env.step_cbf_geometric(t, trk, x_obs, obs_r)
        its.append(cbf.last_iterations().max())
        t += env.CTRL_TIMESTEP
    evs[220].record()
    torch.cuda.synchronize()
    us = np.array([evs[k].elapsed_time(evs[k + 1]) * 1e3 for k in range(220)])
    mx = torch.stack(its).cpu().numpy()
    print(dtype, "us per step: median %.1f mean %.1f max %.1f at step %d; slowest five %s; max iterations per step: max %d at step %d" % (
        np.median(us), us.mean(), us.max(), us.argmax(), np.round(np.sort(us)[-5:], 1), mx.max(), mx.argmax()), flush=True)
    env.close()
