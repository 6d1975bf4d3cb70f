import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import helpers as H
from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
E, D = 65536, 8
xyz, rpy, P = H.c2_setup(E, D, phase="c3")
env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=Physics.DYN, pyb_freq=100, ctrl_freq=100, num_envs=E)
env.set_trajectories(P)
env.step(torch.zeros((E, D, 4), dtype=env.dtype, device=env.device))
T = 50
log = torch.empty((T, E, D, 20), dtype=env.dtype, device=env.device)
for mode in ("log", "nolog", "log", "nolog"):
    for _ in range(2): env.rollout_geometric_fused(0.0, T, log=(mode == "log"), log_out=log if mode == "log" else None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): env.rollout_geometric_fused(0.0, T, log=(mode == "log"), log_out=log if mode == "log" else None)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (10 * T)
    print(mode, "us/step", us, "G drone-steps/s", E * D / us / 1e3)
