"""Catch an env of the fp32 order-3 loop whose QP hits the iteration cap, rebuild the filter's inputs for it (obs of the step, xdes, u_hat) and check
that mds_cbf_filter on those inputs alone reproduces the long solve; dump them -> gpurun_out/r04_o3_cycle.npz.  python3 profiles/tools/r04_o3_capture.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from oracle import np_oracle as O
from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
from multidronesim_amd.cbf.cbf import DroneCBF
from multidronesim_amd.cbf.qptracker import DroneQPTracker
from multidronesim_amd.model.linear_yank_omega import LinearizedYankOmegaModel
from multidronesim_amd.control import LQRYankOmegaController, YankOmegaController
E, D = 4096, 8
xyz, rpy, P = bench.c4_inputs(E, D, 1000)
P[..., 1] = 0.5
env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=Physics.DYN, pyb_freq=100, ctrl_freq=100, num_envs=E, dtype="float32", device=0)
env.set_trajectories(P)
ctrl = LQRYankOmegaController(env, LinearizedYankOmegaModel(env), YankOmegaController(env))
cbf = DroneCBF(env, [LinearizedYankOmegaModel(env) for _ in range(D)], safety_radius=0.125, zscale=2.0, order=3, cbf_poles=np.array([-3.0, -3.6, -5.6]))
trk = DroneQPTracker(cbf, order=3, num_robots=D, xdim=10, env=env)
env.set_cbf_nominal("lqr_yank_omega")
x_obs, obs_r = [np.array([[0.0, 0.0, -3.0], [0, 0, 0], [0, 0, 0]])], [0.1]
env.step(torch.full((E, D, 4), float(env.HOVER_RPM), dtype=env.dtype, device=env.device))
t = 0.0
caught = []
for k in range(220):
    prev = env._obs.clone()
    env.step_cbf_geometric(t, trk, x_obs, obs_r)
    it = cbf.last_iterations()
    if int(it.max()) >= 500:
        e = int(it.argmax())
        caught.append((k, e, int(it.max()), prev[e].double().cpu().numpy().copy(), t))
        print("step", k, "env", e, "iterations", int(it.max()), "envs >= 500:", int((it >= 500).sum()), flush=True)
    t += env.CTRL_TIMESTEP
print("caught", len(caught))
out = {}
for j, (k, e, its, ob, tk) in enumerate(caught[:4]):
    Pe = P[e]
    pos, vel, acc, yaw, yd = O.lemniscate(tk, Pe[:, 0], Pe[:, 1], Pe[:, 2:5], Pe[:, 5], Pe[:, 6])
    des = np.zeros((D, 11)); des[:, 0:3], des[:, 3:6], des[:, 6:9], des[:, 9], des[:, 10] = pos, vel, acc, yaw, yd
    obs_all = np.broadcast_to(ob, (E, D, 20)).copy()
    des_all = np.broadcast_to(des, (E, D, 11)).copy()
    u = ctrl.compute_batched(obs_all, des_all)[0].double().cpu().numpy()
    unom = u.copy(); unom[:, 0] -= O.CF2P.M * O.CF2P.G
    xdes = np.zeros((D, 10)); xdes[:, 2] = yaw; xdes[:, 3] = O.CF2P.M * O.CF2P.G; xdes[:, 4:7] = vel; xdes[:, 7:10] = pos
    us, st = trk.compute_control_batched(obs_all, np.broadcast_to(xdes, (E, D, 10)).copy(), np.broadcast_to(unom, (E, D, 4)).copy(), x_obs, obs_r)
    it2 = cbf.last_iterations()
    print("replay of step", k, "env", e, ": iterations", int(it2[0]), "status", int(st[0]), "(in the loop:", its, ")", flush=True)
    out[f"obs{j}"], out[f"xdes{j}"], out[f"unom{j}"], out[f"its{j}"], out[f"replay_its{j}"] = ob, xdes, unom, its, int(it2[0])
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/r04_o3_cycle.npz", Kcbf=cbf.Kcbf.reshape(-1), umax=cbf.umax, **out)
env.close()
