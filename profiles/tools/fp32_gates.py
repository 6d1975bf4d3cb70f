"""fp32 vs compensated fp32 (MDS_F32C) against the float64 oracle: open-loop 240 Hz flight, the C4 closed loop and the order-3
closed loop of the parity tests, max |state error| by step count.  python3 profiles/tools/fp32_gates.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import np_oracle as O
from tests import helpers as H
from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
from multidronesim_amd.cbf.cbf import DroneCBF
from multidronesim_amd.cbf.qptracker import DroneQPTracker
from multidronesim_amd.model import LinearizedOmegaModel, LinearizedYankOmegaModel
from multidronesim_amd.control import LQRYankOmegaController, YankOmegaController

def npo(t): return t.detach().double().cpu().numpy().reshape(-1, 20)

# 1. open loop
n = 256
xyz, rpy, ph = H.open_loop_setup(n)
for dtype in ("float32", "float32c"):
    ora = O.AviaryOracle(xyz, rpy, pyb_freq=240, ctrl_freq=240)
    env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=1, initial_xyzs=xyz[:, None, :], initial_rpys=rpy[:, None, :], physics=Physics.DYN,
                     pyb_freq=240, ctrl_freq=240, num_envs=n, dtype=dtype)
    s0 = env.get_state().reshape(n, 13)
    ora.pos, ora.quat, ora.vel, ora.rates = s0[:, 0:3].copy(), s0[:, 3:7].copy(), s0[:, 7:10].copy(), s0[:, 10:13].copy()
    out = []
    for k in range(2000):
        a = H.open_loop_rpm(k, ora.CTRL_TIMESTEP, ph)
        obs = ora.step(a)
        g, *_ = env.step(torch.as_tensor(a.reshape(n, 1, 4), dtype=torch.float32))
        if k + 1 in (500, 1000, 2000):
            st = env.get_state().reshape(n, 13)
            out.append((k + 1, np.abs(npo(g)[:, :16] - obs[:, :16]).max(),
                        np.abs(st - np.concatenate([ora.pos, ora.quat, ora.vel, ora.rates], axis=1)).max()))
    print("open loop", dtype, " ".join(f"{k}: obs {e:.2e} state {s:.2e}" for k, e, s in out), flush=True)
    env.close()

# 2. C4 closed loop (test_c4_closed_loop_matches_oracle scene)
E, D = 8, 6
xyz, rpy, P = H.c2_setup(E, D, phase="c3", offset=0.0, omega=1.0)
xyz[..., 2] = 0.5 + 0.25 * np.arange(D)
P[..., 4] = 0.5 + 0.12 * np.arange(D)
x_obs = [np.array([[sx * 0.5, sy * 0.5, 0.5], [0, 0, 0]]) for sx in (-1, 1) for sy in (-1, 1)]
obs_r = [0.1] * 4
marks = (50, 150, 300)
ref = None
for dtype in ("float64", "float32", "float32c"):
    env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=Physics.DYN, pyb_freq=100, ctrl_freq=100, num_envs=E, dtype=dtype)
    env.set_trajectories(P)
    cbf = DroneCBF(env, [LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2)
    trk = DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
    if ref is None:
        ref = {}
        for m in marks:
            ref[m] = H.oracle_cbf_closed_loop(xyz, rpy, P, m, cbf.Kcbf.reshape(-1), cbf.umax, 0.1, 1.0, x_obs, obs_r)
    env.step(torch.zeros((E, D, 4), dtype=env.dtype))
    t, out, mism = 0.0, [], 0
    for k in range(max(marks)):
        g, st = env.step_cbf_geometric(t, trk, x_obs, obs_r)
        t += env.CTRL_TIMESTEP
        if k + 1 in marks:
            oobs, oh = ref[k + 1]
            out.append((k + 1, np.abs(g.double().cpu().numpy()[..., :16] - oobs[..., :16]).max(), int((st.cpu().numpy() != oh[k]).sum())))
    print("c4 loop", dtype, " ".join(f"{k}: {e:.2e} (status mismatches {m})" for k, e, m in out), flush=True)
    env.close()

# 3. order-3 closed loop (test_order3_closed_loop_matches_oracle scene)
E, D = 4, 4
xyz, rpy, P = H.c2_setup(E, D, phase="c3", offset=0.0, omega=0.5)
xyz[..., 2] = 0.5 + 0.6 * np.arange(D)
P[..., 4] = 0.5 + 0.6 * np.arange(D)
x_obs = [np.array([[0.0, 0.0, -0.3], [0, 0, 0], [0, 0, 0]])]
obs_r = [0.1]
marks = (90, 150)
ref = None
for dtype in ("float64", "float32", "float32c"):
    env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=Physics.DYN, pyb_freq=100, ctrl_freq=100, num_envs=E, dtype=dtype)
    env.set_trajectories(P)
    LQRYankOmegaController(env, LinearizedYankOmegaModel(env), YankOmegaController(env))
    cbf = DroneCBF(env, [LinearizedYankOmegaModel(env) for _ in range(D)], safety_radius=0.125, zscale=2.0, order=3, cbf_poles=np.array([-3.0, -3.6, -5.6]))
    trk = DroneQPTracker(cbf, order=3, num_robots=D, xdim=10, env=env)
    env.set_cbf_nominal("lqr_yank_omega")
    if ref is None:
        ref = {m: H.oracle_cbf_closed_loop(xyz, rpy, P, m, cbf.Kcbf.reshape(-1), cbf.umax, 0.125, 2.0, x_obs, obs_r, nominal="lqr_yank_omega", order=3,
                                           first_rpm=O.CF2P.HOVER_RPM) for m in marks}
    env.step(torch.full((E, D, 4), O.CF2P.HOVER_RPM, dtype=env.dtype))
    t, out = 0.0, []
    for k in range(max(marks)):
        g, st = env.step_cbf_geometric(t, trk, x_obs, obs_r)
        t += env.CTRL_TIMESTEP
        if k + 1 in marks:
            oobs, oh = ref[k + 1]
            gg = g.double().cpu().numpy()
            out.append((k + 1, np.abs(gg[..., :16] - oobs[..., :16]).max(), np.abs(gg[..., 16:] - oobs[..., 16:]).max() / O.CF2P.HOVER_RPM,
                        int((st.cpu().numpy() != oh[k]).sum())))
    print("order-3 loop", dtype, " ".join(f"{k}: state {e:.2e} rpm rel {r:.2e} (status mismatches {m})" for k, e, r, m in out), flush=True)
    env.close()
