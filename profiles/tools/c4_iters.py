"""C4 bench scene: histogram of GI iterations / active-set sizes per env (needs <a build with -DMDS_TUNE_ITERS> via MDS_LIB_PATH)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import helpers as H
from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
from multidronesim_amd.cbf.cbf import DroneCBF
from multidronesim_amd.cbf.qptracker import DroneQPTracker
from multidronesim_amd.model.linear_omega import LinearizedOmegaModel
E, D = 16384, 16
xyz, rpy, P = H.c2_setup(E, D, phase="c3")
P[..., 4] = 0.5 + 0.3 * np.arange(D); xyz[..., 2] = 0.5 + 0.3 * np.arange(D)
env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=Physics.DYN, pyb_freq=100, ctrl_freq=100, num_envs=E)
cbf = DroneCBF(env, [LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2, cbf_poles=np.array([-2.2, -2.4]))
tracker = DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
c4_obs = [np.array([[sx * 0.5, sy * 0.5, 0.5], [0, 0, 0]]) for sx in (-1, 1) for sy in (-1, 1)]
env.set_trajectories(P)
env.step(torch.zeros((E, D, 4), dtype=env.dtype, device=env.device))
t = 0.0
for k in range(222):
    obs, st = env.step_cbf_geometric(t, tracker, c4_obs, [0.1] * 4); t += env.CTRL_TIMESTEP
    if k in (0, 20, 100, 221):
        s = st.cpu().numpy()
        fb, it, q, nbox, dur = s & 1, (s >> 1) & 0x7F, (s >> 8) & 0x1F, (s >> 13) & 0x1F, ((s >> 18) & 0x1FFF) * 256
        print(k, "fallback %.3f" % fb.mean(), "mean dur cycles %.0f max %d" % (dur.mean(), dur.max()), "active rows total %d of which box %d" % (q[fb == 0].sum(), nbox[fb == 0].sum()))
        for i in range(0, 18):
            sel = it == i
            if sel.sum(): print("   it=%2d n=%5d dur mean %7.0f | q mean %.1f box mean %.1f fallback %.2f" % (i, sel.sum(), dur[sel].mean(), q[sel].mean(), nbox[sel].mean(), fb[sel].mean()))
