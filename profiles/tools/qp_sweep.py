"""Ad-hoc stress of the QP kernel against the oracle's active-set QP on randomised crowded scenes (the scene generator of
tests/test_gpu_cbf.py): statuses must be equal, solutions within tol.  python3 profiles/tools/qp_sweep.py [E] [seeds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import multidronesim_amd as mds_pkg  # noqa: F401
from oracle import np_oracle as O
from tests import test_gpu_cbf as T


class M:  # the `mds` fixture's surface
    pass


import torch
from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
from multidronesim_amd.cbf.cbf import DroneCBF
from multidronesim_amd.cbf.qptracker import DroneQPTracker
from multidronesim_amd.model.linear_omega import LinearizedOmegaModel
m = M()
m.CtrlAviary, m.DroneModel, m.Physics, m.torch = CtrlAviary, DroneModel, Physics, torch
m.DroneCBF, m.DroneQPTracker, m.LinearizedOmegaModel = DroneCBF, DroneQPTracker, LinearizedOmegaModel

E = int(sys.argv[1]) if len(sys.argv) > 1 else 256
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
D = 16
for dtype, tol in (("float64", 1e-8), ("float32", 3e-5)):
    bad_status = bad_sol = active = fallback = total = 0
    worst = 0.0
    its = []
    t0 = time.time()
    for seed in range(100, 100 + seeds):
        for dz, vz in ((0.3, 0.35), (0.2, 0.7)):
            obs, xdes, unom, x_obs, obs_r = T.c4_scene(E, D, seed=seed, dz=dz, vz=vz)
            env = T.make_env(m, E, D, dtype)
            cbf = DroneCBF(env, [LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2, cbf_poles=np.array([-2.2, -2.4]))
            trk = DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
            us, st = trk.compute_control_batched(obs, xdes, unom, x_obs, obs_r)
            us, st = us.double().cpu().numpy(), st.cpu().numpy()
            its.append(cbf.last_iterations().cpu().numpy())
            for e in range(E):
                x = O.obs_to_lin_model(obs[e], 9)
                u_ref, status = O.cbf_filter(x, xdes[e], unom[e], 2, cbf.Kcbf.reshape(-1), cbf.umax, 0.1, 1.0, O.CF2P, np.array(x_obs), obs_r)
                total += 1
                if st[e] != status:
                    bad_status += 1
                    continue
                if status == 0:
                    d = float(np.abs(us[e] - u_ref).max())
                    worst = max(worst, d)
                    bad_sol += d > tol
                    active += int(np.abs(u_ref[:, 0] - unom[e][:, 0]).max() > 1e-6)
                else:
                    fallback += 1
            env.close()
    its = np.concatenate(its)
    print(f"{dtype}: {total} envs, status mismatches {bad_status}, solutions beyond {tol:g}: {bad_sol} (worst {worst:.3g}), active {active}, fallback {fallback}, "
          f"iterations mean {its.mean():.2f} max {its.max()}, {time.time() - t0:.0f} s", flush=True)
