"""The CBF loop away from the C4 shape, step by step (QP launch + low-level launch from C: mds_rollout_cbf_geometric) against the persistent kernels
(mds_rollout_cbf_geometric_fused, 50 steps per launch), us per control step over steps 20..220, HIP events:
  order 2 (geometric nominal, `under` spheres): 16 384 envs x D for D = 2 (simulations/CBFTest.py's default), 7, 8, 16; fp32, and fp32c / float64 at D = 16
  order 3 (simulations/CBFTestOrd3.py: lqr-yank-omega nominal, YankOmega low level, poles of :452): 4 096 envs x 7 and x 8, fp32 and float64
python3 profiles/tools/r04_cbf_shapes.py [out.json]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
from multidronesim_amd.cbf.cbf import DroneCBF
from multidronesim_amd.cbf.qptracker import DroneQPTracker
from multidronesim_amd.model.linear_omega import LinearizedOmegaModel
from multidronesim_amd.model.linear_yank_omega import LinearizedYankOmegaModel
from multidronesim_amd.control import LQRYankOmegaController, YankOmegaController

out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r04_cbf_shapes.json"
rows = []


def run(order, E, D, dtype):
    xyz, rpy, P = bench.c4_inputs(E, D, 1000)
    if order == 3:
        P[..., 1] = 0.5
    res = {}
    for form in ("step", "persistent"):
        env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=Physics.DYN, pyb_freq=100, ctrl_freq=100,
                         num_envs=E, dtype=dtype, device=0)
        env.set_trajectories(P)
        if order == 2:
            cbf = DroneCBF(env, [LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2, cbf_poles=np.array([-2.2, -2.4]))
            trk = DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
            x_obs, obs_r = bench.c4_spheres("under")
            env.step(torch.zeros((E, D, 4), dtype=env.dtype, device=env.device))
        else:
            LQRYankOmegaController(env, LinearizedYankOmegaModel(env), YankOmegaController(env))
            cbf = DroneCBF(env, [LinearizedYankOmegaModel(env) for _ in range(D)], safety_radius=0.125, zscale=2.0, order=3, cbf_poles=np.array([-3.0, -3.6, -5.6]))
            trk = DroneQPTracker(cbf, order=3, num_robots=D, xdim=10, env=env)
            env.set_cbf_nominal("lqr_yank_omega")
            x_obs, obs_r = [np.array([[0.0, 0.0, -3.0], [0, 0, 0], [0, 0, 0]])], [0.1]
            env.step(torch.full((E, D, 4), float(env.HOVER_RPM), dtype=env.dtype, device=env.device))
        dt = env.CTRL_TIMESTEP
        if form == "step":
            env.set_rollout_streams(0)
            env.rollout_cbf_geometric(0.0, 20, trk, x_obs, obs_r)
            us = bench._timed_steps(env.device, lambda: env.rollout_cbf_geometric(20 * dt, 200, trk, x_obs, obs_r), 200)
        else:
            ring = torch.empty((50, E, D, 20), dtype=env.dtype, device=env.device)
            env.rollout_cbf_geometric_fused(0.0, 20, trk, x_obs, obs_r, steps_per_launch=50, obs_log=ring)
            us = bench._timed_steps(env.device, lambda: env.rollout_cbf_geometric_fused(20 * dt, 200, trk, x_obs, obs_r, steps_per_launch=50, obs_log=ring), 200)
        fb = float((env._cbf_status != 0).float().mean())
        it = float(cbf.last_iterations().float().mean())
        sane = bool(torch.isfinite(env._obs).all().item())
        res[form] = {"us_per_step": us, "G_drone_steps_per_s": E * D / us * 1e-3, "kernel": env.cbf_last_step_kernel(), "fallback_last": fb, "iterations_last_mean": it, "sane": sane}
        env.close()
        torch.cuda.empty_cache()
    row = {"order": order, "envs": E, "drones_per_env": D, "dtype": dtype, **res}
    rows.append(row)
    print(order, E, D, dtype, "step %.2f us (kernel %d)" % (res["step"]["us_per_step"], res["step"]["kernel"]),
          "persistent %.2f us (kernel %d)" % (res["persistent"]["us_per_step"], res["persistent"]["kernel"]), "fallback", res["persistent"]["fallback_last"], flush=True)


for D in (2, 7, 8, 16):
    run(2, 16384, D, "float32")
run(2, 16384, 16, "float32c")
run(2, 16384, 16, "float64")
for D in (7, 8):
    for dtype in ("float32", "float64"):
        run(3, 4096, D, dtype)
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
json.dump({"what": __doc__.split("\n")[0], "rows": rows}, open(out_path, "w"), indent=1)
print("wrote", out_path)
