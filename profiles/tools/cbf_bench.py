"""Timing of the ECBF filter kernel alone on a C4-sized batch (tuning script, not the bench contract)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.test_gpu_cbf import c4_scene
from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
from multidronesim_amd.cbf.cbf import DroneCBF
from multidronesim_amd.cbf.qptracker import DroneQPTracker
from multidronesim_amd.model.linear_omega import LinearizedOmegaModel
E, D = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 16
kind = sys.argv[2] if len(sys.argv) > 2 else "active"
obs, xdes, unom, x_obs, obs_r = c4_scene(E, D, seed=3, crowd=None if kind == "active" else 3.0)
env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=np.zeros((D, 3)), initial_rpys=np.zeros((D, 3)), physics=Physics.DYN,
                 pyb_freq=100, ctrl_freq=100, num_envs=E)
cbf = DroneCBF(env, [LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2)
trk = DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
dev = env.device
o = torch.as_tensor(obs, dtype=torch.float32, device=dev); xd = torch.as_tensor(xdes, dtype=torch.float32, device=dev); un = torch.as_tensor(unom, dtype=torch.float32, device=dev)
for _ in range(3): us, st = trk.compute_control_batched(o, xd, un, x_obs, obs_r)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 20
ev0.record()
for _ in range(K): us, st = trk.compute_control_batched(o, xd, un, x_obs, obs_r)
ev1.record(); torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / K
print(json.dumps({"E": E, "D": D, "scene": kind, "ms_per_filter": ms, "drone_steps_per_s": E * D / ms * 1e3, "fallback_frac": float((st != 0).float().mean()),
                  "changed_frac": float(((us[..., 0] - un[..., 0]).abs().amax(dim=1) > 1e-6).float().mean())}))
