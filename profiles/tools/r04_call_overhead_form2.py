"""20-step calls of mds_rollout_geometric in launch form 2 (one launch) on config 3: GPU idle for 300 ms before the call against straight after a
2000-step rollout; HIP events and wall clock, median of `reps`.  python3 profiles/tools/r04_call_overhead_form2.py [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import bench
from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
E, D = 65536, 8
xyz, rpy, P = bench.make_inputs(E, D, "c3", 1000)
env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=Physics.DYN, pyb_freq=100, ctrl_freq=100, num_envs=E,
                 dtype="float32", device=0)
env.set_trajectories(P)
env.step(torch.zeros((E, D, 4), dtype=env.dtype, device=env.device))
dev, st = env.device, torch.cuda.current_stream(env.device)
env.rollout_geometric(0.0, 20, obs_every_step=True)
torch.cuda.synchronize(dev)
assert env.last_rollout_form() == 2
for name, prep in (("cold (idle 300 ms)", lambda: time.sleep(0.3)), ("warm (after 2000 steps)", lambda: env.rollout_geometric(1.0, 2000, obs_every_step=True)),
                   ("after a 5-step call + sync (the driver's sequence)", lambda: (time.sleep(0.3), env.rollout_geometric(1.0, 5, obs_every_step=True), torch.cuda.synchronize(dev)))):
    ev, wl = [], []
    for r in range(reps):
        prep()
        if not name.startswith("warm"):
            torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        w0 = time.perf_counter()
        e0.record(st)
        env.rollout_geometric(30.0 + r, 20, obs_every_step=True)
        e1.record(st)
        torch.cuda.synchronize(dev)
        wl.append((time.perf_counter() - w0) * 1e6 / 20)
        ev.append(e0.elapsed_time(e1) * 1e3 / 20)
    print(f"{name:52s} events median {np.median(ev):6.2f} min {min(ev):6.2f}   wall median {np.median(wl):7.2f} min {min(wl):7.2f}   us per control step", flush=True)
