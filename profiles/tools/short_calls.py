"""How long must one mds_rollout_geometric call be for the two-chain issue to beat the caller's stream alone?
C3 shard (65 536 x 8), obs every step.  For each (streams, steps per call): one untimed call of the same length through the
same branch, then `reps` timed calls; prints HIP-event and wall-clock (enqueue .. synchronize) microseconds per control step.
Run on the GPU box from the repo root:  python3 profiles/tools/short_calls.py [reps [steps,steps,...]]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import bench
from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7
step_list = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (5, 20, 50, 100, 200, 500, 2000)
E, D = 65536, 8
xyz, rpy, P = bench.make_inputs(E, D, "c3", 1000)
env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=Physics.DYN,
                 pyb_freq=100, ctrl_freq=100, num_envs=E, dtype="float32", device=0)
env.set_trajectories(P)
env.step(torch.zeros((E, D, 4), dtype=env.dtype, device=env.device))
dev = env.device
st = torch.cuda.current_stream(dev)
print("steps streams used  ev_us_min ev_us_med  wall_us_min wall_us_med", flush=True)
for steps in step_list:
    for streams in (1, 2):
        env.set_rollout_streams(streams)
        env.rollout_geometric(0.0, steps, want_obs=True, obs_every_step=True)
        torch.cuda.synchronize(dev)
        ev, wl = [], []
        for r in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            w0 = time.perf_counter()
            e0.record(st)
            env.rollout_geometric(0.01 * steps * (r + 1), steps, want_obs=True, obs_every_step=True)
            e1.record(st)
            torch.cuda.synchronize(dev)
            wl.append((time.perf_counter() - w0) * 1e6 / steps)
            ev.append(e0.elapsed_time(e1) * 1e3 / steps)
        print(f"{steps:5d} {streams} {env.last_rollout_streams()}   {min(ev):8.2f} {float(np.median(ev)):8.2f}   {min(wl):8.2f} {float(np.median(wl)):8.2f}",
              flush=True)
env.close()
