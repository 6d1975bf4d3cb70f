"""Where do the ~35 us that a 20-step two-chain call costs over 20 x its steady-state step go?  (The driver's command is one such call:
`bench.py --steps 20 --warmup 5`.)  C3 shard (65 536 x 8), obs every step.  For each variant, `reps` timed 20-step calls, HIP events and
wall clock (enqueue .. synchronize), median us per control step:
  cold      the GPU idle for 300 ms before the call (what the driver's command sees after its barrier)
  warm      a 2000-step rollout enqueued right before the timed call (no idle gap: clocks up, caches warm)
  graph     the same 20-step call captured once into a hipGraph (torch.cuda.graph) and replayed, cold and warm
for streams = 1 and 2.  Run on the GPU box from the repo root:  python3 profiles/tools/r04_call_overhead.py [steps [reps]]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import bench
from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
E, D = 65536, 8
xyz, rpy, P = bench.make_inputs(E, D, "c3", 1000)
env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=Physics.DYN,
                 pyb_freq=100, ctrl_freq=100, num_envs=E, dtype="float32", device=0)
env.set_trajectories(P)
env.step(torch.zeros((E, D, 4), dtype=env.dtype, device=env.device))
env.set_rollout_form(1)
dev = env.device
st = torch.cuda.current_stream(dev)


def timed(fn, pre):
    ev, wl = [], []
    for r in range(reps):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        w0 = time.perf_counter()
        e0.record(st)
        fn()
        e1.record(st)
        torch.cuda.synchronize(dev)
        wl.append((time.perf_counter() - w0) * 1e6 / steps)
        ev.append(e0.elapsed_time(e1) * 1e3 / steps)
    return float(np.median(ev)), float(min(ev)), float(np.median(wl)), float(min(wl))


def cold():
    torch.cuda.synchronize(dev)
    time.sleep(0.3)


def warm():
    env.rollout_geometric(0.0, 2000, want_obs=True, obs_every_step=True)


print(f"{steps}-step calls, {reps} reps: variant streams  ev_med ev_min  wall_med wall_min   (us per control step)", flush=True)
for streams in (1, 2):
    env.set_rollout_streams(streams)
    call = lambda: env.rollout_geometric(0.0, steps, want_obs=True, obs_every_step=True)
    call()
    torch.cuda.synchronize(dev)
    for name, pre in (("cold", cold), ("warm", warm)):
        r = timed(call, pre)
        print(f"eager-{name:5s} {streams}   {r[0]:7.2f} {r[1]:7.2f}   {r[2]:7.2f} {r[3]:7.2f}", flush=True)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(st)
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            env.rollout_geometric(0.0, steps, want_obs=True, obs_every_step=True)
    st.wait_stream(side)
    g.replay()
    torch.cuda.synchronize(dev)
    for name, pre in (("cold", cold), ("warm", warm)):
        r = timed(g.replay, pre)
        print(f"graph-{name:5s} {streams}   {r[0]:7.2f} {r[1]:7.2f}   {r[2]:7.2f} {r[3]:7.2f}", flush=True)
env.close()
