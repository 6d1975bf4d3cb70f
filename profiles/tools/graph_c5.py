"""C5 shard (262 144 x 2, fp16 storage): mds_rollout_step issued eagerly vs the same call captured once into a hipGraph
(torch.cuda.graph) and replayed -- is the per-step pair of half-shard launches host-bound?  One and two chains, 1000-step episode
(the graph holds one whole episode: 1000 steps, a reset at its start).  python3 profiles/tools/graph_c5.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import bench
from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics

E, D, T, EP = 262144, 2, 64, 1000
xyz, rpy, _ = bench.make_inputs(E, D, "c2", 1000)
print("mode streams  us_per_step (min / median of 5 episodes)", flush=True)
for streams in (1, 2):
    env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=Physics.DYN, pyb_freq=240, ctrl_freq=240,
                     num_envs=E, dtype="float16", device=0)
    env.set_rollout_streams(streams)
    g = torch.Generator(device="cuda").manual_seed(1)
    acts = (env.HOVER_RPM * (1 + 0.05 * torch.randn((8, E, D, 4), device="cuda", generator=g))).clamp(0, env.MAX_RPM).to(env.dtype)
    log = torch.empty((T, E, D, 20), dtype=env.dtype, device="cuda")
    env.reset()
    env.rollout_step(acts, 0, EP, log, episode_len=EP)
    torch.cuda.synchronize()

    def timed(fn):
        out = []
        for r in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) * 1e6 / EP)
        return min(out), float(np.median(out))

    print("eager", streams, "%.2f %.2f" % timed(lambda: env.rollout_step(acts, EP, EP, log, episode_len=EP)), flush=True)
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            env.rollout_step(acts, EP, EP, log, episode_len=EP)
    torch.cuda.current_stream().wait_stream(side)
    graph.replay()
    print("graph", streams, "%.2f %.2f" % timed(graph.replay), flush=True)
    env.close()
