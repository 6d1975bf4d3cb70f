"""Do N equal env shards on N streams (independent step chains, out of phase) beat one full-size shard on one stream?
Each shard is its own env on its own stream, stepped by its own C rollout loop on that stream alone (no LDS padding of the
launches: that is a property of the library's internal two-chain split).  python3 profiles/tools/split_streams.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import helpers as H
from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
D, K = 8, 2000
def mk(E, seed):
    xyz, rpy, P = H.c2_setup(E, D, seed=seed, phase="c3")
    env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=Physics.DYN, pyb_freq=100, ctrl_freq=100, num_envs=E)
    env.set_trajectories(P); env.step(torch.zeros((E, D, 4), dtype=env.dtype, device=env.device)); env.set_rollout_streams(1); return env
NS = (1, 2, 3, 4, 6, 8)
parts = {n: [mk(65536 // n, 10 + k) for k in range(n)] for n in NS}
streams = [torch.cuda.Stream() for _ in range(max(NS))]
lib2 = mk(65536, 0); lib2.set_rollout_streams(2)
def run_split(n):
    for k, env in enumerate(parts[n]):
        with torch.cuda.stream(streams[k]):
            env.rollout_geometric(0.0, K, want_obs=True, obs_every_step=True)
def timeit(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e6
for _ in range(2):
    print(" | ".join("%d shard(s): %.2f" % (n, timeit(lambda: run_split(n)) * 65536 / (65536 // n * n)) for n in NS) +
          " | library two-chain split: %.2f us per step of 524288 drones" % timeit(lambda: lib2.rollout_geometric(0.0, K, want_obs=True, obs_every_step=True)), flush=True)
