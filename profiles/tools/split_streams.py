"""Do two half-size env shards on two streams (independent step chains, out of phase) beat one full-size shard on one stream?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import helpers as H
from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
D, K = 8, 2000
def mk(E, seed):
    xyz, rpy, P = H.c2_setup(E, D, seed=seed, phase="c3")
    env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=Physics.DYN, pyb_freq=100, ctrl_freq=100, num_envs=E)
    env.set_trajectories(P); env.step(torch.zeros((E, D, 4), dtype=env.dtype, device=env.device)); return env
full = mk(65536, 0)
parts = {n: [mk(65536 // n, 10 + k) for k in range(n)] for n in (2, 4)}
streams = [torch.cuda.Stream() for _ in range(4)]
def run_full():
    full.rollout_geometric(0.0, K, want_obs=True, obs_every_step=True)
def run_split(n):
    for k, env in enumerate(parts[n]):
        with torch.cuda.stream(streams[k]):
            env.rollout_geometric(0.0, K, want_obs=True, obs_every_step=True)
def timeit(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e6
for _ in range(2):
    print("one shard 65536 envs, one stream: %.2f us/step | 2 shards x 32768 on 2 streams: %.2f | 4 shards x 16384 on 4 streams: %.2f" % (timeit(run_full), timeit(lambda: run_split(2)), timeit(lambda: run_split(4))))
