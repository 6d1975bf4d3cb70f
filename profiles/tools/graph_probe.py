"""Does a hipGraph of the per-step launches beat the plain C loop on the launch-bound C2 shape?  (torch.cuda.CUDAGraph capture
of ctypes launches; t baked per node -- timing experiment only.)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import helpers as H
from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
E, D = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 4)
xyz, rpy, P = H.c2_setup(E, D)
env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=Physics.DYN, pyb_freq=100, ctrl_freq=100, num_envs=E)
env.set_trajectories(P)
env.step(torch.zeros((E, D, 4), dtype=env.dtype, device=env.device))
K = 100
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * K)
print("C loop (mds_rollout_geometric, obs every step): %.2f us/step" % timeit(lambda: env.rollout_geometric(0.0, K, want_obs=True, obs_every_step=True)))
s = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
obs_ptr = C.c_void_p(env._obs.data_ptr())
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        sp = C.c_void_p(s.cuda_stream)
        for k in range(K):
            rc = env._lib.mds_step_geometric(env._h, C.c_double(0.01 * k), obs_ptr, C.c_void_p(None), sp)
            assert rc == 0
print("hipGraph replay of %d step launches:            %.2f us/step" % (K, timeit(lambda: g.replay())))
print("fused rollout kernel (obs log):                  %.2f us/step" % timeit(lambda: env.rollout_geometric_fused(0.0, K, log=True)))
