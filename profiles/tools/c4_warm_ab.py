"""Same C4 window (control steps 20 .. 220 after a reset), persistent kernel, timed several times in one process: repetition 1 starts
on an idle GPU (as bench.py's C4 section does after building its env), the later ones right behind the previous repetition.  If
they differ, the figure depends on where the GPU's clocks are when the 4 ms window starts, not on the kernel.
python3 profiles/tools/c4_warm_ab.py [scene] [spin_ms before repetition 1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
from multidronesim_amd.cbf.cbf import DroneCBF
from multidronesim_amd.cbf.qptracker import DroneQPTracker
from multidronesim_amd.model.linear_omega import LinearizedOmegaModel
scene = sys.argv[1] if len(sys.argv) > 1 else "under"
spin_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
E, D = 16384, 16
xyz, rpy, P = bench.c4_inputs(E, D, 1000)
env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=Physics.DYN, pyb_freq=100, ctrl_freq=100, num_envs=E)
env.set_trajectories(P)
cbf = DroneCBF(env, [LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2, cbf_poles=np.array([-2.2, -2.4]))
trk = DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
x_obs, obs_r = bench.c4_spheres(scene)
dev = env.device
log = torch.empty((50, E, D, 20), dtype=env.dtype, device=dev)
zeros = torch.zeros((E, D, 4), dtype=env.dtype, device=dev)
spin = torch.ones(64 << 20, device=dev)
for rep in range(5):
    env.reset()
    env._lib.mds_lowlevel_reset(env._h, None)
    env.step(zeros)
    torch.cuda.synchronize(dev)
    if rep == 0 and spin_ms > 0:                       # busy the GPU for a while right before the first repetition
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < spin_ms:
            spin.mul_(1.0000001)
        torch.cuda.synchronize(dev)
    if rep == 3:
        time.sleep(0.5)                                # an idle gap like the one before bench.py's C4 section
    env.rollout_cbf_geometric_fused(0.0, 20, trk, x_obs, obs_r, steps_per_launch=20, obs_log=log)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream(dev))
    env.rollout_cbf_geometric_fused(0.2, 200, trk, x_obs, obs_r, steps_per_launch=50, obs_log=log)
    e1.record(torch.cuda.current_stream(dev))
    torch.cuda.synchronize(dev)
    st = env._cbf_status
    print(f"scene {scene} repetition {rep + 1}{' (after 0.5 s idle)' if rep == 3 else ''}{f' (after {spin_ms:.0f} ms of other GPU work)' if rep == 0 and spin_ms else ''}: "
          f"{e0.elapsed_time(e1) * 1e3 / 200:.2f} us per control step, infeasible envs at the last step {int((st != 0).sum())}, "
          f"checksum {float(env._obs.double().abs().sum()):.6f}")
