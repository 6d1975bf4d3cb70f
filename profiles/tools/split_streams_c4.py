"""C4 (geometric nominal -> ECBF QP -> ThrustOmega -> DYN): do two half-size env shards stepping on two streams beat one
full shard on one stream?  The QP kernel is latency/ALU bound, the other two kernels are memory bound."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
from multidronesim_amd.cbf.cbf import DroneCBF
from multidronesim_amd.cbf.qptracker import DroneQPTracker
from multidronesim_amd.model.linear_omega import LinearizedOmegaModel
D, K = 16, 1000
dev = torch.device("cuda:0")
c4_obs = [np.array([[sx * 0.5, sy * 0.5, 0.5], [0, 0, 0]]) for sx in (-1, 1) for sy in (-1, 1)]
c4_r = [0.1] * 4
def mk(E, seed):
    xyz, rpy, P = bench.make_inputs(E, D, "c3", seed)
    P[..., 4] = 0.5 + 0.3 * np.arange(D); xyz[..., 2] = 0.5 + 0.3 * np.arange(D)
    env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=Physics.DYN,
                     pyb_freq=100, ctrl_freq=100, num_envs=E, dtype="float32")
    cbf = DroneCBF(env, [LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2, cbf_poles=np.array([-2.2, -2.4]))
    tr = DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
    env.set_trajectories(P)
    env.step(torch.zeros((E, D, 4), dtype=env.dtype, device=dev))
    return env, tr
full = mk(16384, 1000)
halves = [mk(8192, 2000 + k) for k in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
clock = {"full": 0.0, "split": 0.0}
def run_full(k):
    env, tr = full
    for _ in range(k):
        env.step_cbf_geometric(clock["full"], tr, c4_obs, c4_r); clock["full"] += 0.01
def run_split(k, lag=0):
    for j in range(k):
        for s, (env, tr) in enumerate(halves):
            with torch.cuda.stream(streams[s]):
                env.step_cbf_geometric(clock["split"], tr, c4_obs, c4_r)
        clock["split"] += 0.01
def timeit(fn, k):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(k); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e6
run_full(200); run_split(200)
for _ in range(3):
    print("one shard 16384 envs: %.2f us/step | 2 shards x 8192 on 2 streams: %.2f" % (timeit(run_full, K), timeit(run_split, K)), flush=True)
