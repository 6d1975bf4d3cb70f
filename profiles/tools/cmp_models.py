"""mds_compare_models on a C3-sized observation log (524288 rows): microseconds per launch by HIP events, for several numbers of
back-to-back launches (host enqueue time beside it).
python3 profiles/tools/cmp_models.py [rows]        (MDS_TUNE_CMP_LDS=<bytes> limits the resident workgroups per CU)"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
from multidronesim_amd.model import LinearizedModel, QuadrotorDynamics
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
E, D = rows // 8, 8
xyz, rpy, P = bench.make_inputs(E, D, "c3", 1000)
env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=Physics.DYN, pyb_freq=100, ctrl_freq=100,
                 num_envs=E, dtype=os.environ.get("CMP_DTYPE", "float32"))
env.set_trajectories(P)
env.step(torch.zeros((E, D, 4), dtype=env.dtype, device=env.device))
dev = env.device
obs = env.step_geometric(0.0).reshape(-1, 20).contiguous()
n, es = obs.shape[0], obs.element_size()
lin, geo = LinearizedModel(env), QuadrotorDynamics(env.PYB_FREQ)
geo.load_env_params(env)
A, B = lin._mats()
PD = C.POINTER(C.c_double)
Ap, Bp = A.ctypes.data_as(PD), B.ctypes.data_as(PD)
J = (C.c_double * 3)(1.05, 1.05, 2.05)
outs = [torch.empty((n, 12), dtype=obs.dtype, device=dev) for _ in range(3)]
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
args = (env._h, C.c_int(n), C.c_void_p(obs.data_ptr()), Ap, Bp, C.c_double(lin.mass * lin.g), C.c_double(geo.m), J, C.c_double(geo.g),
        C.c_void_p(outs[0].data_ptr()), C.c_void_p(outs[1].data_ptr()), C.c_void_p(outs[2].data_ptr()), st)
fn = env._lib.mds_compare_models
for _ in range(10):
    fn(*args)
torch.cuda.synchronize(dev)
res = {"rows": n, "pad": os.environ.get("MDS_TUNE_CMP_LDS", "0"), "dtype": str(obs.dtype)}
for reps in (20, 50, 200, 1000):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    e0.record(torch.cuda.current_stream(dev))
    t0 = time.perf_counter()
    for _ in range(reps):
        fn(*args)
    t1 = time.perf_counter()
    e1.record(torch.cuda.current_stream(dev))
    torch.cuda.synchronize(dev)
    us = e0.elapsed_time(e1) * 1e3 / reps
    res[f"reps{reps}"] = {"us_per_launch": round(us, 2), "host_us_per_call": round((t1 - t0) * 1e6 / reps, 2),
                          "GBps": round(n * 56 * es / us / 1e3, 1)}
print(json.dumps(res))
