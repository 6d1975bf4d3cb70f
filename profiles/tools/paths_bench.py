"""Step time of every widened path at the C3 shape (65536 envs x 8 drones, fp32, DYN Euler, 100 Hz), one MI355X,
inputs resident in HBM, HIP events on the launch stream.  One JSON object per path on stdout.
bytes = algorithmic HBM bytes per drone-step of that path."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
from multidronesim_amd import trajectories as TR
from multidronesim_amd.control import (LQRController, LQROmegaController, LQRYankOmegaController, ThrustOmegaController,
                                       YankOmegaController)
from multidronesim_amd.control.DSLPIDControl import DSLPIDControl
from multidronesim_amd.model import LinearizedModel, LinearizedOmegaModel, LinearizedYankOmegaModel

E, D = int(os.environ.get("PATHS_E", "65536")), 8
K = int(os.environ.get("PATHS_STEPS", "2000"))
dev = torch.device("cuda:0")
xyz, rpy, P = bench.make_inputs(E, D, "c3", 1000)
n = E * D


def mk(physics=Physics.DYN, pyb=100, dtype="float32", hover=False):
    env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=physics,
                     pyb_freq=pyb, ctrl_freq=100, num_envs=E, dtype=dtype)
    env.set_trajectories(P)
    a = torch.full((E, D, 4), float(env.HOVER_RPM) if hover else 0.0, dtype=env.dtype, device=dev)
    env.step(a)
    return env


def timed(fn, steps=K, warm=200):
    fn(warm)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(steps); e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / steps


def report(name, us, bytes_per, note=""):
    print(json.dumps({"path": name, "us_per_step": round(us, 2), "G_drone_steps_per_s": round(n / us * 1e-3, 2),
                      "bytes_per_drone_step": bytes_per, "GBps": round(bytes_per * n / us * 1e-3, 0),
                      "frac_of_8TBps": round(bytes_per * n / us * 1e-3 / 8000, 3), "note": note}), flush=True)


def loop(env, call):
    clock = [0.0]
    def run(k):
        for _ in range(k):
            call(clock[0]); clock[0] += env.CTRL_TIMESTEP
    return run


# 1. geometric controller, Lemniscate planes (the headline kernel, python loop for comparison with the rows below)
env = mk()
report("step_geometric (Lemniscate planes), python loop", timed(loop(env, env.step_geometric)), 212)
env.set_wind(np.array([0.01, -0.02, 0.0]))
report("step_geometric + constant wind force (mds_set_wind)", timed(loop(env, env.step_geometric)), 212)
env.close()

# 2. general trajectories: one compound / circle / rotated / line table per drone slot (segment tables, f64 evaluation)
env = mk()
c = np.array([0.0, 0.0, 1.0])
tr = []
for d in range(D):
    a = xyz[0, d]
    tr.append([TR.CompoundTrajectory([TR.LineTrajectory(start=a, end=a + np.array([0.4, -0.2, 0.3]), speed=0.6),
                                      TR.WaitTrajectory(duration=0.3, position=a + np.array([0.4, -0.2, 0.3]), yaw=0.2),
                                      TR.CircleTrajectory(r=0.3, v=0.5, center=a + np.array([0.4, -0.5, 0.3]), yaw_rate=0.3)]),
               TR.CircleTrajectory(r=1.0, v=1.0, center=a - np.array([1.0, 0, 0]), yaw_rate=0.2),
               TR.Lemniscate(center=a, omega=1.5, yaw_rate=0.1),
               TR.LineTrajectory(start=a, end=a + np.array([2.0, 1.0, 0.5]), speed=0.3)][d % 4])
env.set_trajectories(tr)
report("step_geometric on segment tables (Compound/Circle/Lemniscate/Line per drone slot, the 8 tables shared by all envs)",
       timed(loop(env, env.step_geometric)), 212 + 12, "+ 12 B table index; the 8 shared tables stay in L2")
# the same tables, but every drone owns a private copy (field 39 is unused by the evaluation: it only defeats the sharing)
import ctypes as C
from multidronesim_amd import _capi as capi
from multidronesim_amd.trajectories.base import stream_ptr
rows = [t._segments() for t in tr]
per = [r.shape[0] for r, _ in rows]
segs = np.concatenate([np.concatenate([r for r, _ in rows], axis=0)] * E, axis=0)
segs[:, 39] = np.random.default_rng(0).standard_normal(segs.shape[0])
off = np.concatenate([[0], np.cumsum(per * E)]).astype(np.int32)
comp = np.array([1 if c else 0 for _, c in rows] * E, dtype=np.int32)
anc = np.ascontiguousarray(np.array([t.anchor() for t in tr] * E, dtype=np.float64))
capi.check(env._lib.mds_set_trajectory_segments(env._h, capi.as_double_ptr(np.ascontiguousarray(segs)), off.ctypes.data_as(C.POINTER(C.c_int32)),
                                                comp.ctypes.data_as(C.POINTER(C.c_int32)), capi.as_double_ptr(anc), C.c_int32(segs.shape[0]),
                                                C.c_void_p(stream_ptr(dev))), "mds_set_trajectory_segments")
report("step_geometric on segment tables, a private table per drone (%d segments)" % segs.shape[0], timed(loop(env, env.step_geometric)),
       212 + 12 + 100, "+ 12 B index + the fields the current piece needs: 64..216 B, ~100 B on this mix")
del segs
env.close()

# 3. 12-state LQR (EnvGeometric.py default controller)
env = mk()
LQRController(env, LinearizedModel(env))
report("step_lqr (LQRController + mixer + DYN)", timed(loop(env, env.step_lqr)), 212)
us = timed(lambda k: [env.rollout_geometric_fused(0.0, 50, controller="lqr") for _ in range(k // 50)])
report("rollout_lqr_fused, 50 steps per launch, last obs only", us, (132 + 80) / 50)
env.close()

# 4. LQR-omega + ThrustOmega and LQR-yank-omega + YankOmega (EnvGeometricOmega / EnvGeometricYankOmega loops)
for which in ("lqr_omega", "lqr_yank_omega"):
    env = mk(hover=True)
    if which == "lqr_omega":
        LQROmegaController(env, LinearizedOmegaModel(env), ThrustOmegaController(env))
    else:
        LQRYankOmegaController(env, LinearizedYankOmegaModel(env), YankOmegaController(env))
    env.set_cbf_nominal(which)
    report(f"step_nominal {which} (one launch: LQR + low level + DYN)", timed(loop(env, env.step_nominal)),
           212 + 48 + (80 if which == "lqr_yank_omega" else 0),
           "state R/W 104, traj 28, low-level memory R/W 48, obs W 80" + (", RPM echo of the previous obs row R 80" if which == "lqr_yank_omega" else ""))
    report(f"step_nominal {which}, action wanted (2 launches: nominal, low level + DYN)",
           timed(loop(env, lambda t: env.step_nominal(t, return_action=True))), 132 + 260 + 16, "nominal R 80 W 52; low level R 104 (+80 obs echo) W 156 + action W 16")
    us = timed(lambda k: [env.rollout_geometric_fused(0.0, 50, controller="nominal") for _ in range(k // 50)])
    report(f"rollout_nominal_fused {which}, 50 steps per launch, last obs only", us, (132 + 80 + 2 * 40) / 50)
    env.close()

# 5. DSLPID (PIDEnv.py sim_step)
env = mk()
env.set_dslpid_gains(DSLPIDControl(DroneModel.CF2P))
tp = torch.as_tensor(xyz, dtype=env.dtype, device=dev)
trp = torch.zeros_like(tp)
report("step_dslpid (DSLPID + DYN)", timed(lambda k: [env.step_dslpid(tp, trp) for _ in range(k)]), 212 + 24 + 2 * 28,
       "state R/W 104, targets R 24, PID memory R/W 56, obs W 80")
env.close()

# 6. physics-only env.step with the neighbour-coupled terms
act = None
for phys, nm in ((Physics.DYN, "DYN"), (Physics.PYB_DRAG, "DYN + drag"), (Physics.PYB_GND, "DYN + ground effect"),
                 (Physics.PYB_DW, "DYN + downwash (all pairs of the env)"), (Physics.PYB_GND_DRAG_DW, "DYN + ground effect + drag + downwash")):
    env = mk(physics=phys, hover=True)
    act = torch.full((E, D, 4), float(env.HOVER_RPM), dtype=env.dtype, device=dev)
    extra = 32 if phys in (Physics.PYB_DRAG, Physics.PYB_GND_DRAG_DW) else 0
    report(f"env.step, {nm}", timed(lambda k: [env.step(act) for _ in range(k)]), 212 + extra,
           "R state 52 + origin 12 + action 16, W state 52 + obs 80" + (" + last-RPM R/W 32" if extra else ""))
    env.close()

# 7. RK4, 2 substeps
env = mk(pyb=200)
report("step_geometric, pyb_freq = 2 x ctrl_freq (two substeps in registers)", timed(loop(env, env.step_geometric)), 212)
env.close()
