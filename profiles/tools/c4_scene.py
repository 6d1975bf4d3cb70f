"""Why do envs of the C4 bench scene fall back to the nominal control?  Runs the C4 loop for 220 steps, then rebuilds the dense
(G, h) of every env at the last state (mds_cbf_rows) and classifies, per env and per row kind (pair / obstacle):
  zero   : a barrier row with G = 0 and h < 0            (an agent level with what it avoids: L_g L_f h = 0)
  reach  : G != 0 but even the corner of the input box leaves the row violated
  solver : neither -- the rows are mutually inconsistent (found by the active-set iterations)
python3 profiles/tools/c4_scene.py [level|far] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
from multidronesim_amd.cbf.cbf import DroneCBF
from multidronesim_amd.cbf.qptracker import DroneQPTracker
from multidronesim_amd.model.linear_omega import LinearizedOmegaModel
from multidronesim_amd.utils.model_conversions import obs_to_lin_model

scene = sys.argv[1] if len(sys.argv) > 1 else "level"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 220
E, D = 4096, 16
xyz, rpy, P = bench.make_inputs(E, D, "c3", 1000)
P[..., 4] = 0.5 + 0.3 * np.arange(D); xyz[..., 2] = 0.5 + 0.3 * np.arange(D)
env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=Physics.DYN, pyb_freq=100, ctrl_freq=100, num_envs=E)
cbf = DroneCBF(env, [LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2, cbf_poles=np.array([-2.2, -2.4]))
tracker = DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
xy, z = (0.5, 0.5) if scene == "level" else (100.0, 0.65)
x_obs = [np.array([[sx * xy, sy * xy, z], [0, 0, 0]]) for sx in (-1, 1) for sy in (-1, 1)]
r_obs = [0.1] * 4
env.set_trajectories(P)
env.step(torch.zeros((E, D, 4), dtype=env.dtype, device=env.device))
env.rollout_cbf_geometric(0.0, steps, tracker, x_obs, r_obs)
obs_before = env._obs.clone()
_, st = env.step_cbf_geometric(steps * env.CTRL_TIMESTEP, tracker, x_obs, r_obs)
st = st.cpu().numpy()
x = obs_to_lin_model(obs_before, 9, env)
xdes = env._lib and torch.zeros_like(x)
# xdes of the step just taken: the library's scratch is not exposed; rebuild it from the trajectories (yaw, vel, pos)
from multidronesim_amd.trajectories.Lemniscate import Lemniscate
des = torch.empty((E * D, 11), dtype=env.dtype, device=env.device)
import ctypes as C
from multidronesim_amd import _capi as capi
capi.check(env._lib.mds_lemniscate_eval(env._h, C.c_double(steps * env.CTRL_TIMESTEP), C.c_void_p(des.data_ptr()), env._stream()), "lem")
des = des.reshape(E, D, 11)
xdes = torch.zeros((E, D, 9), dtype=env.dtype, device=env.device)
xdes[..., 2] = des[..., 9]; xdes[..., 3:6] = des[..., 3:6]; xdes[..., 6:9] = des[..., 0:3]
G, h = cbf.build_ineq_const_batched(x.reshape(E, D, 9), xdes, x_obs, r_obs)
G, h = G.double().cpu().numpy(), h.double().cpu().numpy()
npairs = D * (D - 1) // 2
nobs = D * 4
umax = cbf.umax
rows = {"pair": slice(0, npairs), "obstacle": slice(npairs + 8 * D, npairs + 8 * D + nobs)}
print(f"scene {scene}: {E} envs, step {steps}: fallback fraction {st.mean():.3f}")
cause = np.zeros(E, dtype=int)
for kind, sl in rows.items():
    Gk, hk = G[:, sl, :], h[:, sl]
    Gt = Gk[..., 0::4]                                  # thrust columns only (order 2)
    nz = np.abs(Gt).sum(-1)
    zero = (nz == 0) & (hk < 0)
    reach = (nz > 0) & (hk < -(np.abs(Gt) * umax[0]).sum(-1))
    print(f"  {kind:9s}: envs with a zero-G violated row {zero.any(1).mean():.3f}, with an out-of-reach row {reach.any(1).mean():.3f}; "
          f"min |G| among non-zero {nz[nz > 0].min():.2e}")
    cause |= zero.any(1) * 1 | reach.any(1) * 2
print(f"  fallback envs explained by zero rows {(st & (cause & 1 > 0)).sum() / max(st.sum(), 1):.3f}, by out-of-reach rows only "
      f"{(st & ((cause & 1) == 0) & (cause & 2 > 0)).sum() / max(st.sum(), 1):.3f}, by neither {(st & (cause == 0)).sum() / max(st.sum(), 1):.3f}")
zc = obs_before.reshape(E, D, 20)[..., 2].double().cpu().numpy()
print("  drone heights at that step: plane spacing check, min |z_i - z_j| over pairs per env: median %.3f min %.3f" % (
    np.median([np.abs(zc[e][:, None] - zc[e][None, :])[np.triu_indices(D, 1)].min() for e in range(0, E, 8)]),
    min(np.abs(zc[e][:, None] - zc[e][None, :])[np.triu_indices(D, 1)].min() for e in range(0, E, 8))))
if os.environ.get("C4_SCENE_DEBUG"):
    sl = rows["obstacle"]
    bad_env = np.where(st != 0)[0][:3]
    ob = obs_before.reshape(E, D, 20).double().cpu().numpy()
    for e in bad_env:
        hk = h[e, sl].reshape(D, 4)
        Gk = G[e, sl, 0::4].reshape(D, 4, D)
        d = np.unravel_index(np.argmin(hk), hk.shape)
        print("env", e, "worst obstacle row: drone", d[0], "obstacle", d[1], "h", hk[d], "G", Gk[d[0], d[1], d[0]], "pos", ob[e, d[0], 0:3], "vel", ob[e, d[0], 10:13],
              "rpy", ob[e, d[0], 7:10], "xdes vel", xdes[e, d[0], 3:6].cpu().numpy(), "xdes pos", xdes[e, d[0], 6:9].cpu().numpy())
    print("speeds: max |v| %.2f, max |rpy| %.2f, z range %.2f..%.2f" % (np.abs(ob[..., 10:13]).max(), np.abs(ob[..., 7:9]).max(), ob[..., 2].min(), ob[..., 2].max()))
