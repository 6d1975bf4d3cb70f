"""What the reference most likely does on infeasible CBF-QPs, made concrete (analysis tool, CPU only, nothing here is pinned):
the C4 loop of simulations/CBFTest.py:303-350 on the float64 oracle, run twice on the same envs of SURVEY 8d's `level` scene --
 (a) the build's modelled policy: an env whose QP is infeasible keeps the nominal input (status 1);
 (b) the reference's flow with oracle/cvxopt_qp.py in cvxopt's place: the solver returns (status 'unknown' at the iteration limit), so
     _rectify reports success and the LAST ITERATE is applied to every drone of the env (cbf/qptracker.py:103-112, :28-31).
Feasible env-steps use the exact minimiser in both runs (the restated interior point ends within its tolerances of it).
python3 tests/tools/c4_cvxopt_probe.py [envs] [steps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench                                        # noqa: E402  (scene generator only)
from oracle import cvxopt_qp as CQ                  # noqa: E402
from oracle import np_oracle as O                   # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 6
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
D = 16
c = O.CF2P
Kcbf, umax = np.array([5.28, 4.6]), np.array([c.MAX_THRUST, 10, 10, 10])
xyz, rpy, P = bench.c4_inputs(E, D, 1000)
x_obs, obs_r = bench.c4_spheres("level")
x_obs = np.array(x_obs)


def run(policy):
    n = E * D
    Pf = P.reshape(-1, 7)
    ora = O.AviaryOracle(xyz.reshape(-1, 3), rpy.reshape(-1, 3), c, 100, 100, drones_per_env=D)
    ll = O.ThrustOmegaOracle(n, c)
    obs = ora.step(np.zeros((n, 4)))
    t, infeasible, applied = 0.0, 0, []
    for k in range(steps):
        pos, vel, acc, yaw, yd = O.lemniscate(t, Pf[:, 0], Pf[:, 1], Pf[:, 2:5], Pf[:, 5], Pf[:, 6])
        force, w_des, _ = O.geometric_compute(obs, pos, vel, acc, yaw, yd, c, return_omegas=True)
        unom = np.concatenate([(force - c.M * c.G)[:, None], w_des], axis=1)
        xdes = np.concatenate([np.zeros((n, 2)), yaw[:, None], vel, pos], axis=1)
        x = O.obs_to_lin_model(obs, 9)
        usafe = unom.copy()
        for e in range(E):
            sl = slice(e * D, (e + 1) * D)
            G, h = O.cbf_rows(x[sl], xdes[sl], 2, Kcbf, umax, 0.1, 1.0, c, x_obs=x_obs, obs_r=obs_r)
            ok, u, _ = O.qp_project(unom[sl].reshape(-1), G, h)
            if ok:
                usafe[sl] = u.reshape(D, 4)
            else:
                infeasible += 1
                if policy == "reference-like":
                    _, ui, sol = CQ.rectify(unom[sl], G, h)
                    usafe[sl] = ui
                    applied.append((k, e, sol["status"], sol["iterations"], float(np.abs(ui - unom[sl]).max()), float(np.abs(ui[:, 0]).max())))
        usafe[:, 0] += c.M * c.G
        obs = ora.step(ll.compute_low_level(usafe, obs, ora.CTRL_TIMESTEP))
        t += ora.CTRL_TIMESTEP
    return obs.reshape(E, D, 20), infeasible, applied


oa, na, _ = run("modelled")
ob, nb, applied = run("reference-like")
print(f"C4 `level` scene (bench.py's generator, seed 1000), {E} envs x {D} drones x {steps} control steps (float64 oracle)")
print(f"(a) modelled fallback (u_hat on infeasible envs):           {na} infeasible env-steps of {E * steps}")
print(f"(b) reference-like (restated interior point's last iterate): {nb} infeasible env-steps of {E * steps}")
if applied:
    a = np.array([(x[3], x[4], x[5]) for x in applied], dtype=float)
    print(f"    every one of them: status {set(x[2] for x in applied)}, iterations {int(a[:, 0].min())}..{int(a[:, 0].max())} (limit {CQ.MAXITERS}); "
          f"max |u - u_hat| per solve: median {np.median(a[:, 1]):.3g}, max {a[:, 1].max():.3g}; max |thrust variable| {a[:, 2].max():.3g} N")
print(f"state after {steps} steps, (a) against (b): max |position difference| {np.abs(oa[..., :3] - ob[..., :3]).max():.3g} m, "
      f"finite (a) {bool(np.isfinite(oa).all())} (b) {bool(np.isfinite(ob).all())}")
