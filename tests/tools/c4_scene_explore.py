"""Scene exploration for BASELINE config 4 on the CPU oracle (test tooling; nothing here is shipped or timed).

Runs the CBFTest.py:303-350 loop (geometric nominal -> order-2 ECBF QP -> ThrustOmega -> DYN step) of SURVEY 8d's generator on a
few envs with the oracle's own pieces, the rows vectorised over envs (same closed form, oracle._cbf_pair_terms) and the thrust QP
(the only coupled part at order 2) through oracle.qp_project, and reports per step: share of envs whose QP is infeasible
(status 1), share that needed the solver (a violated row at the nominal input), number of active rows.

    python tests/tools/c4_scene_explore.py --dz 0.3 --obs-xy 0.5 --obs-z 0.5 --omega 1.5 --envs 16 --steps 220
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import np_oracle as O  # noqa: E402


def make_inputs(E, D, seed, dz, dz_start, omega, offset, a=1.0, phase="c3"):
    rng = np.random.default_rng(seed)
    cen = np.zeros((E, D, 3))
    cen[..., :2] = rng.uniform(-offset, offset, size=(E, 1, 2))
    cen[..., 2] = 0.5
    ang = 2 * np.pi * np.arange(D) / D
    xyz = cen.copy()
    xyz[..., 0] += np.sin(ang)
    xyz[..., 1] += np.cos(ang)
    P = np.zeros((E, D, 7))
    P[..., 0], P[..., 1], P[..., 2:5] = a, omega, cen
    P[..., 6] = 2 * np.pi * np.arange(D) / (D + 0.25) if phase == "c3" else -(np.pi / 4) * (np.arange(D) - 1)
    P[..., 4] = 0.5 + dz * np.arange(D)
    xyz[..., 2] = 0.5 + dz_start * np.arange(D)
    return xyz, np.zeros((E, D, 3)), P


def run(E, D, steps, dz, dz_start, omega, offset, obs_xy, obs_z, obs_rel, seed=1000, safety_radius=0.1, a=1.0, verbose=True, start_on_traj=False,
        spheres=None):
    c = O.CF2P
    xyz, rpy, P = make_inputs(E, D, seed, dz, dz_start, omega, offset, a)
    Pf = P.reshape(-1, 7)
    if start_on_traj:
        pos0, *_ = O.lemniscate(0.0, Pf[:, 0], Pf[:, 1], Pf[:, 2:5], Pf[:, 5], Pf[:, 6])
        xyz = pos0.reshape(E, D, 3).copy()
    n = E * D
    Kcbf = O.place_poles_chain([-2.2, -2.4])
    umax0 = c.MAX_THRUST
    I, J = np.triu_indices(D, 1)
    if spheres is None:
        spheres = np.array([[sx * obs_xy, sy * obs_xy, obs_z] for sx in (-1, 1) for sy in (-1, 1)])
    r_obs = 0.1
    ora = O.AviaryOracle(xyz.reshape(-1, 3), rpy.reshape(-1, 3), c, 100, 100, drones_per_env=D)
    ll = O.ThrustOmegaOracle(n, c)
    obs = ora.step(np.zeros((n, 4)))
    t = 0.0
    hist = []
    for k in range(steps):
        pos, vel, acc, yaw, yd = O.lemniscate(t, Pf[:, 0], Pf[:, 1], Pf[:, 2:5], Pf[:, 5], Pf[:, 6])
        force, w_des, _ = O.geometric_compute(obs, pos, vel, acc, yaw, yd, c, return_omegas=True)
        unom = np.concatenate([(force - c.M * c.G)[:, None], w_des], axis=1)
        xdes = np.concatenate([np.zeros((n, 2)), yaw[:, None], vel, pos], axis=1).reshape(E, D, 9)
        x = O.obs_to_lin_model(obs, 9).reshape(E, D, 9)
        hp, Lgp = O._cbf_pair_terms(x[:, I], x[:, J], xdes[:, I], xdes[:, J], 2, 2 * safety_radius, 1.0, Kcbf, c.M, c.G)
        gp = Lgp[..., 0]                                   # row: -g F_i + g F_j <= h
        cen_e = P[:, 0, 2:5].copy()
        ho = np.zeros((E, D, 4))
        go = np.zeros((E, D, 4))
        for o in range(4):
            xo = np.zeros((E, 1, 9))
            xo[:, 0, 6:9] = spheres[o] + (cen_e * [1, 1, 0] if obs_rel else 0.0)
            h_, Lg_ = O._cbf_pair_terms(x, xo, xdes, xo, 2, safety_radius + r_obs, 1.0, Kcbf, c.M, c.G)
            ho[:, :, o], go[:, :, o] = h_, Lg_[..., 0]
        F = unom[:, 0].reshape(E, D)
        usafe = unom.copy().reshape(E, D, 4)
        st = np.zeros(E, dtype=int)
        need = np.zeros(E, dtype=bool)
        nact = np.zeros(E, dtype=int)
        for e in range(E):
            viol_p = -gp[e] * F[e, I] + gp[e] * F[e, J] - hp[e]
            viol_o = -go[e] * F[e][:, None] - ho[e]
            box = np.abs(F[e]) - umax0
            if viol_p.max() <= 0 and viol_o.max() <= 0 and box.max() <= 0:
                usafe[e, :, 1:] = np.clip(usafe[e, :, 1:], -10, 10)
                continue
            need[e] = True
            m = len(I) + 4 * D + 2 * D
            G = np.zeros((m, D))
            h = np.zeros(m)
            G[np.arange(len(I)), I] = -gp[e]
            G[np.arange(len(I)), J] = gp[e]
            h[:len(I)] = hp[e]
            r0 = len(I)
            for d in range(D):
                for o in range(4):
                    G[r0 + 4 * d + o, d] = -go[e, d, o]
                    h[r0 + 4 * d + o] = ho[e, d, o]
            r0 += 4 * D
            G[r0 + np.arange(D), np.arange(D)] = 1
            G[r0 + D + np.arange(D), np.arange(D)] = -1
            h[r0:] = umax0
            ok, u, lam = O.qp_project(F[e], G, h)
            if ok:
                usafe[e, :, 0] = u
                usafe[e, :, 1:] = np.clip(usafe[e, :, 1:], -10, 10)
                nact[e] = int((lam > 0).sum())
            else:
                st[e] = 1
        hist.append((st.mean(), need.mean(), nact.mean(), nact.max()))
        us = usafe.reshape(n, 4).copy()
        us[:, 0] += c.M * c.G
        rpm = ll.compute_low_level(us, obs, ora.CTRL_TIMESTEP)
        obs = ora.step(rpm)
        t += ora.CTRL_TIMESTEP
    H = np.array(hist)
    o3 = obs.reshape(E, D, 20)
    pz = o3[..., 0:3]
    dmin = min(np.linalg.norm(pz[e][:, None] - pz[e][None, :], axis=-1)[I, J].min() for e in range(E))
    err = np.linalg.norm(o3[..., 0:3] - pos.reshape(E, D, 3), axis=-1)
    if verbose:
        for k in range(0, steps, max(1, steps // 11)):
            print(f"   step {k:4d}: fallback {H[k, 0]:.2f} solver {H[k, 1]:.2f} mean active {H[k, 2]:.2f} max {int(H[k, 3])}")
    w = H[20:]
    print(f"{'' if spheres is None else spheres.tolist()} dz {dz} start {dz_start} omega {omega} obs ({obs_xy},{obs_z}) rel {obs_rel}: window 20..{steps}: fallback mean {w[:, 0].mean():.3f} max {w[:, 0].max():.3f} | "
          f"solver share mean {w[:, 1].mean():.3f} last {H[-1, 1]:.3f} | active rows mean {w[:, 2].mean():.2f} max {int(w[:, 3].max())} | "
          f"min pair distance at end {dmin:.3f} m, tracking error max {err.max():.3f} m, finite {np.isfinite(obs).all()}")
    return H


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=8)
    ap.add_argument("--drones", type=int, default=16)
    ap.add_argument("--steps", type=int, default=220)
    ap.add_argument("--dz", type=float, default=0.3)
    ap.add_argument("--dz-start", type=float, default=None)
    ap.add_argument("--omega", type=float, default=1.5)
    ap.add_argument("--a", type=float, default=1.0)
    ap.add_argument("--offset", type=float, default=5.0)
    ap.add_argument("--obs-xy", type=float, default=0.5)
    ap.add_argument("--obs-z", type=float, default=0.5)
    ap.add_argument("--obs-rel", action="store_true", help="spheres relative to each env's centre instead of world coordinates")
    ap.add_argument("--start-on-traj", action="store_true")
    ap.add_argument("--spheres", default=None, help="x,y,z;x,y,z;x,y,z;x,y,z")
    ap.add_argument("-q", action="store_true")
    a = ap.parse_args()
    run(a.envs, a.drones, a.steps, a.dz, a.dz if a.dz_start is None else a.dz_start, a.omega, a.offset, a.obs_xy, a.obs_z, a.obs_rel,
        a=a.a, verbose=not a.q, start_on_traj=a.start_on_traj,
        spheres=None if a.spheres is None else np.array([[float(v) for v in t.split(",")] for t in a.spheres.split(";")]))
