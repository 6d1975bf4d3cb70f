"""TEST INFRASTRUCTURE ONLY -- a restatement of the third-party solver the reference calls for the CBF-QP:
``cvxopt.solvers.qp(P, q, G, h)`` of **cvxopt 1.3.2** (environment.yaml:52; call site cbf/qptracker.py:106), i.e. ``coneqp`` with linear
inequality constraints only (dims = {'l': m, 'q': [], 's': []}, no equality constraints) and its default options
(maxiters 100, abstol 1e-7, reltol 1e-6, feastol 1e-7, no iterative refinement for a pure 'l' cone).

cvxopt is NOT installed here and cannot be installed (no network): this file follows the PUBLISHED algorithm -- L. Vandenberghe,
"The CVXOPT linear and quadratic cone program solvers" (2010), sections 4-7: primal-dual path following with Nesterov-Todd
scaling (for the 'l' cone W = diag(sqrt(s / z)), lambda = sqrt(s o z)), Mehrotra predictor-corrector, step length 0.99 of the way
to the boundary -- in the form ``coneqp`` implements it (initial point from the KKT system with W = I and a shift into the cone;
sigma = min(1, max(0, 1 - step + <ds_a, dz_a> / gap * step^2))^3; stopping rule on (pres, dres, gap, relgap)).  **UNPINNED**: no
output of the real cvxopt is available to check it against.  What it is used for (tests/test_oracle_cvxopt_cpu.py,
tests/tools/c4_cvxopt_probe.py):

 * on FEASIBLE QPs the minimiser is unique, and any faithful interior-point solve must end within its stopping tolerances of it:
   this restatement measures how far that is for the reference's problem scaling (thrust in newtons, |u - u*| ~ 1e-9 .. 1e-7), i.e.
   how much of north_star's 1e-5 the reference's OWN solver noise may use, and shows that the exact active-set solver of
   np_oracle.qp_project / the HIP kernels returns that point;
 * on INFEASIBLE QPs it shows what DESIGN.md section 4 (C4, round 3) says in words: the solver does not raise, it stops at the iteration
   limit with status 'unknown' and returns its last iterate -- which the reference then applies (cbf/qptracker.py:103-112 sets
   success = True as soon as the call returns).  The iterate itself depends on every detail of the implementation and is not
   claimed to be cvxopt's.

Nothing in multidronesim_amd/ imports this file."""
from __future__ import annotations

import numpy as np

MAXITERS, ABSTOL, RELTOL, FEASTOL = 100, 1e-7, 1e-6, 1e-7
STEP, EXPON = 0.99, 3


def coneqp_l(P, q, G, h, maxiters=MAXITERS, abstol=ABSTOL, reltol=RELTOL, feastol=FEASTOL):
    """min 1/2 x'Px + q'x  s.t.  G x <= h.  Returns a dict like cvxopt's: status ('optimal' | 'unknown'), x, s, z, gap,
    relative gap, primal / dual infeasibility, iterations."""
    P = np.asarray(P, dtype=np.float64)
    q = np.asarray(q, dtype=np.float64).reshape(-1)
    G = np.asarray(G, dtype=np.float64)
    h = np.asarray(h, dtype=np.float64).reshape(-1)
    m = h.shape[0]
    resx0, resz0 = max(1.0, np.linalg.norm(q)), max(1.0, np.linalg.norm(h))

    def kkt(d2):
        """Solver of  P ux + G' uz = bx,  G ux - diag(d2) uz = bz  (d2 = s / z = W^2; W = I for the initial point)."""
        H = P + G.T @ (G / d2[:, None])
        L = np.linalg.cholesky(H)

        def solve(bx, bz):
            ux = np.linalg.solve(L.T, np.linalg.solve(L, bx + G.T @ (bz / d2)))
            return ux, (G @ ux - bz) / d2
        return solve

    # ---- initial point: [P G'; G -I] [x; z] = [-q; h], s = -z, both shifted into the cone when they are not inside it
    x, z = kkt(np.ones(m))(-q, h)
    s = -z
    ts = np.max(-s)
    if ts >= -1e-8 * max(np.linalg.norm(s), 1.0):
        s = s + (1.0 + ts)
    tz = np.max(-z)
    if tz >= -1e-8 * max(np.linalg.norm(z), 1.0):
        z = z + (1.0 + tz)
    gap = float(s @ z)

    for iters in range(maxiters + 1):
        f0 = 0.5 * x @ (P @ x) + q @ x
        rx = P @ x + q + G.T @ z
        rz = s + G @ x - h
        resx, resz = np.linalg.norm(rx), np.linalg.norm(rz)
        pcost = f0
        dcost = f0 + z @ rz - gap
        relgap = gap / -pcost if pcost < 0.0 else (gap / dcost if dcost > 0.0 else None)
        pres, dres = resz / resz0, resx / resx0
        done = pres <= feastol and dres <= feastol and (gap <= abstol or (relgap is not None and relgap <= reltol))
        if done or iters == maxiters:
            return {"status": "optimal" if done else "unknown", "x": x, "s": s, "z": z, "gap": gap, "relative gap": relgap,
                    "primal infeasibility": pres, "dual infeasibility": dres, "iterations": iters}
        try:
            solve = kkt(s / z)
        except np.linalg.LinAlgError:              # cvxopt: "Terminated (singular KKT matrix)" after iteration 0
            return {"status": "unknown", "x": x, "s": s, "z": z, "gap": gap, "relative gap": relgap, "primal infeasibility": pres,
                    "dual infeasibility": dres, "iterations": iters}
        mu = gap / m
        sigma, dsdz_corr = 0.0, np.zeros(m)
        for i in (0, 1):
            # P dx + G' dz = -rx;  G dx + ds = -rz;  z o ds + s o dz = -s o z - dsa o dza + sigma mu e
            # (the scaled system of the paper with lambda o (W^-1 ds + W dz) on the left, multiplied through by lambda)
            comp = -s * z - dsdz_corr + sigma * mu
            # eliminate ds = -rz - G dx:  s o dz - z o (G dx) = comp + z o rz  ->  G dx - (s / z) dz = -rz - comp / z
            dx, dz = solve(-rx, -rz - comp / z)
            ds = -rz - G @ dx
            ts, tz = np.max(-ds / s), np.max(-dz / z)
            t = max(0.0, ts, tz)
            if i == 0:
                step = 1.0 if t == 0.0 else min(1.0, 1.0 / t)
                dsdz = float(ds @ dz)
                sigma = min(1.0, max(0.0, 1.0 - step + dsdz / gap * step ** 2)) ** EXPON
                dsdz_corr = ds * dz
            else:
                step = 1.0 if t == 0.0 else min(1.0, STEP / t)
        x = x + step * dx
        s = s + step * ds
        z = z + step * dz
        gap = float(s @ z)
    raise AssertionError("unreachable")


def rectify(uhat, G, h):
    """QPTracker._rectify (cbf/qptracker.py:86-114) with this solver in cvxopt's place: P = I, q = -uhat -> (success, u, info).
    success is True whenever the solver RETURNS (as in the reference): also with status 'unknown'."""
    uhat = np.asarray(uhat, dtype=np.float64)
    n = uhat.size
    sol = coneqp_l(np.eye(n), -uhat.reshape(-1), G, h)
    return True, sol["x"].reshape(uhat.shape), sol
