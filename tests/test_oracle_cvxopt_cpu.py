"""The reference's QP solver, restated (oracle/cvxopt_qp.py: cvxopt 1.3.2's coneqp for linear inequalities, from the published
algorithm -- UNPINNED, cvxopt is absent) against the oracle's exact active-set solver on CBF-QPs of the reference's shape
(cbf/qptracker.py:86-114: P = I, q = -u_hat, dense G, h of cbf/cbf.py:308-367; 8 drones, 4 spheres, 96 rows, 32 variables).

What the numbers say (measured here, asserted with margins):
 * driven to convergence (tolerances 1e-15 / 1e-14 / 1e-12) the interior-point iterates end 1e-10 from the exact solver's point on every
   feasible QP: two unrelated algorithms agree on the unique minimiser;
 * with cvxopt's DEFAULT tolerances (abstol 1e-7, reltol 1e-6, feastol 1e-7) the solve stops up to 4e-4 away from it -- the strong
   convexity bound |u - u*| <= sqrt(2 gap) = 4.5e-4 at gap = abstol: the reference's own answers carry that much solver noise, forty
   times north_star's 1e-5, so agreement of ANY implementation with the reference's C4 controls is only defined to ~1e-4;
 * on an infeasible QP the solver does not raise: it stops at the iteration limit with status 'unknown' and hands back its last
   iterate, which QPTracker._rectify passes on as a success (cbf/qptracker.py:103-112)."""
import numpy as np
import pytest

from oracle import cvxopt_qp as CQ
from oracle import np_oracle as O

C = O.CF2P
KCBF, UMAX = np.array([5.28, 4.6]), np.array([C.MAX_THRUST, 10, 10, 10])          # cbf/cbf.py:119-124 with poles (-2.2, -2.4); :566-572


def make_qp(seed, N=8, spread=0.35, nobs=4):
    """A crowded env of the C4 generator's kind: N drones 0.3 m apart in z within +-spread in x, y, tracking errors, four spheres below."""
    rng = np.random.default_rng(seed)
    x, xdes = np.zeros((N, 9)), np.zeros((N, 9))
    x[:, 6:9] = rng.uniform(-spread, spread, size=(N, 3))
    x[:, 8] = 0.5 + 0.3 * np.arange(N) + rng.normal(size=N) * 0.05
    x[:, 0:3] = rng.normal(size=(N, 3)) * 0.1
    x[:, 3:6] = rng.normal(size=(N, 3)) * 0.5
    xdes[:, 6:9] = x[:, 6:9] + rng.normal(size=(N, 3)) * 0.1
    xdes[:, 3:6] = rng.normal(size=(N, 3)) * 0.3
    x_obs = np.array([[[0.5 * (-1) ** k, 0.5 * (-1) ** (k // 2), -3.0], [0, 0, 0]] for k in range(nobs)])
    G, h = O.cbf_rows(x, xdes, 2, KCBF, UMAX, 0.1, 1.0, C, x_obs=x_obs, obs_r=[0.1] * nobs)
    uhat = np.zeros((N, 4))
    uhat[:, 0] = rng.normal(size=N) * 0.05
    uhat[:, 1:] = rng.normal(size=(N, 3)) * 0.5
    return G, h, uhat


def test_interior_point_on_a_known_answer():
    """Projection of (2, 0) onto {x1 <= 1, x2 >= -5}: (1, 0), multiplier 1 on the first row."""
    sol = CQ.coneqp_l(np.eye(2), -np.array([2.0, 0.0]), np.array([[1.0, 0.0], [0.0, -1.0]]), np.array([1.0, 5.0]))
    assert sol["status"] == "optimal" and sol["iterations"] < 20
    np.testing.assert_allclose(sol["x"], [1.0, 0.0], atol=1e-6)
    np.testing.assert_allclose(sol["z"], [1.0, 0.0], atol=1e-6)


def test_default_tolerances_leave_solver_noise_and_convergence_reaches_the_exact_minimiser():
    worst_default, worst_tight, with_active, n_feas, n_inf = 0.0, 0.0, 0, 0, 0
    for seed in range(40):
        G, h, uhat = make_qp(seed)
        ok, u, lam = O.qp_project(uhat.reshape(-1), G, h)
        success, ui, sol = CQ.rectify(uhat, G, h)
        assert success                                              # the call returned: the reference proceeds with ui
        if not ok:
            n_inf += 1
            assert sol["status"] == "unknown" and sol["iterations"] == CQ.MAXITERS and np.isfinite(ui).all()
            assert sol["primal infeasibility"] > CQ.FEASTOL         # never primal feasible: the rows are inconsistent
            assert np.abs(ui - uhat).max() > 1e-3                   # and what comes back is not the nominal input either
            continue
        n_feas += 1
        with_active += int((lam > 1e-12).any())
        assert sol["status"] == "optimal" and sol["iterations"] < 60
        assert sol["primal infeasibility"] <= CQ.FEASTOL and sol["dual infeasibility"] <= CQ.FEASTOL
        d = float(np.abs(ui.reshape(-1) - u).max())
        assert d <= np.sqrt(2 * max(sol["gap"], CQ.ABSTOL)) * 1.5 + 1e-6
        worst_default = max(worst_default, d)
        tight = CQ.coneqp_l(np.eye(uhat.size), -uhat.reshape(-1), G, h, maxiters=200, abstol=1e-15, reltol=1e-14, feastol=1e-12)
        worst_tight = max(worst_tight, float(np.abs(tight["x"] - u).max()))
    assert n_feas >= 30 and with_active >= 25 and n_inf >= 1        # the sample has what it is meant to have
    assert worst_tight < 1e-8                                       # measured 1.3e-10: both solvers find the unique minimiser
    assert 1e-5 < worst_default < 1e-3                              # measured 4.0e-4: more than north_star's 1e-5, inside sqrt(2 abstol)
