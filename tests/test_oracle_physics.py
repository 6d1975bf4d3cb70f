"""Self-consistency of the [UPSTREAM]-spec parts of the oracle (no reference code exists
for these in the tree; parity against real PyBullet is unpinned -- SURVEY.md 8c)."""
import numpy as np
import pytest
from scipy.integrate import solve_ivp
from scipy.optimize import minimize
from scipy.spatial.transform import Rotation

from oracle import np_oracle as O


def test_quat_euler_roundtrip_and_scipy_agreement():
    rng = np.random.default_rng(0)
    rpy = rng.uniform(-1.4, 1.4, size=(256, 3))
    q = O.quat_from_euler_bullet(rpy)
    np.testing.assert_allclose(q, Rotation.from_euler("xyz", rpy).as_quat(), atol=1e-14)
    np.testing.assert_allclose(O.euler_from_quat_bullet(q), rpy, atol=1e-12)
    np.testing.assert_allclose(O.quat_to_rotmat_bullet(q), Rotation.from_quat(q).as_matrix(), atol=1e-14)
    np.testing.assert_allclose(O.quat_to_rotmat_scipy(3.0 * q), Rotation.from_quat(q).as_matrix(), atol=1e-14)


def test_euler_gimbal_branches():
    for pitch, sign in ((np.pi / 2, 1.0), (-np.pi / 2, -1.0)):
        q = Rotation.from_euler("xyz", [0.0, pitch, 0.3]).as_quat()
        rpy = O.euler_from_quat_bullet(q)
        assert rpy[0] == 0.0 and rpy[1] == sign * np.pi / 2
        np.testing.assert_allclose(Rotation.from_euler("xyz", rpy).as_matrix(), Rotation.from_quat(q).as_matrix(),
                                   atol=1e-7)


def test_integrate_q_is_exact_exponential():
    rng = np.random.default_rng(1)
    q = Rotation.random(64, random_state=2).as_quat()
    w = rng.normal(size=(64, 3)) * 3
    dt = 1 / 240
    qn = O.integrate_q(q, w, dt)
    want = (Rotation.from_quat(q) * Rotation.from_rotvec(w * dt)).as_quat()
    sgn = np.sign(np.sum(qn * want, axis=-1, keepdims=True))
    np.testing.assert_allclose(qn, sgn * want, atol=1e-14)
    np.testing.assert_allclose(O.integrate_q(q, np.zeros((64, 3)), dt), q, atol=0)
    # R(q_new) w == R(q_old) w : the world-frame ang_v may be formed from either rotation
    np.testing.assert_allclose(O.matvec(O.quat_to_rotmat_bullet(qn), w), O.matvec(O.quat_to_rotmat_bullet(q), w),
                               atol=1e-13)


def test_hover_is_equilibrium_and_obs_layout():
    env = O.AviaryOracle(np.array([[0, 0, 1.0], [1, 0, 1.0]]), np.zeros((2, 3)), pyb_freq=240, ctrl_freq=240)
    obs0 = env.reset()
    assert obs0.shape == (2, 20)
    np.testing.assert_allclose(obs0[:, 3:7], [[0, 0, 0, 1]] * 2)
    a = np.full((2, 4), O.CF2P.HOVER_RPM)
    for _ in range(240):
        obs = env.step(a)
    np.testing.assert_allclose(obs[:, 0:3], [[0, 0, 1.0], [1, 0, 1.0]], atol=1e-12)
    np.testing.assert_allclose(obs[:, 16:20], a)
    # clipping to [0, MAX_RPM] is what lands in obs[16:20]
    obs = env.step(np.array([[-5.0, 1e6, 100.0, 200.0]] * 2))
    np.testing.assert_allclose(obs[0, 16:20], [0.0, O.CF2P.MAX_RPM, 100.0, 200.0])
    with pytest.raises(ValueError):
        O.AviaryOracle(np.zeros((1, 3)), np.zeros((1, 3)), pyb_freq=240, ctrl_freq=100)


def test_derived_constants():
    c = O.CF2P
    np.testing.assert_allclose(c.HOVER_RPM, 14468.43, atol=5e-3)
    np.testing.assert_allclose(c.MAX_RPM, 21702.64, atol=5e-3)
    np.testing.assert_allclose(c.MAX_THRUST, 0.59535, atol=1e-6)


def test_free_fall_zero_rpm_first_step():
    # reference loops start with env.step(zeros) (EnvGeometric.py:431)
    env = O.AviaryOracle(np.array([[0, 0, 0.5]]), np.zeros((1, 3)), pyb_freq=100, ctrl_freq=100)
    obs = env.step(np.zeros((1, 4)))
    np.testing.assert_allclose(obs[0, 12], -9.8 * 0.01, atol=1e-15)
    np.testing.assert_allclose(obs[0, 2], 0.5 - 9.8 * 0.01 * 0.01, atol=1e-15)


def test_substeps_equal_repeated_fine_steps():
    rng = np.random.default_rng(3)
    xyz = rng.normal(size=(8, 3))
    rpy = rng.uniform(-0.3, 0.3, size=(8, 3))
    a = O.CF2P.HOVER_RPM * (1 + 0.02 * rng.normal(size=(8, 4)))
    e1 = O.AviaryOracle(xyz, rpy, pyb_freq=240, ctrl_freq=48)
    e2 = O.AviaryOracle(xyz, rpy, pyb_freq=240, ctrl_freq=240)
    o1 = e1.step(a)
    for _ in range(5):
        o2 = e2.step(a)
    np.testing.assert_allclose(o1, o2, atol=1e-14)


def test_rk4_tracks_solve_ivp():
    c = O.CF2P
    rng = np.random.default_rng(4)
    n = 4
    pos = rng.normal(size=(n, 3))
    quat = Rotation.from_euler("xyz", rng.uniform(-0.4, 0.4, size=(n, 3))).as_quat()
    vel = rng.normal(size=(n, 3))
    rates = rng.normal(size=(n, 3))
    rpm = c.HOVER_RPM * (1 + 0.03 * rng.normal(size=(n, 4)))
    dt, steps = 1 / 240, 48
    s = (pos, quat, vel, rates)
    for _ in range(steps):
        out = O.dyn_step_rk4(*s, rpm, dt, c)
        s = out[:4]
    for i in range(n):
        def f(t, y):
            d = O.dyn_derivative(y[0:3], y[3:7], y[7:10], y[10:13], rpm[i], c)
            return np.concatenate(d)
        y0 = np.concatenate([pos[i], quat[i], vel[i], rates[i]])
        sol = solve_ivp(f, [0, dt * steps], y0, rtol=1e-12, atol=1e-14, method="DOP853")
        y = sol.y[:, -1]
        y[3:7] /= np.linalg.norm(y[3:7])
        got = np.concatenate([s[0][i], s[1][i], s[2][i], s[3][i]])
        np.testing.assert_allclose(got, y, atol=2e-9)


def test_euler_converges_to_rk4_first_order():
    c = O.CF2P
    pos = np.zeros((1, 3)); quat = np.array([[0, 0, 0, 1.0]]); vel = np.zeros((1, 3)); rates = np.array([[0.3, -0.2, 0.1]])
    rpm = c.HOVER_RPM * np.array([[1.01, 0.99, 1.02, 0.98]])
    def run(stepf, dt, T=0.1):
        s = (pos, quat, vel, rates)
        for _ in range(int(round(T / dt))):
            s = stepf(*s, rpm, dt, c)[:4]
        return np.concatenate([x[0] for x in s])
    ref = run(O.dyn_step_rk4, 1e-4)
    e1 = np.abs(run(O.dyn_step_euler, 1e-3) - ref).max()
    e2 = np.abs(run(O.dyn_step_euler, 5e-4) - ref).max()
    assert 1.6 < e1 / e2 < 2.4


def test_drag_opposes_velocity_and_uses_previous_rpm():
    xyz = np.array([[0, 0, 1.0]])
    env = O.AviaryOracle(xyz, np.zeros((1, 3)), pyb_freq=240, ctrl_freq=240, physics="dyn_drag")
    env.vel[:] = [[1.0, 0, 0]]
    a = np.full((1, 4), O.CF2P.HOVER_RPM)
    o1 = env.step(a)          # previous action is zero -> no drag yet
    np.testing.assert_allclose(o1[0, 10], 1.0, atol=1e-15)
    o2 = env.step(a)
    want = 1.0 - (1 / 240) * O.CF2P.DRAG[0] * 4 * (2 * np.pi * O.CF2P.HOVER_RPM / 60) / O.CF2P.M
    np.testing.assert_allclose(o2[0, 10], want, atol=1e-15)


def _qp_ref(uhat, G, h):
    res = minimize(lambda u: 0.5 * np.sum((u - uhat) ** 2), uhat, jac=lambda u: u - uhat,
                   constraints=[{"type": "ineq", "fun": lambda u: h - G @ u, "jac": lambda u: -G}], method="SLSQP",
                   options=dict(ftol=1e-14, maxiter=500))
    return res


def test_qp_project_matches_slsqp_and_kkt():
    rng = np.random.default_rng(5)
    for trial in range(20):
        n, m = 8, 14
        G = rng.normal(size=(m, n))
        u_feas = rng.normal(size=n)
        h = G @ u_feas + rng.uniform(0.0, 1.0, size=m)
        uhat = u_feas + rng.normal(size=n) * 2
        ok, u, lam = O.qp_project(uhat, G, h)
        assert ok
        assert (G @ u - h).max() < 1e-8
        assert lam.min() >= -1e-12
        np.testing.assert_allclose(u - uhat + G.T @ lam, 0, atol=1e-9)      # stationarity
        np.testing.assert_allclose(lam * (G @ u - h), 0, atol=1e-8)         # complementarity
        res = _qp_ref(uhat, G, h)
        if res.success:
            np.testing.assert_allclose(u, res.x, atol=1e-5)


def test_qp_infeasible_falls_back_to_nominal():
    G = np.array([[1.0, 0.0], [-1.0, 0.0]])
    h = np.array([-1.0, -1.0])          # x <= -1 and x >= 1
    ok, u, _ = O.qp_project(np.zeros(2), G, h)
    assert not ok
    x = np.zeros((2, 9)); x[0, 6:] = [0, 0, 0.5]; x[1, 6:] = [0.05, 0, 0.5]
    u_nom = np.zeros((2, 4))
    u, status = O.cbf_filter(x, x.copy(), u_nom, 2, [5.28, 4.6], np.array([1e-9, 10, 10, 10]), 0.1, 1.0)
    assert status in (0, 1)
    if status == 1:
        np.testing.assert_array_equal(u, u_nom)


def test_cbf_filter_on_golden_rows_is_feasible_or_fallback():
    import os
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "cbf_rows_o2.npz"))
    rng = np.random.default_rng(6)
    for k in range(int(d["n_cases"])):
        G, h = d[f"c{k}_G"], d[f"c{k}_h"]
        N = G.shape[1] // 4
        uhat = np.concatenate([rng.normal(size=(N, 1)) * 0.05, rng.normal(size=(N, 3)) * 2], axis=1).reshape(-1)
        ok, u, lam = O.qp_project(uhat, G, h)
        if ok:
            assert (G @ u - h).max() < 1e-7 * max(1.0, np.abs(h).max())
            np.testing.assert_allclose(u - uhat + G.T @ lam, 0, atol=1e-8)
