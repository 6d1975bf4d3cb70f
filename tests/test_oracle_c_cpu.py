"""The plain-C restatement of the reference's control loop (oracle/c_oracle.c) pinned on the same reference-minted fixtures as the
NumPy oracle, and against the NumPy oracle itself: two restatements written separately (closed-form mixer inverse here, scalar
branches instead of masks) must walk the same path.  CPU only; the library is built by `make -C oracle`."""
import os

import numpy as np
import pytest

from oracle import c_oracle as CO
from oracle import np_oracle as O
from tests import helpers as H

G = os.path.join(os.path.dirname(__file__), "golden")


def test_c_oracle_operators_match_the_reference_fixtures():
    g = np.load(os.path.join(G, "geometric_compute.npz"))           # control/geometric.py through the reference's own objects
    rpm = CO.geometric_compute(g["obs"], g["des"])
    np.testing.assert_allclose(rpm, g["rpm"], rtol=1e-12, atol=0)
    assert int(g["n_tilt"]) >= 16                                    # the tilt-clamp branch is in the fixture
    np.testing.assert_allclose(CO.geometric_compute(g["spot_obs"], g["spot_des"])[0], g["spot_rpm"], rtol=1e-12)
    l = np.load(os.path.join(G, "lemniscate.npz"))                   # trajectories/Lemniscate.py
    for i, t in enumerate(l["ts"]):
        np.testing.assert_allclose(CO.lemniscate(t, l["params"]), l["out"][:, i], rtol=0, atol=1e-12)


def test_c_oracle_walks_the_reference_objects_loop():
    """closed_loop_ref_in_loop.npz: the loop whose trajectory sampling and controller are the reference's own objects."""
    d = np.load(os.path.join(G, "closed_loop_ref_in_loop.npz"))
    P, every = d["params"], int(d["every"])
    D = P.shape[0]
    a = CO.AviaryC(d["xyz"], np.zeros((D, 3)))
    obs = a.step(np.zeros((D, 4)))
    np.testing.assert_allclose(obs, d["obs_log"][0], rtol=0, atol=1e-14)
    t = 0.0
    for i in range(int(d["steps"])):
        act = CO.geometric_compute(obs, CO.lemniscate(t, P), a.c)
        obs = a.step(act)
        t += 0.01
        if (i + 1) % every == 0:
            k = (i + 1) // every
            np.testing.assert_allclose(act, d["action_log"][k - 1], rtol=1e-9)
            np.testing.assert_allclose(obs, d["obs_log"][k], rtol=0, atol=1e-8)
    # the same run as ONE call of the C loop (what bench.py times), continued calls included
    b = CO.AviaryC(d["xyz"], np.zeros((D, 3)))
    o1, _ = b.geometric_loop(P, 400)
    t400 = sum([0.01] * 400)                                                 # the loop's own accumulated time (4.000000000000003), not 4.0:
    o2, _ = b.geometric_loop(P, 600, t0=t400, first_zero_step=False)          # this closed loop amplifies 3e-15 s to 5e-10 in the state
    np.testing.assert_allclose(o2, obs, rtol=0, atol=1e-10)
    np.testing.assert_allclose(o1, d["obs_log"][8], rtol=0, atol=1e-8)


@pytest.mark.parametrize("pyb,ctrl", [(100, 100), (240, 48)])
def test_c_oracle_equals_numpy_oracle_over_1000_steps(pyb, ctrl):
    E, D = 16, 8
    xyz, rpy, P = H.c2_setup(E, D, seed=5, phase="c3")
    rpy = np.random.default_rng(0).uniform(-0.2, 0.2, size=rpy.shape)       # non-trivial initial attitudes
    ref, _ = H.oracle_closed_loop(xyz, rpy, P, 1000, pyb_freq=pyb, ctrl_freq=ctrl)
    a = CO.AviaryC(xyz, rpy, pyb, ctrl)
    one, used1 = a.geometric_loop(P, 1000, threads=1)
    np.testing.assert_allclose(one, ref, rtol=0, atol=1e-9)
    b = CO.AviaryC(xyz, rpy, pyb, ctrl)
    four, used4 = b.geometric_loop(P, 1000, threads=4)
    assert used1 == 1 and used4 >= 1
    np.testing.assert_array_equal(one, four)                                 # drones are independent: the thread count changes nothing


def test_c_oracle_step_pieces_match_numpy_oracle():
    """env.step alone on arbitrary states / commands (clipping on both sides, gimbal branches of the observation's Euler angles)."""
    rng = np.random.default_rng(3)
    n = 200
    xyz = rng.normal(size=(n, 3))
    rpy = rng.uniform(-3.1, 3.1, size=(n, 3))
    rpy[:8, 1] = np.pi / 2
    rpy[8:16, 1] = -np.pi / 2
    a, ora = CO.AviaryC(xyz, rpy, 240, 240), O.AviaryOracle(xyz, rpy, pyb_freq=240, ctrl_freq=240)
    for k in range(30):
        act = rng.uniform(-3000, 26000, size=(n, 4))
        np.testing.assert_allclose(a.step(act), ora.step(act), rtol=0, atol=1e-10)


# ---- the CBF-filtered loop (BASELINE config 4) on the C restatement ----------------------------------------------------------
def test_c_oracle_cbf_rows_match_the_reference_minted_rows():
    """cbf_rows_o2.npz holds (G, h) built by the reference's own CBF._build_ineq_const: the C rows must be those rows, in that order."""
    g = np.load(os.path.join(G, "cbf_rows_o2.npz"))
    done = 0
    for k in range(int(g["n_cases"])):
        x, xd, xo, r, Gr, hr = (g[f"c{k}_{n}"] for n in ("x", "xdes", "xobs", "obsr", "G", "h"))
        b = CO.cbf_params(g["Kcbf"], g["umax"], float(g["safety_radius"]), float(g["zscale"]), xo if len(r) else None, list(r) if len(r) else None)
        Gc, hc = CO.cbf_rows(x, xd, b)
        assert Gc.shape == Gr.shape
        np.testing.assert_allclose(Gc, Gr, rtol=1e-12, atol=1e-12 * max(1.0, np.abs(Gr).max()))
        np.testing.assert_allclose(hc, hr, rtol=1e-12, atol=1e-12 * max(1.0, np.abs(hr).max()))
        done += 1
    assert done >= 8


def test_c_oracle_qp_equals_the_numpy_active_set_and_the_converged_interior_point():
    from oracle import cvxopt_qp as CQ
    from tests.test_oracle_cvxopt_cpu import make_qp
    solved = 0
    for seed in range(40):
        Gm, h, uhat = make_qp(seed)
        ok, u, _ = O.qp_project(uhat.reshape(-1), Gm, h)
        okc, uc, it = CO.qp_project(uhat, Gm, h)
        assert ok == okc                                              # the same feasibility decision
        if ok:
            solved += 1
            np.testing.assert_allclose(uc, u, rtol=0, atol=1e-12)
            tight = CQ.coneqp_l(np.eye(uhat.size), -uhat.reshape(-1), Gm, h, maxiters=200, abstol=1e-15, reltol=1e-14, feastol=1e-12)
            np.testing.assert_allclose(uc, tight["x"], rtol=0, atol=1e-8)       # a third algorithm, the same point
        else:
            np.testing.assert_array_equal(uc, uhat.reshape(-1))       # untouched
    assert solved >= 30


def test_c_oracle_cbf_loop_with_the_lqr_omega_nominal_equals_numpy_oracle():
    """simulations/CBFTest.py's default nominal controller (LQROmegaController, :290-293) in the loop."""
    E, D, steps = 3, 8, 150
    xyz, rpy, P = H.c2_setup(E, D, phase="c3", offset=1.0)
    P[..., 4] = 0.5 + 0.3 * np.arange(D)
    xyz[..., 2] = 0.5 + 0.3 * np.arange(D)
    x_obs, obs_r = [np.array([[0.3, 0.2, 0.9], [0, 0, 0]])], [0.1]
    Kcbf, umax = np.array([5.28, 4.6]), np.array([O.CF2P.MAX_THRUST, 10, 10, 10])
    ref, hist = H.oracle_cbf_closed_loop(xyz, rpy, P, steps, Kcbf, umax, 0.1, 1.0, x_obs, obs_r, nominal="lqr_omega")
    got, st, its, _ = CO.CbfLoopC(xyz, rpy, CO.cbf_params(Kcbf, umax, 0.1, 1.0, x_obs, obs_r)).run(P, steps, K_lqr_omega=O.lqr_omega_gain(O.CF2P))
    np.testing.assert_array_equal(st, np.array(hist))
    assert st.sum() > 50 and its > 500                              # the infeasible branch and many iterations are part of it
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-9)


@pytest.mark.parametrize("D,nobs", [(8, 1), (16, 4), (4, 0)])
def test_c_oracle_cbf_loop_equals_numpy_oracle(D, nobs):
    """simulations/CBFTest.py:303-350 on both restatements: statuses equal at every step (infeasible env-steps included), states to 1e-9."""
    E, steps = 3, 120
    xyz, rpy, P = H.c2_setup(E, D, phase="c3", offset=1.0)
    P[..., 4] = 0.5 + 0.3 * np.arange(D)
    xyz[..., 2] = 0.5 + 0.3 * np.arange(D)
    x_obs = [np.array([[0.3 * (-1) ** k, 0.2 * (-1) ** (k // 2), 0.9 + 0.3 * k], [0, 0, 0]]) for k in range(nobs)] or None
    obs_r = [0.1] * nobs if nobs else None
    Kcbf, umax = np.array([5.28, 4.6]), np.array([O.CF2P.MAX_THRUST, 10, 10, 10])
    ref, hist = H.oracle_cbf_closed_loop(xyz, rpy, P, steps, Kcbf, umax, 0.1, 1.0, x_obs, obs_r)
    L = CO.CbfLoopC(xyz, rpy, CO.cbf_params(Kcbf, umax, 0.1, 1.0, x_obs, obs_r))
    got, st, its, _ = L.run(P, steps)
    np.testing.assert_array_equal(st, np.array(hist))
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-9)
    assert its > 0
    L2 = CO.CbfLoopC(xyz, rpy, CO.cbf_params(Kcbf, umax, 0.1, 1.0, x_obs, obs_r))
    got2, st2, _, _ = L2.run(P, steps, threads=3)
    np.testing.assert_array_equal(got2, got)
    np.testing.assert_array_equal(st2, st)


def test_c_oracle_lqr_default_loop_matches_the_reference_fixtures():
    """The scripts' default controller (LQRController on LinearizedModel) on the C restatement: operator against lqr12.npz (true and
    'noisy' gains, min-thrust clips included), the loop with wind against closed_loop_lqr_ref_in_loop.npz (reference objects in the loop)."""
    g = np.load(os.path.join(G, "lqr12.npz"))
    des = np.zeros((g["obs"].shape[0], 11))
    des[:, 0:3], des[:, 3:6], des[:, 9], des[:, 10] = g["pos_d"], g["vel_d"], g["yaw_d"], g["om_d"]
    for tag in ("true", "noisy"):
        np.testing.assert_allclose(CO.lqr12_compute(g["obs"], des, g["K_" + tag]), g["act_" + tag], rtol=1e-12)
    f = np.load(os.path.join(G, "closed_loop_lqr_ref_in_loop.npz"))
    P, every, D = f["params"], int(f["every"]), f["params"].shape[0]
    av = CO.AviaryC(f["xyz"], np.zeros((D, 3)))
    t, first = 0.0, True
    for k in range(1, f["obs_log"].shape[0]):
        obs, _ = CO.lqr_loop(av, P, f["K"], every, wind=f["wind"], t0=t, first_zero_step=first)
        first = False
        t = sum([0.01] * (k * every))
        np.testing.assert_allclose(obs, f["obs_log"][k], rtol=0, atol=1e-7)


def test_c_oracle_order3_rows_and_loop():
    """The order-3 (yank / body-rate) path on the C restatement: rows against the reference-minted (G, h) of cbf_rows_o3.npz (incl. the
    slot quirk of custom_hdots and the column quirk of the thrust-state box), the CBFTestOrd3.py loop against the NumPy oracle."""
    g = np.load(os.path.join(G, "cbf_rows_o3.npz"))
    done = 0
    for k in range(int(g["n_cases"])):
        x, xd, xo, r, Gr, hr = (g[f"c{k}_{n}"] for n in ("x", "xdes", "xobs", "obsr", "G", "h"))
        b = CO.cbf_params(g["Kcbf"], g["umax"], float(g["safety_radius"]), float(g["zscale"]), xo if len(r) else None, list(r) if len(r) else None,
                          order=3, Fmin=float(g["Fmin"]), Fmax=float(g["Fmax"]))
        Gc, hc = CO.cbf_rows(x, xd, b)
        assert Gc.shape == Gr.shape
        np.testing.assert_allclose(Gc, Gr, rtol=1e-12, atol=1e-12 * max(1.0, np.abs(Gr).max()))
        np.testing.assert_allclose(hc, hr, rtol=1e-12, atol=1e-12 * max(1.0, np.abs(hr).max()))
        done += 1
    assert done >= 8
    E, D, steps = 4, 4, 150
    xyz, rpy, P = H.c2_setup(E, D, phase="c3", offset=0.0, omega=0.5)
    xyz[..., 2] = 0.5 + 0.6 * np.arange(D)
    P[..., 4] = 0.5 + 0.6 * np.arange(D)
    x_obs, obs_r = [np.array([[0.0, 0.0, -0.3], [0, 0, 0], [0, 0, 0]])], [0.1]
    c = O.CF2P
    Kcbf, umax = O.place_poles_chain([-3.0, -3.6, -5.6]), np.array([c.MAX_THRUST / 0.01 / 100, 10, 10, 10])
    ref, hist = H.oracle_cbf_closed_loop(xyz, rpy, P, steps, Kcbf, umax, 0.125, 2.0, x_obs, obs_r, nominal="lqr_yank_omega", order=3,
                                         first_rpm=c.HOVER_RPM)
    L = CO.CbfLoopC(xyz, rpy, CO.cbf_params(Kcbf, umax, 0.125, 2.0, x_obs, obs_r, order=3), first_rpm=c.HOVER_RPM)
    got, st, its, _ = L.run3(P, steps, O.lqr_yank_omega_gain(c, 0.01))
    np.testing.assert_array_equal(st, np.array(hist))
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-9)
    assert its > 100
