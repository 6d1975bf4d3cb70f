"""Pin the float64 oracle against golden vectors minted from the reference's own NumPy
code (tests/golden/mint_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import np_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name))


def test_lemniscate_matches_reference():
    d = load("lemniscate.npz")
    P, ts, ref = d["params"], d["ts"], d["out"]
    for k in range(P.shape[0]):
        for i, t in enumerate(ts):
            pos, vel, acc, yaw, om = O.lemniscate(t, P[k, 0], P[k, 1], P[k, 2:5], P[k, 5], P[k, 6])
            got = np.hstack([pos, vel, acc, yaw, om])
            np.testing.assert_allclose(got, ref[k, i], rtol=0, atol=1e-12)
    # SURVEY.md 8c spot value
    pos, vel, acc, yaw, om = O.lemniscate(0.37, 1.0, 1.5, np.array([0, 0, .5]), 0.3, 0.0)
    np.testing.assert_allclose(np.hstack([pos, vel, acc, yaw, om]), d["spot"], atol=1e-14)
    np.testing.assert_allclose(pos, [0.3505205637, 0.6651959776, 0.5], atol=1e-9)


def test_lemniscate_batched_params():
    d = load("lemniscate.npz")
    P, ts, ref = d["params"], d["ts"], d["out"]
    pos, vel, acc, yaw, om = O.lemniscate(ts[37], P[:, 0], P[:, 1], P[:, 2:5], P[:, 5], P[:, 6])
    got = np.concatenate([pos, vel, acc, yaw[:, None], om[:, None]], axis=-1)
    np.testing.assert_allclose(got, ref[:, 37], atol=1e-12)


def test_geometric_compute_matches_reference():
    d = load("geometric_compute.npz")
    obs, des = d["obs"], d["des"]
    rpm = O.geometric_compute(obs, des[:, 0:3], des[:, 3:6], des[:, 6:9], des[:, 9], des[:, 10])
    np.testing.assert_allclose(rpm, d["rpm"], rtol=1e-11, atol=1e-7)
    f, w, Rd = O.geometric_compute(obs, des[:, 0:3], des[:, 3:6], des[:, 6:9], des[:, 9], des[:, 10],
                                   return_omegas=True)
    np.testing.assert_allclose(f, d["force"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(w, d["w_des"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(Rd, d["R_des"], rtol=0, atol=1e-12)
    # both quirk-sensitive spot values from SURVEY.md 8c
    so, sd = d["spot_obs"], d["spot_des"]
    r1 = O.geometric_compute(so, sd[0:3], sd[3:6], sd[6:9], sd[9], sd[10])
    np.testing.assert_allclose(r1, [14426.96568, 14788.12966, 11884.78050, 16030.20029], atol=1e-4)
    _, w1, _ = O.geometric_compute(so, sd[0:3], sd[3:6], sd[6:9], sd[9], sd[10], return_omegas=True)
    np.testing.assert_allclose(w1, [0.5956908439, -0.0106877434, 0.7451911107], atol=1e-9)


def test_geometric_golden_covers_clamp_and_clip():
    d = load("geometric_compute.npz")
    rpm = d["rpm"]
    lo = 9440.3
    assert (np.abs(rpm - lo) < 1e-6).any(axis=1).sum() >= 16          # min-thrust clip exercised
    assert (np.abs(rpm - lo) > 1.0).all(axis=1).sum() >= 64           # and plenty of unclipped rows
    # tilt clamp rows: recompute the pre-clamp tilt with the oracle pieces
    obs, des = d["obs"], d["des"]
    n_rand, n_tilt = int(d["n_rand"]), int(d["n_tilt"])
    f, _, Rd = O.geometric_compute(obs[n_rand:n_rand + n_tilt], des[n_rand:n_rand + n_tilt, 0:3],
                                   des[n_rand:n_rand + n_tilt, 3:6], des[n_rand:n_rand + n_tilt, 6:9],
                                   des[n_rand:n_rand + n_tilt, 9], des[n_rand:n_rand + n_tilt, 10], return_omegas=True)
    tilt = np.arccos(np.abs(Rd[:, 2, 2]))
    assert (np.abs(tilt - 40 * np.pi / 180) < 1e-9).sum() >= 16      # clamped exactly to 40 deg


def test_mixer_matches_reference():
    d = load("mixer.npz")
    np.testing.assert_allclose(O.input_to_action(d["u"], O.CF2P), d["rpm"], rtol=1e-12, atol=1e-8)
    np.testing.assert_allclose(O.action_to_input(d["act"], O.CF2P), d["u_back"], rtol=1e-13, atol=1e-18)
    np.testing.assert_allclose(O.action_to_input(d["act"], O.CF2P, cap_rpm=False), d["u_back_nocap"], rtol=1e-13,
                               atol=1e-18)


def test_obs_conversions_match_reference():
    d = load("mixer.npz")
    obs = d["obs"]
    np.testing.assert_allclose(O.obs_to_lin_model(obs, 9), d["lin9"], atol=0)
    np.testing.assert_allclose(O.obs_to_lin_model(obs, 10), d["lin10"], rtol=1e-14)
    np.testing.assert_allclose(O.obs_to_lin_model(obs, 12), d["lin12"], atol=0)
    p, R, v, w = O.obs_to_geo_model(obs)
    got = np.concatenate([p, R.reshape(-1, 9), v, w], axis=-1)
    np.testing.assert_allclose(got, d["geo18"], atol=1e-15)
    with pytest.raises(ValueError):
        O.obs_to_lin_model(obs, 7)


def test_quadrotor_dynamics_matches_reference():
    d = load("dynamics_deriv.npz")
    np.testing.assert_allclose(O.quadrotor_dynamics(d["state"], d["u"]), d["out_hb"], rtol=1e-13, atol=1e-13)
    # stale-J quirk: after load_env_params J is still the Hummingbird one
    np.testing.assert_allclose(d["env_J"], [1.05, 1.05, 2.05])
    got = O.quadrotor_dynamics(d["state"], d["u_env"], m=float(d["env_m"]), g=float(d["env_g"]), J=d["env_J"])
    np.testing.assert_allclose(got, d["out_env"], rtol=1e-13, atol=1e-13)
    assert bool(d["step_raises"])  # QuadrotorDynamics.step() raises ValueError in the reference


@pytest.mark.parametrize("order", [2, 3])
def test_cbf_rows_match_reference(order):
    d = load(f"cbf_rows_o{order}.npz")
    Kcbf = d["Kcbf"]
    np.testing.assert_allclose(O.place_poles_chain(d["poles"]), Kcbf, rtol=1e-10)
    assert bool(d["raises_when_nobs_gt_n"])
    for k in range(int(d["n_cases"])):
        x, xdes, xobs, obsr = d[f"c{k}_x"], d[f"c{k}_xdes"], d[f"c{k}_xobs"], d[f"c{k}_obsr"]
        Gr, hr = d[f"c{k}_G"], d[f"c{k}_h"]
        G, h = O.cbf_rows(x, xdes, order, Kcbf, d["umax"], float(d["safety_radius"]), float(d["zscale"]), O.CF2P,
                          xobs if len(obsr) else None, list(obsr) if len(obsr) else None,
                          Fmin=float(d["Fmin"]), Fmax=float(d["Fmax"]))
        assert G.shape == Gr.shape and h.shape == hr.shape
        np.testing.assert_allclose(G, Gr, rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(h, hr, rtol=1e-10, atol=1e-9)


def test_place_poles_known_values():
    np.testing.assert_allclose(O.place_poles_chain([-2.2, -2.4]), [5.28, 4.6], rtol=1e-12)
    np.testing.assert_allclose(O.place_poles_chain([-3.0, -3.6, -5.6]), [60.48, 47.76, 12.2], rtol=1e-12)
    np.testing.assert_allclose(O.place_poles_chain([-2.2, -2.4, -2.6]), [13.728, 17.24, 7.2], rtol=1e-12)


@pytest.mark.parametrize("model", ["cf2p", "cf2x"])
def test_thrust_omega_matches_reference(model):
    """control/low_level/thrust_omega_ctrl.py (fixture minted with a stubbed [UPSTREAM] BaseControl)."""
    d = load("thrust_omega.npz")
    assert str(d["base_class"]) == "stubbed"
    u, cur, rpm = d[f"{model}_u"], d[f"{model}_cur"], d[f"{model}_rpm"]
    ctl = O.ThrustOmegaOracle(u.shape[1], O.CF2P if model == "cf2p" else O.CF2X)
    for t in range(u.shape[0]):
        got = ctl.compute_from_input(u[t], float(d["dt"]), cur[t])
        np.testing.assert_allclose(got, rpm[t], rtol=1e-13, atol=1e-9)
    np.testing.assert_allclose(ctl.integral, d[f"{model}_integral"], atol=1e-14)
    lo, hi = 0.2685 * 20000 + 4070.3, 0.2685 * 65535 + 4070.3
    assert (np.abs(rpm - lo) < 1e-9).sum() > 10                      # MIN_PWM clip exercised (MAX_PWM is barely reachable: torque clip 3200)


def test_lqr_omega_matches_reference():
    """control/lqr/lqr_omega_controller.py: ARE gain and compute(obs, skip_low_level=True) incl. cap_u."""
    d = load("lqr_omega.npz")
    K = O.lqr_omega_gain(O.CF2P)
    np.testing.assert_allclose(K, d["K"], rtol=1e-8, atol=1e-10)
    u = O.lqr_omega_compute(d["obs"], d["pos_d"], d["vel_d"], d["yaw_d"], d["K"])
    np.testing.assert_allclose(u, d["u"], rtol=1e-10, atol=1e-10)
    lo, hi = 4 * 9440.3 ** 2 * O.CF2P.KF, O.CF2P.MAX_THRUST
    assert (np.abs(d["u"][:, 0] - lo) < 1e-12).sum() >= 8 and (np.abs(d["u"][:, 0] - hi) < 1e-12).sum() >= 8


def test_lqr_yank_omega_matches_reference():
    """control/lqr/lqr_YO_controller.py + control/low_level/yank_omega_ctrl.py: ARE gain, u = -K e with the thrust state
    from calc_z_thrust(obs), and the stateful low level (thrust = cur_thrust + yank dt -> ThrustOmega PID)."""
    d = load("lqr_yank_omega.npz")
    assert str(d["low_level_base_class"]) == "stubbed"
    K = O.lqr_yank_omega_gain(O.CF2P, float(d["dt"]))
    np.testing.assert_allclose(K, d["K"], rtol=1e-8, atol=1e-10)
    low = O.YankOmegaOracle(d["obs"].shape[1], O.CF2P)
    for t in range(d["obs"].shape[0]):
        u = O.lqr_yank_omega_compute(d["obs"][t], d["pos_d"][t], d["vel_d"][t], d["yaw_d"][t], d["K"])
        np.testing.assert_allclose(u, d["u"][t], rtol=1e-10, atol=1e-10)
        rpm = low.compute_low_level(d["u"][t], d["obs"][t], float(d["dt"]))
        np.testing.assert_allclose(rpm, d["rpm"][t], rtol=1e-13, atol=1e-9)


def test_lqr12_matches_reference():
    """control/lqr/lqr_controller.py + model/linearized.py (EnvGeometric.py's default 'lqr'): both gain matrices, u and the RPM."""
    d = load("lqr12.npz")
    for tag, noisy in (("true", False), ("noisy", True)):
        K = O.lqr12_gain(O.CF2P, noisy)
        np.testing.assert_allclose(K, d["K_" + tag], rtol=1e-7, atol=1e-9)
        act, u = O.lqr12_compute(d["obs"], d["pos_d"], d["vel_d"], d["yaw_d"], d["om_d"], d["K_" + tag])
        np.testing.assert_allclose(u, d["u_" + tag], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(act, d["act_" + tag], rtol=1e-10, atol=1e-7)
    assert (np.abs(d["act_true"] - 9440.3) < 1e-6).sum() > 50 and (np.abs(d["act_true"] - 9440.3) > 1).sum() > 300


def test_trajectory_family_matches_reference():
    """trajectories/{Circle,LineTrajectory,CompoundTrajectory,RotateTrajectory}.py via oracle/np_trajectories.py."""
    import types
    from oracle import np_trajectories as NT
    from tests.golden.mint_golden import trajectory_cases
    d = load("trajectories.npz")
    cases = trajectory_cases(types.SimpleNamespace(Lemniscate=NT.Lemniscate, Circle=NT.Circle, Line=NT.Line, Wait=NT.Wait,
                                                   Compound=NT.Compound, Rotate=NT.Rotate))
    assert list(d["names"]) == list(cases)
    for name, tr in cases.items():
        np.testing.assert_allclose(tr.get_total_time(), float(d[name + "_total"]), rtol=1e-14)
        for t, want in zip(d[name + "_t"], d[name + "_out"]):
            pos, vel, acc, yaw, om = tr(float(t))
            got = np.hstack([pos, vel, np.asarray(acc) * np.ones(3), yaw, om])
            np.testing.assert_allclose(got, want, rtol=0, atol=1e-12, err_msg=f"{name} t={t}")


def test_oracle_reproduces_its_own_1000_step_rollouts():
    """SURVEY 8c G7 (oracle-only fixture): the restated Physics.DYN must keep producing the committed 1000-step rollouts
    (Euler, RK4, Euler + drag; 64 drones, 240 Hz, clipped commands included)."""
    from tests.golden import mint_oracle_rollouts as R
    d = np.load(os.path.join(G, "dyn_rollouts_1000.npz"))
    for name, (ph, integ) in dict(euler=("dyn", "euler"), rk4=("dyn", "rk4"), drag=("dyn_drag", "euler")).items():
        r = R.rollout(ph, integ)
        for j, k in enumerate(d["check"]):
            np.testing.assert_allclose(r[int(k)], d[name][j], rtol=1e-12, atol=1e-12, err_msg=f"{name} step {k}")


def test_dyn_wrench_and_accelerations_match_the_reference_tree():
    """a3 pinned where the reference tree restates it: RPM -> (thrust, torques) against utils/model_conversions.py:69-83
    (action_to_input) and (v_dot, w_dot) against model/dynamics.py:83-106 with the env's m, g, J.  What stays spec-level
    ([UPSTREAM]-only) in a1-a4 after this and test_integrate_q_is_the_flow_...: the update ORDER (v, w first; p with the new v,
    q with the new w) and the pybullet euler / quaternion conversions (checked against scipy's, which the reference's
    controllers apply to the same observation: test_oracle_physics.py)."""
    d = load("dyn_wrench_accel.npz")
    c = O.CF2P
    assert abs(c.MAX_RPM - float(d["max_rpm"])) < 1e-9 and c.M == float(d["m"]) and c.G == float(d["g"])
    np.testing.assert_allclose(np.asarray(c.J), d["J"], rtol=0, atol=0)
    clipped = np.clip(d["rpm"], 0, c.MAX_RPM)
    thrust, tau = O.rotor_wrench(clipped, c)
    np.testing.assert_allclose(thrust, d["u"][:, 0], rtol=1e-13, atol=1e-16)
    np.testing.assert_allclose(tau, d["u"][:, 1:], rtol=1e-12, atol=1e-18)
    # continuous-time derivative used by the RK4 integrator
    _, _, acc, wdot = O.dyn_derivative(d["pos"], d["quat"], d["vel"], d["rates"], clipped, c)
    np.testing.assert_allclose(acc, d["v_dot"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(wdot, d["w_dot"], rtol=1e-12, atol=1e-9)
    # one explicit-Euler substep: (v_new - v) / dt and (w_new - w) / dt are exactly those accelerations
    for dt in (1 / 240, 1 / 100):
        _, _, v1, w1, _ = O.dyn_step_euler(d["pos"], d["quat"], d["vel"], d["rates"], clipped, dt, c)
        np.testing.assert_allclose((v1 - d["vel"]) / dt, d["v_dot"], rtol=0, atol=1e-10)
        np.testing.assert_allclose((w1 - d["rates"]) / dt, d["w_dot"], rtol=1e-10, atol=1e-7)
    # AviaryOracle.step clips like action_to_input(cap_rpm=True) does
    ora = O.AviaryOracle(d["pos"], np.zeros_like(d["pos"]), c, 240, 240)
    ora.quat, ora.vel, ora.rates = d["quat"].copy(), d["vel"].copy(), d["rates"].copy()
    obs = ora.step(d["rpm"])
    np.testing.assert_allclose((obs[:, 10:13] - d["vel"]) * 240, d["v_dot"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(obs[:, 16:20], clipped, rtol=0, atol=0)


def test_integrate_q_is_the_flow_of_the_reference_attitude_kinematics():
    """[UPSTREAM] _integrateQ pinned where the reference tree states the kinematics: body-frame rates, R_dot = R hat(w)
    (model/dynamics.py:62-66, :102).  tests/golden/attitude_flow.npz holds R expm(hat(w) dt) built from the reference's own hat_map
    for 256 random (q, w, dt), |w| dt from 0 to 44 rad; the oracle's quaternion update must land on that rotation, and its
    finite difference on the reference's R_dot."""
    d = load("attitude_flow.npz")
    qn = O.integrate_q(d["quat"], d["w"], d["dt"])
    np.testing.assert_allclose(np.linalg.norm(qn, axis=1), 1.0, atol=1e-14)
    np.testing.assert_allclose(O.quat_to_rotmat_scipy(qn), d["R_next"], rtol=0, atol=2e-13)
    still = np.linalg.norm(d["w"], axis=1) == 0
    assert still.sum() >= 5 and np.array_equal(qn[still], d["quat"][still])
    slow = np.linalg.norm(d["w"], axis=1) < 40.0             # central difference: error ~ h^2 |w|^3 / 6
    h = 1e-6
    q0, w0 = d["quat"][slow], d["w"][slow]
    dR = (O.quat_to_rotmat_scipy(O.integrate_q(q0, w0, h)) - O.quat_to_rotmat_scipy(O.integrate_q(q0, w0, -h))) / (2 * h)
    assert slow.sum() >= 128
    np.testing.assert_allclose(dR, d["R_dot"][slow], rtol=0, atol=1e-6)


def test_closed_loop_with_the_reference_objects_in_the_loop():
    """tests/golden/closed_loop_ref_in_loop.npz: 1000 control steps of the EnvGeometric.py loop in which the trajectory sampling and
    the controller are the REFERENCE's own Lemniscate / GeometricControl objects (per drone, reference call order) and only env.step is
    the oracle's DYN step.  The oracle's vectorised loop -- its own lemniscate() and geometric_compute() -- must walk the same path."""
    d = load("closed_loop_ref_in_loop.npz")
    P, every = d["params"], int(d["every"])
    D = P.shape[0]
    ora = O.AviaryOracle(d["xyz"], np.zeros((D, 3)), O.CF2P, 100, 100)
    obs = ora.step(np.zeros((D, 4)))
    np.testing.assert_allclose(obs, d["obs_log"][0], rtol=0, atol=1e-14)
    t = 0.0
    for i in range(int(d["steps"])):
        pos, vel, acc, yaw, yd = O.lemniscate(t, P[:, 0], P[:, 1], P[:, 2:5], P[:, 5], P[:, 6])
        act = O.geometric_compute(obs, pos, vel, acc, yaw, yd)
        obs = ora.step(act)
        t += 0.01
        if (i + 1) % every == 0:
            k = (i + 1) // every
            np.testing.assert_allclose(act, d["action_log"][k - 1], rtol=1e-9)
            np.testing.assert_allclose(obs, d["obs_log"][k], rtol=0, atol=1e-8)


def test_closed_loop_lqr_with_the_reference_objects_in_the_loop():
    """tests/golden/closed_loop_lqr_ref_in_loop.npz: the default 'lqr' loop of simulations/EnvGeometric.py (reference LQRController on
    LinearizedModel, reference Lemniscate, wind on from the first control step), 600 steps; the oracle's lqr12 loop walks the same path."""
    d = load("closed_loop_lqr_ref_in_loop.npz")
    P, every = d["params"], int(d["every"])
    D = P.shape[0]
    K = O.lqr12_gain(O.CF2P)
    np.testing.assert_allclose(K, d["K"], rtol=1e-8, atol=1e-10)
    ora = O.AviaryOracle(d["xyz"], np.zeros((D, 3)), O.CF2P, 100, 100)
    obs = ora.step(np.zeros((D, 4)))
    ora.wind = d["wind"]
    t = 0.0
    for i in range(int(d["steps"])):
        pos, vel, acc, yaw, yd = O.lemniscate(t, P[:, 0], P[:, 1], P[:, 2:5], P[:, 5], P[:, 6])
        act, _ = O.lqr12_compute(obs, pos, vel, yaw, yd, K)
        obs = ora.step(act)
        t += 0.01
        if (i + 1) % every == 0:
            k = (i + 1) // every
            np.testing.assert_allclose(act, d["action_log"][k - 1], rtol=1e-7)
            np.testing.assert_allclose(obs, d["obs_log"][k], rtol=0, atol=1e-7)


def test_euler_convention_is_the_reference_trees():
    """obs[7:10] as the build fills it (pybullet's getEulerFromQuaternion, restated: oracle.euler_from_quat_bullet) read back through the
    reference tree's own convention must give the attitude of obs[3:7]: the fixture holds, minted from the tree, the rotation
    obs_to_geo_model assigns to each quaternion (utils/model_conversions.py:108-113), rpy_to_rot(rpy) (:4-19, R = Rz Ry Rx) and scipy
    'xyz' as control/lqr/lqr_omega_controller.py:97-101 applies it.  Outside pybullet's gimbal branches (|sin pitch| < 0.99999) the
    three agree to 1e-12; inside them pybullet returns roll = 0, yaw = 2 atan2(+-x, -+y), exact only AT pitch = +-pi/2: the attitude error
    is bounded by acos(0.99999) = 4.47e-3 rad there (upstream behaviour, kept)."""
    d = np.load(os.path.join(G, "euler_convention.npz"))
    q, g = d["quat"], d["gimbal"]
    rpy = O.euler_from_quat_bullet(q)
    np.testing.assert_allclose(rpy, d["rpy"], rtol=0, atol=1e-13)                       # the restatement has not moved since minting
    assert g.sum() >= 48 and (~g).sum() >= 200 and (rpy[g][:, 0] == 0).all()            # both regimes covered; roll = 0 in the branches
    assert np.abs(np.abs(rpy[g][:, 1]) - np.pi / 2).max() == 0.0
    np.testing.assert_allclose(d["R_rpy"], d["R_rpy_scipy"], rtol=0, atol=1e-14)         # rpy_to_rot == scipy 'xyz' (the tree's two readings)
    np.testing.assert_allclose(d["R_rpy"][~g], d["R_quat"][~g], rtol=0, atol=1e-12)      # convention pinned: R(rpy) == R(q)
    assert np.abs(d["R_rpy"][g] - d["R_quat"][g]).max() < 4.6e-3                          # gimbal branches: pybullet's own approximation
    # exactly at +-pi/2 the branch formula is exact
    ex = np.abs(np.abs(d["euler_in"][:, 1]) - np.pi / 2) == 0
    assert ex.sum() >= 3
    np.testing.assert_allclose(d["R_rpy"][ex], d["R_quat"][ex], rtol=0, atol=1e-7)
    # the oracle's quaternion -> rotation (used by its physics) is the tree's too
    np.testing.assert_allclose(O.quat_to_rotmat_bullet(q), d["R_quat"], rtol=0, atol=1e-13)


def test_compare_models_call_site_matches_reference():
    """simulations/CompareModels.py:46-56 through the reference's own LinearizedModel / QuadrotorDynamics / model_conversions
    (compare_models.npz): the loop body, calc_xdot on free states with (A, B) and (Ahat, Bhat), rpy_to_rot, geo_model_to_obs."""
    d = load("compare_models.npz")
    a, b, c = O.compare_models(d["obs"], d["A"], d["B"], dyn_m=float(d["dyn_m"]), dyn_J=d["dyn_J"], dyn_g=float(d["dyn_g"]))
    np.testing.assert_allclose(a, d["xdot_lin"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(b, d["xdot_geo"], rtol=0, atol=1e-12)
    np.testing.assert_array_equal(c, d["x_lin"])
    A0, B0 = O.linearized_AB()
    np.testing.assert_array_equal(A0, d["A"])
    np.testing.assert_array_equal(B0, d["B"])
    Ah, Bh = O.linearized_AB(noisy=True)
    np.testing.assert_allclose(O.linear_calc_xdot(d["x_free"], d["obs"][:, 16:], d["A"], d["B"]), d["xdot_free"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(O.linear_calc_xdot(d["x_free"], d["obs"][:, 16:], Ah, Bh), d["xdot_free_hat"], rtol=0, atol=1e-12)
    # the stale-J quirk is in the fixture: the geometric side's w_dot divides by the Hummingbird inertia, the linear side by the env's
    assert np.allclose(d["dyn_J"], [1.05, 1.05, 2.05]) and float(d["dyn_m"]) == 0.027
    np.testing.assert_allclose(O.rpy_to_rot(d["rpy"]), d["R_of_rpy"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(O.geo_model_to_obs(d["x18"]), d["obs16"], rtol=0, atol=1e-15)
    # all four branches of scipy's from_matrix are in the fixture (largest of R00, R11, R22, trace)
    R = d["x18"][:, 3:12].reshape(-1, 3, 3)
    dec = np.stack([R[:, 0, 0], R[:, 1, 1], R[:, 2, 2], np.trace(R, axis1=1, axis2=2)], axis=1)
    assert set(np.argmax(dec, axis=1)) == {0, 1, 2, 3}
    np.testing.assert_array_equal(O.geo_x_dot_to_linear(np.arange(12.0)), [3, 4, 5, 9, 10, 11, 6, 7, 8, 0, 1, 2])
