"""The reference's driver scripts (simulations/EnvGeometric.py, CBFTest.py, CBFTestOrd3.py) through their mirrors:
same GeometricEnv / do_control calls, the loop compared with the oracle's restatement of the same loop."""
import os

import numpy as np
import pytest

from oracle import np_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no HIP device")
    return torch


def test_envgeometric_do_control_with_wind_matches_oracle(gpu, tmp_path):
    from multidronesim_amd.simulations import EnvGeometric as S
    args = S.parse_args(["--num_drones", "3", "--duration_sec", "2", "--dtype", "float64", "--controller", "geometric"])
    geo = S.GeometricEnv(args, circle_init=True)
    env = geo.create_env(gui=True)
    assert env.NUM_DRONES == 3 and geo.conversion_mat.shape == (4, 4)
    np.testing.assert_allclose(geo.INIT_XYZS[2], [np.sin(2 * np.pi / 3), np.cos(2 * np.pi / 3), 0.0], atol=1e-15)     # EnvGeometric.py:507-510
    trajs = [S.Lemniscate(center=np.array([0, 0, .5]), omega=1.5, yaw_rate=0.0, phase_shift=(-np.pi / 4) * (num - 1)) for num in range(3)]
    geo.do_control(trajs=trajs)
    obs = np.asarray(geo.observations)
    assert obs.shape == (200, 3, 20) and len(geo.obs_ts) == 200                      # np.save(...) layout of :553
    path = tmp_path / "observations.npy"
    np.save(path, geo.observations)
    assert np.load(path).shape == (200, 3, 20)
    # the same loop on the oracle: wind 2.5e-4 N along x on every drone, every step (:34, :463-467)
    ora = O.AviaryOracle(geo.INIT_XYZS, geo.INIT_RPYS, pyb_freq=100, ctrl_freq=100)
    o = ora.step(np.zeros((3, 4)))                                                   # :431, before the first applyExternalForce
    ora.wind = np.array([S.wind_force, 0, 0])
    P = np.array([[1.0, 1.5, 0, 0, .5, 0.0, (-np.pi / 4) * (num - 1)] for num in range(3)])
    t = 0.0
    for k in range(200):
        pos, vel, acc, yaw, yd = O.lemniscate(t, P[:, 0], P[:, 1], P[:, 2:5], P[:, 5], P[:, 6])
        o = ora.step(O.geometric_compute(o, pos, vel, acc, yaw, yd))
        t += 0.01
        if k in (0, 99, 199):
            np.testing.assert_allclose(obs[k], o, atol=1e-8)
    xd = geo.geometric_xdot(obs[-1, 0])
    assert xd.shape == (12,) and np.isfinite(xd).all()


def test_envgeometric_setpoint_and_batch(gpu):
    """trajs=None: regulation towards TARGET_POSITIONS (:449-455); num_envs > 1 adds the leading axis to the log."""
    from multidronesim_amd.simulations import EnvGeometric as S
    args = S.parse_args(["--num_drones", "2", "--duration_sec", "3", "--num_envs", "5", "--controller", "geometric"])
    geo = S.GeometricEnv(args, circle_init=True)
    geo.create_env()
    geo.do_control(trajs=None, wind=False)
    obs = np.asarray(geo.observations)
    assert obs.shape == (300, 5, 2, 20)
    assert np.abs(obs[-1, :, :, 0:3] - geo.TARGET_POSITIONS).max() < 0.2          # 1 m step response, 3 s in (Kp 2.25, Kv 3.5)
    assert np.abs(obs[-1, :, :, 0:2] - geo.TARGET_POSITIONS[:, 0:2]).max() < 1e-3 and (np.diff(obs[::50, 0, 0, 2]) > 0).all()
    args.controller = "dlqr"
    geo2 = S.GeometricEnv(args)
    geo2.create_env()
    with pytest.raises(NotImplementedError):
        geo2.do_control()


@pytest.mark.parametrize("nd,kernel", [(2, 2), (3, 2), (4, 2), (7, 2)])
def test_cbftest_do_control_matches_oracle(gpu, nd, kernel):
    """simulations/CBFTest.py __main__ (:414-427): LQR-omega nominal, one sphere at the lemniscate centre, order-2 filter -- at the script's
    own default of 2 drones (:31), and with 3, 4 and 7: the mirror's loop goes through the persistent rollout kernel (any drone count up
    to 16 since round 4), statuses equal to the oracle loop's at every step."""
    from multidronesim_amd.simulations import CBFTest as S
    args = S.parse_args(["--num_drones", str(nd), "--duration_sec", "1", "--dtype", "float64"])
    assert args.controller == "lqr" and args.init_rad == .2
    geo = S.GeometricEnv(args, circle_init=True)
    geo.INIT_XYZS[:, 2] = 0.5 + 0.3 * np.arange(nd)                                  # no ground here: start at flight height, stacked
    env = geo.create_env()
    trajs = [S.Lemniscate(center=np.array([0, 0, 0.5 + 0.3 * k]), omega=0.5, yaw_rate=0) for k in range(nd)]
    cbf = S.DroneCBF(env, geo.linear_models, safety_radius=0.1, zscale=1)
    trk = S.DroneQPTracker(cbf, num_robots=nd)
    x_obs = np.array([np.array([[0, 0, .5], np.zeros(3)])])
    lib, h = env._lib, env._h
    geo.do_control(trajs=trajs, qpTracker=trk, x_obs_list=x_obs, obs_r_list=[.1])
    obs = np.asarray(geo.observations)
    assert obs.shape == (100, nd, 20) and geo.statuses.shape == (100, 1)
    assert geo.last_cbf_kernel == kernel                                             # which form the mirror ran
    P = np.array([[1.0, 0.5, 0, 0, 0.5 + 0.3 * k, 0.0, 0.0] for k in range(nd)])
    oobs, ohist = H.oracle_cbf_closed_loop(geo.INIT_XYZS[None], geo.INIT_RPYS[None], P[None], 100, cbf.Kcbf.reshape(-1), cbf.umax, 0.1, 1.0,
                                           list(x_obs), [.1], nominal="lqr_omega")
    np.testing.assert_array_equal(geo.statuses, ohist)
    np.testing.assert_allclose(obs[-1][:, :16], oobs[0][:, :16], atol=1e-6)


def test_cbftest_ord3_runs_the_yank_loop(gpu):
    from multidronesim_amd.simulations import CBFTestOrd3 as S
    args = S.parse_args(["--num_drones", "3", "--duration_sec", "1", "--dtype", "float64"])
    geo = S.GeometricEnv(args, init_type='lemniscate', lemniscate_a=1, center=np.array([0, 0, 0.5]))
    D = 3
    for i in range(D):                                                                 # lemniscate_initialize (:388-401)
        pos = O.lemniscate(0.0, 1.0, 0.5, np.array([0, 0, 0.5]), 0.0, (2 * np.pi / (D + 0.25)) * i)[0]
        np.testing.assert_allclose(geo.INIT_XYZS[i], pos, atol=1e-15)
    env = geo.create_env()
    trajs = [S.Lemniscate(a=1, center=np.array([0, 0, 0.5]), omega=0.5, yaw_rate=0, phase_shift=(2 * np.pi / (D + 0.25)) * num) for num in range(D)]
    cbf = S.DroneCBF(env, geo.linear_models, safety_radius=0.125, zscale=2, order=3, cbf_poles=np.array([-3.0, -3.6, -5.6]))
    trk = S.DroneQPTracker(cbf, num_robots=D, xdim=10, env=env, order=3)
    geo.do_control(trajs=trajs, qpTracker=trk, x_obs_list=None, obs_r_list=None)
    obs = np.asarray(geo.observations)
    assert obs.shape == (100, 3, 20) and np.isfinite(obs).all()
    assert geo.last_cbf_kernel == 2                                                    # the order-3 persistent kernel (k_cbf_rollout_o3, round 4)
    P = np.array([[1.0, 0.5, 0, 0, 0.5, 0.0, (2 * np.pi / (D + 0.25)) * num] for num in range(D)])
    oobs, ohist = H.oracle_cbf_closed_loop(geo.INIT_XYZS[None], geo.INIT_RPYS[None], P[None], 100, cbf.Kcbf.reshape(-1), cbf.umax, 0.125, 2.0,
                                           None, None, nominal="lqr_yank_omega", order=3, first_rpm=O.CF2P.HOVER_RPM)
    np.testing.assert_array_equal(geo.statuses, ohist)
    np.testing.assert_allclose(obs[-1][:, :16], oobs[0][:, :16], atol=1e-6)


@pytest.mark.parametrize("dtype,rtol", [("float64", 1e-9), ("float32", 3e-5)])
def test_lqr12_golden(gpu, dtype, rtol):
    """control/lqr/lqr_controller.py on model/linearized.py against the reference-minted fixture: host ARE gain for the true and
    the 'noisy' model, u (with the mixer's in-place clip of u[0]) and the RPM."""
    import os
    from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
    from multidronesim_amd.control import LQRController
    from multidronesim_amd.model import LinearizedModel
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "lqr12.npz"))
    n = d["obs"].shape[0]
    env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=1, initial_xyzs=np.zeros((1, 3)), initial_rpys=np.zeros((1, 3)),
                     physics=Physics.DYN, pyb_freq=100, ctrl_freq=100, num_envs=n, dtype=dtype)
    des = np.zeros((n, 1, 11))
    des[:, 0, 0:3], des[:, 0, 3:6], des[:, 0, 9], des[:, 0, 10] = d["pos_d"], d["vel_d"], d["yaw_d"], d["om_d"]
    for tag, noisy in (("true", False), ("noisy", True)):
        ctrl = LQRController(env, LinearizedModel(env), use_noisy_model=noisy)
        np.testing.assert_allclose(ctrl.K, d["K_" + tag], rtol=1e-7, atol=1e-9)
        act, u = ctrl.compute_batched(d["obs"].reshape(n, 1, 20), des)
        act, u = act.double().cpu().numpy().reshape(n, 4), u.double().cpu().numpy().reshape(n, 4)
        scale = np.abs(d["u_" + tag]).max(axis=0)
        assert (np.abs(u - d["u_" + tag]) / scale).max() < rtol
        # RPM: sqrt amplifies relative error near the thrust clip; compare motor thrusts instead (rpm^2)
        assert (np.abs(act ** 2 - d["act_" + tag] ** 2) / (d["act_" + tag] ** 2).max()).max() < rtol * 20
    ctrl.set_desired_trajectory(0, d["pos_d"][40], d["vel_d"][40], np.zeros(3), d["yaw_d"][40], d["om_d"][40])
    a1, u1 = ctrl.compute(d["obs"][40])                                              # reference signature
    assert (np.abs(u1 - d["u_noisy"][40]) / scale).max() < rtol
    env.close()


def test_envgeometric_default_lqr_controller_matches_oracle(gpu):
    """simulations/EnvGeometric.py as it runs out of the box: controller 'lqr' (LQRController, 12-state), the 'noisy' model
    (use_noisy_model=True, :551), wind, Lemniscates of :540 -- against the oracle's restatement of that loop; then the same
    controller on general (segment-table) trajectories."""
    from multidronesim_amd.simulations import EnvGeometric as S
    D = 3
    args = S.parse_args(["--num_drones", str(D), "--duration_sec", "2", "--dtype", "float64"])
    assert args.controller == "lqr"
    geo = S.GeometricEnv(args, circle_init=True)
    geo.INIT_XYZS[:, 2] = 0.5                                                         # no ground here: start at flight height
    geo.create_env()
    assert geo.linear_models[0].A.shape == (12, 12)
    trajs = [S.Lemniscate(center=np.array([0, 0, .5]), omega=1.5, yaw_rate=0.0, phase_shift=(-np.pi / 4) * (num - 1)) for num in range(D)]
    geo.do_control(trajs=trajs, use_noisy_model=True)
    obs = np.asarray(geo.observations)
    K = O.lqr12_gain(O.CF2P, noisy=True)
    P = np.array([[1.0, 1.5, 0, 0, .5, 0.0, (-np.pi / 4) * (num - 1)] for num in range(D)])

    def oracle_loop(desired):
        ora = O.AviaryOracle(geo.INIT_XYZS, geo.INIT_RPYS, pyb_freq=100, ctrl_freq=100)
        o = ora.step(np.zeros((D, 4)))
        ora.wind = np.array([S.wind_force, 0, 0])
        t = 0.0
        for k in range(200):
            pos, vel, yaw, yd = desired(t)
            act, _ = O.lqr12_compute(o, pos, vel, yaw, yd, K)
            o = ora.step(act)
            t += 0.01
        return o

    def lem(t):
        pos, vel, acc, yaw, yd = O.lemniscate(t, P[:, 0], P[:, 1], P[:, 2:5], P[:, 5], P[:, 6])
        return pos, vel, yaw, yd
    np.testing.assert_allclose(obs[-1], oracle_loop(lem), atol=1e-7)
    assert np.abs(obs[-1][:, :3] - O.lemniscate(2.0, P[:, 0], P[:, 1], P[:, 2:5], P[:, 5], P[:, 6])[0]).max() < 0.8   # it tracks (the LQR lags a 1.5 rad/s lemniscate; 0.51 m here)

    from oracle import np_trajectories as NT
    geo2 = S.GeometricEnv(args, circle_init=True)
    geo2.INIT_XYZS[:, 2] = 0.5
    geo2.create_env()
    mk = lambda T, j: T.Compound([T.Line(start=geo2.INIT_XYZS[j], end=geo2.INIT_XYZS[j] + np.array([0.5, 0.2, 0.4]), speed=.5),
                                  T.Wait(duration=0.5, position=geo2.INIT_XYZS[j] + np.array([0.5, 0.2, 0.4]))])
    import types
    Tg = types.SimpleNamespace(Compound=S.CompoundTrajectory, Line=S.LineTrajectory, Wait=S.WaitTrajectory)
    To = types.SimpleNamespace(Compound=NT.Compound, Line=NT.Line, Wait=NT.Wait)
    geo2.do_control(trajs=[mk(Tg, j) for j in range(D)], use_noisy_model=True)
    otr = [mk(To, j) for j in range(D)]

    def seg(t):
        rows = [tr(t) for tr in otr]
        return (np.array([r[0] for r in rows]), np.array([r[1] for r in rows]), np.array([r[3] for r in rows]), np.array([r[4] for r in rows]))
    np.testing.assert_allclose(np.asarray(geo2.observations)[-1], oracle_loop(seg), atol=1e-7)


@pytest.mark.parametrize("which", ["omega", "yank"])
def test_plain_lqr_lowlevel_loop_matches_oracle(gpu, which):
    """do_control without a qpTracker in the CBFTest / CBFTestOrd3 scripts (= the loops of EnvGeometricOmega.py:314-327 and
    EnvGeometricYankOmega.py:319-332): LQR nominal + its low-level controller + env.step, no hover offset games."""
    if which == "omega":
        from multidronesim_amd.simulations import CBFTest as S
    else:
        from multidronesim_amd.simulations import CBFTestOrd3 as S
    D, steps = 3, 150
    args = S.parse_args(["--num_drones", str(D), "--duration_sec", "1", "--dtype", "float64", "--control_freq_hz", "150", "--simulation_freq_hz", "150"])
    geo = S.GeometricEnv(args) if which == "omega" else S.GeometricEnv(args, init_type='circle', center=np.array([0, 0, 0.5]))
    geo.INIT_XYZS[:, 2] = 0.5
    env = geo.create_env()
    P = np.array([[1.0, 0.5, 0, 0, 0.5 + 0.1 * k, 0.3, 0.4 * k] for k in range(D)])
    trajs = [S.Lemniscate(a=1.0, omega=0.5, center=P[k, 2:5], yaw_rate=0.3, phase_shift=0.4 * k) for k in range(D)]
    geo.do_control(trajs=trajs, qpTracker=None)
    obs = np.asarray(geo.observations)
    assert obs.shape == (steps, D, 20) and not geo.statuses.any()
    c = O.CF2P
    dt = 1 / 150
    ora = O.AviaryOracle(geo.INIT_XYZS, geo.INIT_RPYS, pyb_freq=150, ctrl_freq=150)
    first = c.HOVER_RPM if which == "yank" else 0.0
    o = ora.step(np.full((D, 4), first))
    ll = O.YankOmegaOracle(D, c) if which == "yank" else O.ThrustOmegaOracle(D, c)
    K = O.lqr_yank_omega_gain(c, dt) if which == "yank" else O.lqr_omega_gain(c)
    t = 0.0
    for k in range(steps):
        pos, vel, acc, yaw, yd = O.lemniscate(t, P[:, 0], P[:, 1], P[:, 2:5], P[:, 5], P[:, 6])
        u = O.lqr_yank_omega_compute(o, pos, vel, yaw, K, c) if which == "yank" else O.lqr_omega_compute(o, pos, vel, yaw, K, c)
        o = ora.step(ll.compute_low_level(u, o, dt))
        t += dt
    np.testing.assert_allclose(obs[-1][:, :16], o[:, :16], atol=1e-7)
    assert np.abs(obs[-1][:, :3] - pos).max() < 0.5


@pytest.mark.parametrize("dtype,T,tol", [("float64", 120, 1e-9), ("float32", 60, 5e-5)])
def test_lqr_whole_rollout_equals_stepwise(gpu, dtype, T, tol):
    """mds_rollout_lqr_fused (T control steps of trajectory -> LQRController -> step in one launch, every obs logged) against
    T calls of mds_step_lqr on an identical env.  This closed loop amplifies perturbations ~50x per 0.2 s (fp32 and float64 runs
    of the SAME kernel drift apart at that rate), so the fp32 comparison uses a short horizon."""
    from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
    from multidronesim_amd.control import LQRController
    from multidronesim_amd.model import LinearizedModel
    E, D = 37, 3
    xyz, rpy, P = H.c2_setup(E, D, seed=4, offset=2.0, yaw_rate=0.2)
    envs = []
    for _ in range(2):
        env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=Physics.DYN, pyb_freq=200,
                         ctrl_freq=100, num_envs=E, dtype=dtype)
        env.set_trajectories(P)
        LQRController(env, LinearizedModel(env))
        env.step(gpu.zeros((E, D, 4), dtype=env.dtype))
        envs.append(env)
    a, b = envs
    last, log = a.rollout_geometric_fused(0.0, T, log=True, controller="lqr")
    t = 0.0
    for k in range(T):
        o = b.step_lqr(t)
        t += b.CTRL_TIMESTEP
        if k in (0, T // 2, T - 1):
            d = np.abs(log[k].double().cpu().numpy()[..., :16] - o.double().cpu().numpy()[..., :16]).max()
            assert d < tol, (k, d)
    np.testing.assert_allclose(last.double().cpu().numpy(), log[-1].double().cpu().numpy(), atol=0)
    np.testing.assert_allclose(a.get_state(), b.get_state(), atol=tol)
    a.close(); b.close()


@pytest.mark.parametrize("which", ["lqr_omega", "lqr_yank_omega"])
def test_nominal_whole_rollout_equals_stepwise(gpu, which):
    """mds_rollout_nominal_fused (LQR + low level + step for T control steps in one launch, PID memory in registers) against
    T calls of mds_step_nominal; float64, with drag and two physics substeps per control step."""
    from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
    from multidronesim_amd.control import LQROmegaController, LQRYankOmegaController, ThrustOmegaController, YankOmegaController
    from multidronesim_amd.model import LinearizedOmegaModel, LinearizedYankOmegaModel
    E, D, T = 21, 3, 90
    xyz, rpy, P = H.c2_setup(E, D, seed=6, offset=1.0, omega=0.6, yaw_rate=0.1)
    envs = []
    for _ in range(2):
        env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=Physics.PYB_DRAG, pyb_freq=200,
                         ctrl_freq=100, num_envs=E, dtype="float64")
        env.set_trajectories(P)
        if which == "lqr_omega":
            LQROmegaController(env, LinearizedOmegaModel(env), ThrustOmegaController(env))
        else:
            LQRYankOmegaController(env, LinearizedYankOmegaModel(env), YankOmegaController(env))
        env.set_cbf_nominal(which)
        env.step(gpu.full((E, D, 4), float(env.HOVER_RPM), dtype=env.dtype))
        envs.append(env)
    a, b = envs
    last, log = a.rollout_geometric_fused(0.0, T, log=True, controller="nominal")
    t = 0.0
    for k in range(T):
        o = b.step_nominal(t, return_action=True)[0]       # with the action wanted: the two-launch path (nominal, low level + step)
        t += b.CTRL_TIMESTEP
        if k in (0, 1, T // 2, T - 1):
            np.testing.assert_allclose(log[k].cpu().numpy(), o.cpu().numpy(), atol=1e-9, rtol=1e-12)
    np.testing.assert_allclose(a.get_state(), b.get_state(), atol=1e-9)
    o1 = a.rollout_geometric_fused(t, 10, controller="nominal")[0].cpu().numpy().copy()      # PID memory and RPM echo carried over
    for k in range(10):
        o2 = b.step_nominal(t + k * b.CTRL_TIMESTEP)
    np.testing.assert_allclose(o1, o2.cpu().numpy(), atol=1e-9, rtol=1e-12)
    a.close(); b.close()


@pytest.mark.parametrize("dtype,tol_transient,tol", [("float64", 1e-10, 1e-10), ("float32", 2e-3, 2e-5)])
def test_lqr_loop_against_the_reference_objects_in_the_loop(gpu, dtype, tol_transient, tol):
    """mds_step_lqr (trajectory -> LQRController -> mixer -> DYN step, wind on) against tests/golden/closed_loop_lqr_ref_in_loop.npz: 600
    control steps in which trajectory sampling and controller were the reference's OWN objects (only env.step was the oracle's).
    Measured: float64 <= 9e-13 throughout; fp32 3e-6 at step 50, 6e-4 at step 100 (the start-up transient saturates the motors -- the
    1000 rad^-2 attitude weights -- and amplifies rounding ~100x), back under 1e-5 from step 250 on (2e-6 .. 4e-6 to the end)."""
    from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
    from multidronesim_amd.control import LQRController
    from multidronesim_amd.model import LinearizedModel
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "closed_loop_lqr_ref_in_loop.npz"))
    P, every = d["params"], int(d["every"])
    D = P.shape[0]
    env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=d["xyz"], initial_rpys=np.zeros((D, 3)), physics=Physics.DYN,
                     pyb_freq=100, ctrl_freq=100, num_envs=1, dtype=dtype)
    env.set_trajectories(P)
    ctrl = LQRController(env, LinearizedModel(env))
    np.testing.assert_allclose(ctrl.K, d["K"], rtol=1e-8, atol=1e-10)
    env.step(gpu.zeros((1, D, 4), dtype=env.dtype))
    env.set_wind(d["wind"])
    t = 0.0
    for i in range(int(d["steps"])):
        o = env.step_lqr(t)
        t += env.CTRL_TIMESTEP
        if (i + 1) % every == 0:
            k = (i + 1) // every
            g = o.double().cpu().numpy().reshape(D, 20)
            assert np.abs(g[:, :16] - d["obs_log"][k][:, :16]).max() < (tol_transient if i + 1 < 300 else tol), (i + 1)
    env.close()


def test_fp16_storage_instantiations_of_the_lqr_paths(gpu):
    """fp16 state storage / fp32 arithmetic through the entry points added around the 12-state LQR: step, whole rollout, operator,
    obs adapters.  Throughput configuration: finite, unit quaternions, within fp16 storage accuracy of the fp32 run."""
    from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
    from multidronesim_amd.control import LQRController
    from multidronesim_amd.model import LinearizedModel
    from multidronesim_amd.utils import obs_to_lin_model
    E, D, T = 33, 2, 40
    xyz, rpy, P = H.c2_setup(E, D, seed=8, offset=0.0, omega=0.8)
    out = {}
    for dtype in ("float32", "float16"):
        env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=Physics.DYN, pyb_freq=100, ctrl_freq=100,
                         num_envs=E, dtype=dtype)
        env.set_trajectories(P)
        ctrl = LQRController(env, LinearizedModel(env))
        env.step(gpu.zeros((E, D, 4), dtype=env.dtype))
        t = 0.0
        for k in range(T // 2):
            o = env.step_lqr(t)
            t += env.CTRL_TIMESTEP
        last, log = env.rollout_geometric_fused(t, T // 2, log=True, controller="lqr")
        des = gpu.zeros((E, D, 11), dtype=env.dtype)
        act, u = ctrl.compute_batched(last, des)
        x12 = obs_to_lin_model(last, 12, env)
        for name, ten in (("last", last), ("log", log), ("act", act), ("u", u), ("x12", x12)):
            assert gpu.isfinite(ten).all(), (dtype, name)
        out[dtype] = last.double().cpu().numpy()
        assert np.abs(np.linalg.norm(out[dtype][..., 3:7], axis=-1) - 1).max() < 2e-3
        env.close()
    assert np.abs(out["float16"][..., :3] - out["float32"][..., :3]).max() < 5e-2


@pytest.mark.parametrize("which", ["omega", "yank"])
def test_envgeometric_omega_and_yank_omega_scripts_match_oracle(gpu, which):
    """The mirrors of simulations/EnvGeometricOmega.py / EnvGeometricYankOmega.py: their defaults and initial conditions (:28-30, :57,
    :356-381 / :381-406), and do_control(trajs, render, computed_K, use_noisy_model) -- LQR nominal + its low level + env.step
    (:314-327 / :319-332), one fused launch here -- against the oracle loop; 'dlqr' / computed_K are outside the path."""
    if which == "omega":
        from multidronesim_amd.simulations import EnvGeometricOmega as S
        d = S.parse_args([])
        assert (d.duration_sec, d.num_drones, d.init_rad, d.controller) == (50, 2, .2, 'lqr')
    else:
        from multidronesim_amd.simulations import EnvGeometricYankOmega as S
        d = S.parse_args([])
        assert (d.duration_sec, d.num_drones, d.init_rad, d.controller) == (5, 1, .2, 'lqr')
    D, steps = 3, 150
    args = S.parse_args(["--num_drones", str(D), "--duration_sec", "1", "--dtype", "float64", "--control_freq_hz", "150", "--simulation_freq_hz", "150",
                         "--physics", "dyn"])
    geo = S.GeometricEnv(args, circle_init=True)
    ang = 2 * np.pi * np.arange(D) / D
    xy = np.stack([np.sin(ang), np.cos(ang)] if which == "omega" else [np.cos(ang), np.sin(ang)], axis=1) * 0.2
    xy[0] = 0.0                                                          # the first drone stays at the origin
    np.testing.assert_allclose(geo.INIT_XYZS[:, :2], xy, atol=1e-15)
    np.testing.assert_allclose(geo.TARGET_POSITIONS, geo.INIT_XYZS + [0, 0, 1], atol=1e-15)
    np.testing.assert_allclose(geo.TARGET_RPYS, [[0, 0, np.pi / 2]] * D)
    geo.INIT_XYZS[:, 2] = 0.5
    env = geo.create_env()
    assert type(geo.linear_models[0]).__name__ == ("LinearizedOmegaModel" if which == "omega" else "LinearizedYankOmegaModel")
    with pytest.raises(NotImplementedError):
        geo.do_control(trajs=None, computed_K=np.zeros((4, 9)))
    P = np.array([[1.0, 0.5, 0, 0, 0.5 + 0.1 * k, 0.3, 0.4 * k] for k in range(D)])
    trajs = [S.Lemniscate(a=1.0, omega=0.5, center=P[k, 2:5], yaw_rate=0.3, phase_shift=0.4 * k) for k in range(D)]
    geo.do_control(trajs=trajs, render=False, use_noisy_model=False)
    obs = np.asarray(geo.observations)
    assert obs.shape == (steps, D, 20) and len(geo.obs_ts) == steps
    c = O.CF2P
    dt = 1 / 150
    ora = O.AviaryOracle(geo.INIT_XYZS, geo.INIT_RPYS, pyb_freq=150, ctrl_freq=150)
    o = ora.step(np.full((D, 4), c.HOVER_RPM if which == "yank" else 0.0))
    ll = O.YankOmegaOracle(D, c) if which == "yank" else O.ThrustOmegaOracle(D, c)
    K = O.lqr_yank_omega_gain(c, dt) if which == "yank" else O.lqr_omega_gain(c)
    t = 0.0
    for k in range(steps):
        pos, vel, acc, yaw, yd = O.lemniscate(t, P[:, 0], P[:, 1], P[:, 2:5], P[:, 5], P[:, 6])
        u = O.lqr_yank_omega_compute(o, pos, vel, yaw, K, c) if which == "yank" else O.lqr_omega_compute(o, pos, vel, yaw, K, c)
        o = ora.step(ll.compute_low_level(u, o, dt))
        np.testing.assert_allclose(obs[k][:, :16], o[:, :16], atol=1e-7)
        t += dt
    # use_noisy_model=True designs the gain on (Ahat, Bhat): a different gain, hence a different (still stable) run
    geo2 = S.GeometricEnv(args, circle_init=True)
    geo2.INIT_XYZS[:, 2] = 0.5
    geo2.create_env()
    geo2.do_control(trajs=trajs, render=False, use_noisy_model=True)
    obs2 = np.asarray(geo2.observations)
    assert np.isfinite(obs2).all() and np.abs(obs2[-1][:, :3] - obs[-1][:, :3]).max() > (1e-6 if which == "omega" else 0.0)
    assert np.abs(obs2[-1][:, :3] - pos).max() < 0.6
    # the reference's default is use_noisy_model=True (:265): a call without the argument is the explicit-True run, bit for bit; and the
    # flag is per call, not sticky: False after True on the same object is the first run again
    geo3 = S.GeometricEnv(args, circle_init=True)
    geo3.INIT_XYZS[:, 2] = 0.5
    geo3.create_env()
    geo3.do_control(trajs=trajs, render=False)
    np.testing.assert_array_equal(np.asarray(geo3.observations), obs2)
    geo3.observations, geo3.obs_ts = [], []
    geo3.INIT_XYZS[:, 2] = 0.5
    geo3.create_env()
    geo3.do_control(trajs=trajs, render=False, use_noisy_model=False)
    np.testing.assert_array_equal(np.asarray(geo3.observations), obs)


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-10), ("float32", 1e-5)])
def test_default_lqr_loop_at_config2_size_every_drone_against_the_c_oracle(gpu, dtype, tol):
    """The loop simulations/EnvGeometric.py runs out of the box (controller 'lqr' on the 12-state model, wind on from the first control
    step, Lemniscates) at BASELINE config 2's size -- 4 096 envs x 4 drones, 600 control steps -- every drone against the plain-C oracle
    (oracle/c_oracle.c, pinned on lqr12.npz and on the reference-objects loop): the step-by-step fused kernel and the whole-rollout
    kernel.  Measured: float64 4e-14, fp32 3.8e-6 / 4.4e-6 at step 600 (north_star's 1e-5; the saturating start-up transient, where fp32
    is 6e-4 off at step 100, has died out by then: DESIGN.md section 2)."""
    from oracle import c_oracle as CO
    from multidronesim_amd.control import LQRController
    from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
    from multidronesim_amd.model import LinearizedModel
    from multidronesim_amd.simulations.EnvGeometric import wind_force
    E, D, steps = 4096, 4, 600
    xyz, rpy, P = H.c2_setup(E, D, seed=1000, phase="c2")
    K = O.lqr12_gain(O.CF2P)
    wind = np.array([wind_force, 0.0, 0.0])
    ref, _ = CO.lqr_loop(CO.AviaryC(xyz.reshape(-1, 3), rpy.reshape(-1, 3)), P, K, steps, wind=wind, threads=H.oracle_threads())

    def make():
        env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=Physics.DYN, pyb_freq=100,
                         ctrl_freq=100, num_envs=E, dtype=dtype)
        ctrl = LQRController(env, LinearizedModel(env))
        np.testing.assert_allclose(ctrl.K, K, rtol=1e-9, atol=1e-12)
        env.set_trajectories(P)
        env.step(gpu.zeros((E, D, 4), dtype=env.dtype, device=env.device))
        env.set_wind(wind)
        return env

    env = make()
    t = 0.0
    for _ in range(steps):
        obs = env.step_lqr(t)
        t += env.CTRL_TIMESTEP
    got = obs.double().cpu().numpy().reshape(-1, 20)
    err = np.abs(got[:, :16] - ref[:, :16]).max()
    print(f"[LQR default loop, {E * D} drones x {steps} steps, {dtype}] step by step: max |state err| {err:.3e}")
    assert err < tol
    env.close()
    env = make()
    last = env.rollout_geometric_fused(0.0, steps, controller="lqr")[0]
    err2 = np.abs(last.double().cpu().numpy().reshape(-1, 20)[:, :16] - ref[:, :16]).max()
    print(f"[LQR default loop, {dtype}] whole-rollout kernel: max |state err| {err2:.3e}")
    assert err2 < tol
    env.close()
