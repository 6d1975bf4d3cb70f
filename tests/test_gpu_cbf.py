"""GPU parity of the ECBF safety filter (a11-a15): constraint rows against the golden vectors
minted from the reference's own dense construction, and the QP against the oracle's exact
active-set solver (cvxopt is absent: QP solution parity against cvxopt is unpinned; the
problem is strictly convex so the minimiser is unique)."""
import os

import numpy as np
import pytest

from oracle import np_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def mds():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no HIP device")
    from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
    from multidronesim_amd.cbf.cbf import DroneCBF
    from multidronesim_amd.cbf.qptracker import DroneQPTracker
    from multidronesim_amd.model.linear_omega import LinearizedOmegaModel
    from multidronesim_amd.model.linear_yank_omega import LinearizedYankOmegaModel
    import types
    return types.SimpleNamespace(**locals())


def make_env(mds, E, D, dtype="float64"):
    return mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=D, initial_xyzs=np.zeros((D, 3)), initial_rpys=np.zeros((D, 3)),
                          physics=mds.Physics.DYN, pyb_freq=100, ctrl_freq=100, num_envs=E, dtype=dtype)


@pytest.mark.parametrize("order", [2, 3])
@pytest.mark.parametrize("dtype,rtol", [("float64", 1e-10), ("float32", 2e-5)])
def test_cbf_rows_golden(mds, order, dtype, rtol):
    d = np.load(os.path.join(G, f"cbf_rows_o{order}.npz"))
    Model = mds.LinearizedOmegaModel if order == 2 else mds.LinearizedYankOmegaModel
    for k in range(int(d["n_cases"])):
        x, xdes, xobs, obsr = d[f"c{k}_x"], d[f"c{k}_xdes"], d[f"c{k}_xobs"], d[f"c{k}_obsr"]
        Gr, hr = d[f"c{k}_G"], d[f"c{k}_h"]
        N = x.shape[0]
        env = make_env(mds, 1, N, dtype)
        cbf = mds.DroneCBF(env, [Model(env) for _ in range(N)], safety_radius=float(d["safety_radius"]), zscale=float(d["zscale"]),
                           order=order, cbf_poles=d["poles"])
        np.testing.assert_allclose(cbf.Kcbf.reshape(-1), d["Kcbf"], rtol=1e-10)
        np.testing.assert_allclose(cbf.umax, d["umax"], rtol=1e-12)
        cbf.set_xdes(xdes)
        Gg, hg = cbf._build_ineq_const(x, False, list(xobs) if len(obsr) else None, list(obsr) if len(obsr) else None)
        assert Gg.shape == Gr.shape and hg.shape == hr.shape
        sG = np.abs(Gr).max(axis=1, keepdims=True) + 1e-30
        assert np.abs(Gg - Gr).max() <= rtol * np.abs(Gr).max()
        # per-row relative: each row against its own scale (terms of h_ij cancel; scale = sum of |terms| ~ |h| + |G| row scale)
        scale = np.maximum(np.abs(hr), 1.0) + 50 * np.abs(Gr).max(axis=1)
        assert (np.abs(hg - hr) / scale).max() <= rtol * 10
        env.close()


def c4_scene(E, D, seed=0, dz=0.3, vz=0.35, xy=0.12, crowd=None):
    """C4-like scene that makes barrier rows ACTIVE but feasible.  With the omega linearisation only
    the thrust reaches the barrier, through the vertical separation e_z, so drones are stacked
    ~dz apart and closing vertically; 4 static spheres r=0.1 at (+-0.25, +-0.25, 0.5).  Every 8th
    env is made infeasible (two drones 5 cm apart at the same height, at rest) to exercise the
    nominal fallback.  crowd != None spreads the drones far apart instead (no active row)."""
    rng = np.random.default_rng(seed)
    obs = np.zeros((E, D, 20))
    obs[..., 0] = rng.normal(size=(E, D)) * xy
    obs[..., 1] = rng.normal(size=(E, D)) * xy
    obs[..., 2] = 0.75 + dz * np.arange(D) + rng.normal(size=(E, D)) * 0.03
    if crowd is not None:
        ang = 2 * np.pi * np.arange(D) / D
        obs[..., 0], obs[..., 1] = crowd * np.cos(ang), crowd * np.sin(ang)
        obs[..., 2] = 2.0 + 1.0 * np.arange(D)
    obs[..., 3:7] = [0, 0, 0, 1]
    obs[..., 7:10] = rng.uniform(-0.05, 0.05, size=(E, D, 3))
    obs[..., 10:12] = rng.normal(size=(E, D, 2)) * 0.1
    obs[..., 12] = rng.normal(size=(E, D)) * (vz if crowd is None else 0.01)
    obs[..., 16:20] = O.CF2P.HOVER_RPM
    if crowd is None and D >= 2:
        bad = np.arange(E) % 8 == 7
        obs[bad, 1, 0:3] = obs[bad, 0, 0:3] + np.array([0.05, 0.0, 0.001])
        obs[bad, 0, 10:13], obs[bad, 1, 10:13] = 0.0, 0.0        # at rest inside each other's radius: k0*h < 0, g ~ 0
    xdes = np.zeros((E, D, 9))
    xdes[..., 2] = rng.uniform(-0.5, 0.5, size=(E, D))
    xdes[..., 3:6] = rng.normal(size=(E, D, 3)) * 0.05
    xdes[..., 6:9] = obs[..., 0:3] + rng.normal(size=(E, D, 3)) * 0.02
    unom = np.concatenate([rng.normal(size=(E, D, 1)) * 0.05, rng.normal(size=(E, D, 3)) * 6.0], axis=-1)
    x_obs = [np.array([[sx * 0.25, sy * 0.25, 0.5], [0, 0, 0]]) for sx in (-1, 1) for sy in (-1, 1)]
    obs_r = [0.1] * 4
    return obs, xdes, unom, x_obs, obs_r


@pytest.mark.parametrize("D,dtype,tol", [(16, "float64", 1e-8), (16, "float32", 2e-5), (5, "float32", 2e-5), (2, "float64", 1e-8)])
def test_cbf_filter_matches_oracle_qp(mds, D, dtype, tol):
    E = 48
    obs, xdes, unom, x_obs, obs_r = c4_scene(E, D, seed=D)
    env = make_env(mds, E, D, dtype)
    cbf = mds.DroneCBF(env, [mds.LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2,
                       cbf_poles=np.array([-2.2, -2.4]))
    trk = mds.DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
    us, st = trk.compute_control_batched(obs, xdes, unom, x_obs, obs_r)
    us, st = us.double().cpu().numpy(), st.cpu().numpy()
    n_active, n_fallback = 0, 0
    for e in range(E):
        x = O.obs_to_lin_model(obs[e], 9)
        u_ref, status = O.cbf_filter(x, xdes[e], unom[e], 2, cbf.Kcbf.reshape(-1), cbf.umax, 0.1, 1.0, O.CF2P, np.array(x_obs), obs_r)
        assert st[e] == status, (e, st[e], status)
        if status == 0:
            np.testing.assert_allclose(us[e], u_ref, atol=tol, rtol=0)
            n_active += int(np.abs(u_ref[:, 0] - unom[e][:, 0]).max() > 1e-6)
        else:
            np.testing.assert_array_equal(us[e], unom[e].astype(np.float32 if dtype == "float32" else np.float64))
            n_fallback += 1
    assert n_active >= E // 2, "scene must exercise active barrier rows"
    assert n_fallback >= E // 8 - 1, "scene must exercise the nominal fallback"
    env.close()


def test_cbf_filter_inactive_rows_leave_nominal_untouched(mds):
    """Far-apart drones, small nominal inputs: no row is active -> u_safe == clip(u_hat) exactly."""
    E, D = 8, 4
    obs, xdes, unom, x_obs, obs_r = c4_scene(E, D, seed=1, crowd=3.0)
    unom[..., 1:] = np.clip(unom[..., 1:], -20, 20)
    env = make_env(mds, E, D, "float32")
    cbf = mds.DroneCBF(env, [mds.LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2)
    trk = mds.DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
    us, st = trk.compute_control_batched(obs, xdes, unom, x_obs, obs_r)
    assert (st.cpu().numpy() == 0).all()
    want = unom.astype(np.float32).copy()
    want[..., 1:] = np.clip(want[..., 1:], -10, 10)
    np.testing.assert_array_equal(us.cpu().numpy(), want)
    env.close()


def test_cbf_filter_infeasible_falls_back_and_reference_signature(mds):
    """Two drones 1 cm apart with a tiny thrust box: no feasible thrust -> status 1, nominal returned;
    also the reference's single-env compute_control(obs, xdes, u_nominal, x_obs=..., obs_r_list=...)."""
    D = 2
    env = make_env(mds, 1, D, "float64")
    cbf = mds.DroneCBF(env, [mds.LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2)
    trk = mds.DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
    obs = np.zeros((D, 20)); obs[:, 6] = 1
    obs[0, 0:3] = [0, 0, 0.5]; obs[1, 0:3] = [0.0, 0.0, 0.51]
    obs[0, 12] = 1.0; obs[1, 12] = -1.0           # closing fast along z
    xdes = np.zeros((D, 9)); xdes[:, 6:9] = obs[:, 0:3]
    unom = np.zeros((D, 4))
    x = O.obs_to_lin_model(obs, 9)
    u_ref, status = O.cbf_filter(x, xdes, unom, 2, cbf.Kcbf.reshape(-1), cbf.umax, 0.1, 1.0, O.CF2P)
    u = trk.compute_control(obs, xdes, unom)
    assert u.shape == (D, 4)
    if status == 1:
        np.testing.assert_array_equal(u, unom)
    else:
        np.testing.assert_allclose(u, u_ref, atol=1e-8)
    # feasible, active case through the same signature, with one obstacle as the reference passes it
    obs[1, 0:3] = [0.05, 0, 0.8]; obs[0, 12] = 0.35; obs[1, 12] = -0.35     # stacked 0.3 m apart, closing vertically
    xdes[:, 6:9] = obs[:, 0:3]
    x_obs, obs_r = [np.array([[0.1, 0.1, 0.2], [0, 0, 0]])], [0.1]
    u = trk.compute_control(obs, xdes, unom, x_obs=x_obs, obs_r_list=obs_r)
    x = O.obs_to_lin_model(obs, 9)
    u_ref, status = O.cbf_filter(x, xdes, unom, 2, cbf.Kcbf.reshape(-1), cbf.umax, 0.1, 1.0, O.CF2P, np.array(x_obs), obs_r)
    assert status == 0 and np.abs(u_ref[:, 0]).max() > 1e-3
    np.testing.assert_allclose(u, u_ref, atol=1e-8)
    env.close()


@pytest.mark.parametrize("model", ["cf2p", "cf2x"])
def test_thrust_omega_golden(mds, model):
    """control/low_level/thrust_omega_ctrl.py through mds_thrust_omega_from_rates (stateful, 40 calls)."""
    from multidronesim_amd.control.low_level.thrust_omega_ctrl import ThrustOmegaController
    d = np.load(os.path.join(G, "thrust_omega.npz"))
    u, cur, rpm = d[f"{model}_u"], d[f"{model}_cur"], d[f"{model}_rpm"]
    n = u.shape[1]
    for dtype, rtol in (("float64", 1e-12), ("float32", 2e-6)):
        env = mds.CtrlAviary(drone_model=mds.DroneModel(model), num_drones=1, initial_xyzs=np.zeros((n, 1, 3)), initial_rpys=np.zeros((n, 1, 3)),
                             physics=mds.Physics.DYN, pyb_freq=100, ctrl_freq=100, num_envs=n, dtype=dtype)
        ctl = ThrustOmegaController(env)
        for t in range(u.shape[0]):
            got = ctl.compute_batched(u[t].reshape(n, 1, 4), cur[t].reshape(n, 1, 3)).double().cpu().numpy().reshape(n, 4)
            np.testing.assert_allclose(got, rpm[t], rtol=rtol)
        ctl.reset()
        got = ctl.computeControlFromInput(u[0, 0], 0.01, cur[0, 0])          # reference signature, fresh state
        np.testing.assert_allclose(got, rpm[0, 0], rtol=rtol)
        env.close()


# gates = next round number above what tests/tools/c4_parity_probe.py measures on MI355X (gpurun_out/r3_parity_probe.log, round 3):
# max abs state error against the float64 oracle at steps 150 / 220 (the bench window) / 1000 (north_star's horizon):
#   float64 4.9e-14 / 6.6e-14 / 1.9e-12, float32 2.7e-5 / 3.5e-5 / 1.4e-3, float32c 8.2e-6 / 1.1e-5 / 3.6e-4; statuses equal at every step.
# The closed loop amplifies a perturbation ~1e3-fold over 1000 steps on this scene (float64's own 1e-15 rounding ends at 2e-12), so
# fp32 -- 6e-8 per stored state -- cannot hold north_star's 1e-5 to step 1000 on C4 whatever the kernel does; it holds it over the
# bench window on the bench's headline scene (test_c4_full_size_properties).
@pytest.mark.parametrize("dtype,tol150,tol220,tol1000", [("float64", 1e-11, 1e-11, 1e-9), ("float32", 1e-4, 1e-4, 5e-3), ("float32c", 5e-5, 5e-5, 2e-3)])
def test_c4_closed_loop_matches_oracle(mds, dtype, tol150, tol220, tol1000):
    """Config 4 pipeline (geometric nominal -> ECBF QP -> ThrustOmega -> step), 8 envs x 6 drones on crossing Lemniscates with 4
    sphere obstacles, against the oracle loop over 1000 control steps (north_star's horizon; the bench window ends at step 220):
    per-env statuses equal at EVERY step -- 19 % of the env-steps of this scene are infeasible, so both branches are exercised --
    and the state within the gates above at steps 150, 220 and 1000."""
    from tests import helpers as H2
    E, D, steps = 8, 6, 1000
    xyz, rpy, P = H2.c2_setup(E, D, phase="c3", offset=0.0, omega=1.0)
    xyz[..., 2] = 0.5 + 0.25 * np.arange(D)            # stacked start: barrier rows act through e_z
    P[..., 4] = 0.5 + 0.12 * np.arange(D)              # trajectories 12 cm apart vertically: rows become active
    x_obs = [np.array([[sx * 0.5, sy * 0.5, 0.5], [0, 0, 0]]) for sx in (-1, 1) for sy in (-1, 1)]
    obs_r = [0.1] * 4
    env = mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=mds.Physics.DYN,
                         pyb_freq=100, ctrl_freq=100, num_envs=E, dtype=dtype)
    env.set_trajectories(P)
    cbf = mds.DroneCBF(env, [mds.LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2)
    trk = mds.DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
    marks = {150: tol150, 220: tol220, 1000: tol1000}
    _, ohist, ref = H2.oracle_cbf_closed_loop(xyz, rpy, P, steps, cbf.Kcbf.reshape(-1), cbf.umax, 0.1, 1.0, x_obs, obs_r, record_at=tuple(marks))
    assert 0.05 < ohist.mean() < 0.5                     # feasible and infeasible env-steps both occur
    env.step(mds.torch.zeros((E, D, 4), dtype=env.dtype))
    t = 0.0
    for k in range(steps):
        gobs, st = env.step_cbf_geometric(t, trk, x_obs, obs_r)
        got = st.cpu().numpy()
        assert np.array_equal(got, ohist[k]), f"{dtype}: status differs from the oracle's first at step {k}: envs {np.nonzero(got != ohist[k])[0].tolist()}"
        t += env.CTRL_TIMESTEP
        if k + 1 in marks:
            g = gobs.double().cpu().numpy()
            err = np.abs(g[..., :16] - ref[k + 1][..., :16]).max()
            assert err < marks[k + 1], f"{dtype}: state error {err:.2e} at step {k + 1}"
            assert np.isfinite(g).all()
    env.close()


@pytest.mark.parametrize("dtype,rtol", [("float64", 1e-10), ("float32", 2e-5)])
def test_lqr_omega_golden(mds, dtype, rtol):
    """control/lqr/lqr_omega_controller.py: host ARE gain == reference K, u = -K e + cap through the kernel."""
    from multidronesim_amd.control.lqr.lqr_omega_controller import LQROmegaController
    d = np.load(os.path.join(G, "lqr_omega.npz"))
    n = d["obs"].shape[0]
    env = make_env(mds, n, 1, dtype)
    ctrl = LQROmegaController(env, mds.LinearizedOmegaModel(env), None)
    np.testing.assert_allclose(ctrl.K, d["K"], rtol=1e-8, atol=1e-10)
    des = np.zeros((n, 1, 11))
    des[:, 0, 0:3], des[:, 0, 3:6], des[:, 0, 9] = d["pos_d"], d["vel_d"], d["yaw_d"]
    u = ctrl.compute_batched(d["obs"].reshape(n, 1, 20), des).double().cpu().numpy().reshape(n, 4)
    scale = np.abs(d["u"]).max(axis=0)
    assert (np.abs(u - d["u"]) / scale).max() < rtol
    ctrl.set_desired_trajectory(0, d["pos_d"][5], d["vel_d"][5], np.zeros(3), d["yaw_d"][5], 0.0)
    _, u1 = ctrl.compute(d["obs"][5], skip_low_level=True)                     # reference signature
    assert (np.abs(u1 - d["u"][5]) / scale).max() < rtol
    env.close()


def test_c4_closed_loop_lqr_nominal_matches_oracle(mds):
    """The loop exactly as simulations/CBFTest.py runs it: LQROmegaController nominal -> ECBF QP -> ThrustOmega."""
    from multidronesim_amd.control.lqr.lqr_omega_controller import LQROmegaController
    from tests import helpers as H2
    E, D, steps = 6, 5, 120
    xyz, rpy, P = H2.c2_setup(E, D, phase="c3", offset=0.0, omega=0.8)
    xyz[..., 2] = 0.5 + 0.3 * np.arange(D)
    P[..., 4] = 0.5 + 0.3 * np.arange(D)
    x_obs = [np.array([[0.0, 0.0, 0.5], [0, 0, 0]])]                            # CBFTest.py:421-422
    obs_r = [0.1]
    env = mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=mds.Physics.DYN,
                         pyb_freq=100, ctrl_freq=100, num_envs=E, dtype="float64")
    env.set_trajectories(P)
    LQROmegaController(env, mds.LinearizedOmegaModel(env), None)
    env.set_cbf_nominal("lqr_omega")
    cbf = mds.DroneCBF(env, [mds.LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2)
    trk = mds.DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
    oobs, ohist = H2.oracle_cbf_closed_loop(xyz, rpy, P, steps, cbf.Kcbf.reshape(-1), cbf.umax, 0.1, 1.0, x_obs, obs_r, nominal="lqr_omega")
    env.step(mds.torch.zeros((E, D, 4), dtype=env.dtype))
    t = 0.0
    for k in range(steps):
        gobs, st = env.step_cbf_geometric(t, trk, x_obs, obs_r)
        np.testing.assert_array_equal(st.cpu().numpy(), ohist[k])
        t += env.CTRL_TIMESTEP
    assert np.abs(gobs.double().cpu().numpy()[..., :16] - oobs[..., :16]).max() < 1e-6
    env.close()


def c4_scene_o3(E, D, seed=0):
    """Order-3 (yank-omega, xdim 10) version of the stacked scene: F from the RPM echo of the obs."""
    obs, xdes9, unom, x_obs, obs_r = c4_scene(E, D, seed=seed, dz=0.75, vz=0.15)   # zscale 2: e_z counts half; stiffer poles
    rng = np.random.default_rng(seed + 100)
    obs[..., 16:20] = O.CF2P.HOVER_RPM * (1 + 0.05 * rng.normal(size=(E, D, 4)))
    xdes = np.zeros((E, D, 10))
    xdes[..., 0:3] = xdes9[..., 0:3]
    xdes[..., 3] = O.CF2P.M * O.CF2P.G
    xdes[..., 4:10] = xdes9[..., 3:9]
    unom = np.concatenate([rng.normal(size=(E, D, 1)) * 0.2, rng.normal(size=(E, D, 3)) * 4.0], axis=-1)
    x_obs3 = [np.array([xo[0], [0, 0, 0], [0, 0, 0]]) for xo in x_obs]
    return obs, xdes, unom, x_obs3, obs_r


@pytest.mark.parametrize("D,dtype,tol", [(16, "float64", 1e-7), (16, "float32", 1e-4), (4, "float64", 1e-7), (21, "float32", 1e-4)])
def test_cbf_filter_order3_matches_oracle_qp(mds, D, dtype, tol):
    """Order 3 (LinearizedYankOmegaModel, poles of CBFTestOrd3.py:452): 3D coupled variables per env."""
    E = 24
    obs, xdes, unom, x_obs, obs_r = c4_scene_o3(E, D, seed=D)
    env = make_env(mds, E, D, dtype)
    cbf = mds.DroneCBF(env, [mds.LinearizedYankOmegaModel(env) for _ in range(D)], safety_radius=0.125, zscale=2.0, order=3,
                       cbf_poles=np.array([-3.0, -3.6, -5.6]))
    trk = mds.DroneQPTracker(cbf, order=3, num_robots=D, xdim=10, env=env)
    us, st = trk.compute_control_batched(obs, xdes, unom, x_obs, obs_r)
    us, st = us.double().cpu().numpy(), st.cpu().numpy()
    n_active = 0
    for e in range(E):
        x = O.obs_to_lin_model(obs[e], 10)
        u_ref, status = O.cbf_filter(x, xdes[e], unom[e], 3, cbf.Kcbf.reshape(-1), cbf.umax, 0.125, 2.0, O.CF2P, np.array(x_obs), obs_r)
        assert st[e] == status, (e, st[e], status)
        if status == 0:
            np.testing.assert_allclose(us[e], u_ref, atol=tol, rtol=0)
            n_active += int(np.abs(u_ref[:, :3] - unom[e][:, :3]).max() > 1e-6)
    assert n_active >= E // 3
    env.close()


@pytest.mark.parametrize("D,n_obs", [(1, 2), (32, 16), (32, 0)])
def test_cbf_filter_size_limits(mds, D, n_obs):
    """Edge sizes of the wave-per-env QP: a single drone (no pair rows), and the maxima D = 32, 16 obstacles
    (1072 rows, 17 per lane)."""
    E = 6
    obs, xdes, unom, _, _ = c4_scene(E, D, seed=40 + D)
    rng = np.random.default_rng(D)
    x_obs = [np.array([[rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(0.2, 0.6)], [0, 0, 0]]) for _ in range(n_obs)]
    obs_r = [0.08] * n_obs
    env = make_env(mds, E, D, "float64")
    cbf = mds.DroneCBF(env, [mds.LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2)
    trk = mds.DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
    us, st = trk.compute_control_batched(obs, xdes, unom, x_obs if n_obs else None, obs_r if n_obs else None)
    us, st = us.double().cpu().numpy(), st.cpu().numpy()
    for e in range(E):
        x = O.obs_to_lin_model(obs[e], 9)
        u_ref, status = O.cbf_filter(x, xdes[e], unom[e], 2, cbf.Kcbf.reshape(-1), cbf.umax, 0.1, 1.0, O.CF2P,
                                     np.array(x_obs) if n_obs else None, obs_r if n_obs else None)
        assert st[e] == status
        if status == 0:
            np.testing.assert_allclose(us[e], u_ref, atol=1e-8)
    env.close()
    with pytest.raises(Exception):
        big = make_env(mds, 1, 33, "float32")
        mds.DroneCBF(big, [mds.LinearizedOmegaModel(big) for _ in range(33)], order=2).configure()


@pytest.mark.parametrize("dtype,rtol", [("float64", 1e-10), ("float32", 3e-5)])
def test_lqr_yank_omega_golden(mds, dtype, rtol):
    """control/lqr/lqr_YO_controller.py + control/low_level/yank_omega_ctrl.py against the reference-minted fixture:
    host ARE gain, u = -K e with the thrust state from the obs' RPM echo, and the stateful yank -> thrust -> PID low level."""
    from multidronesim_amd.control import LQRYankOmegaController, YankOmegaController
    d = np.load(os.path.join(G, "lqr_yank_omega.npz"))
    T, n = d["obs"].shape[:2]
    env = make_env(mds, n, 1, dtype)
    ctrl = LQRYankOmegaController(env, mds.LinearizedYankOmegaModel(env), YankOmegaController(env))
    np.testing.assert_allclose(ctrl.K, d["K"], rtol=1e-8, atol=1e-10)
    scale = np.abs(d["u"]).reshape(-1, 4).max(axis=0)
    for t in range(T):
        des = np.zeros((n, 1, 11))
        des[:, 0, 0:3], des[:, 0, 3:6], des[:, 0, 9] = d["pos_d"][t], d["vel_d"][t], d["yaw_d"][t]
        u = ctrl.compute_batched(d["obs"][t].reshape(n, 1, 20), des).double().cpu().numpy().reshape(n, 4)
        assert (np.abs(u - d["u"][t]) / scale).max() < rtol
        rpm = ctrl.yo_controller.compute_low_level_batched(d["u"][t].reshape(n, 1, 4), d["obs"][t].reshape(n, 1, 20))
        np.testing.assert_allclose(rpm.double().cpu().numpy().reshape(n, 4), d["rpm"][t], rtol=max(rtol, 1e-12) * 10)
    env.close()
    env = make_env(mds, 1, 1, dtype)                                              # reference signatures, single drone
    ctrl = LQRYankOmegaController(env, mds.LinearizedYankOmegaModel(env), YankOmegaController(env))
    ctrl.set_desired_trajectory(0, d["pos_d"][0, 5], d["vel_d"][0, 5], np.zeros(3), d["yaw_d"][0, 5], 0.0)
    act, u1 = ctrl.compute(d["obs"][0, 5])
    assert (np.abs(u1 - d["u"][0, 5]) / scale).max() < rtol
    np.testing.assert_allclose(act, d["rpm"][0, 5], rtol=max(rtol, 1e-12) * 10)
    env.close()


@pytest.mark.parametrize("dtype,steps,tol", [("float64", 150, 1e-10), ("float32", 150, 1e-4), ("float32c", 150, 1e-4)])
def test_order3_closed_loop_matches_oracle(mds, dtype, steps, tol):
    """simulations/CBFTestOrd3.py:306-352 for every env: LQRYankOmegaController nominal (yank - M G, kept quirk) ->
    order-3 ECBF QP (poles and radii of :452) -> YankOmega low level -> step, against the oracle loop.  The order-3
    filter pushes hard once active (the oracle drifts ~0.7 m off the nominal path here).  Measured over the full 150 steps
    (profiles/r02_fp32_gates.log): float64 6e-14, float32 1.5e-5, compensated fp32 1.3e-5; statuses equal at every step."""
    from multidronesim_amd.control import LQRYankOmegaController, YankOmegaController
    from tests import helpers as H2
    E, D = 4, 4
    xyz, rpy, P = H2.c2_setup(E, D, phase="c3", offset=0.0, omega=0.5)             # CBFTestOrd3.py:450
    xyz[..., 2] = 0.5 + 0.6 * np.arange(D)
    P[..., 4] = 0.5 + 0.6 * np.arange(D)
    x_obs = [np.array([[0.0, 0.0, -0.3], [0, 0, 0], [0, 0, 0]])]
    obs_r = [0.1]
    env = mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=mds.Physics.DYN,
                         pyb_freq=100, ctrl_freq=100, num_envs=E, dtype=dtype)
    env.set_trajectories(P)
    LQRYankOmegaController(env, mds.LinearizedYankOmegaModel(env), YankOmegaController(env))
    cbf = mds.DroneCBF(env, [mds.LinearizedYankOmegaModel(env) for _ in range(D)], safety_radius=0.125, zscale=2.0, order=3,
                       cbf_poles=np.array([-3.0, -3.6, -5.6]))
    trk = mds.DroneQPTracker(cbf, order=3, num_robots=D, xdim=10, env=env)
    with pytest.raises(RuntimeError):                                             # geometric nominal has no yank output
        env.step_cbf_geometric(0.0, trk, x_obs, obs_r)
    env.set_cbf_nominal("lqr_yank_omega")
    oobs, ohist = H2.oracle_cbf_closed_loop(xyz, rpy, P, steps, cbf.Kcbf.reshape(-1), cbf.umax, 0.125, 2.0, x_obs, obs_r,
                                            nominal="lqr_yank_omega", order=3, first_rpm=O.CF2P.HOVER_RPM)
    plain, _ = H2.oracle_cbf_closed_loop(xyz, rpy, P, steps, cbf.Kcbf.reshape(-1), np.array([1e9] * 4), 1e-3, 2.0, None, None,
                                         nominal="lqr_yank_omega", order=3, first_rpm=O.CF2P.HOVER_RPM, Fmin=-1e9, Fmax=1e9)
    assert np.abs(plain[..., :3] - oobs[..., :3]).max() > (0.02 if steps >= 150 else 1e-3)   # the filter is doing something in this scene
    env.step(mds.torch.full((E, D, 4), O.CF2P.HOVER_RPM, dtype=env.dtype))
    t = 0.0
    for k in range(steps):
        gobs, st = env.step_cbf_geometric(t, trk, x_obs, obs_r)
        np.testing.assert_array_equal(st.cpu().numpy(), ohist[k])
        t += env.CTRL_TIMESTEP
    g = gobs.double().cpu().numpy()
    assert np.isfinite(g).all()
    assert np.abs(g[..., :16] - oobs[..., :16]).max() < tol
    rel = np.abs(g[..., 16:] - oobs[..., 16:]).max() / O.CF2P.HOVER_RPM
    assert rel < tol
    env.close()


def test_cbf_filter_longest_first_dispatch_is_invisible(mds):
    """From the second call on the QP kernel walks the envs in cost classes rebuilt from earlier iteration counts
    (k_cbf_order, batches >= 1024 envs).  The mapping must stay a bijection: same inputs -> bit-identical outputs on
    every call, and a different batch afterwards is still solved env by env like the oracle."""
    E, D = 2048, 6
    obs, xdes, unom, x_obs, obs_r = c4_scene(E, D, seed=11)
    env = make_env(mds, E, D, "float32")
    cbf = mds.DroneCBF(env, [mds.LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2,
                       cbf_poles=np.array([-2.2, -2.4]))
    trk = mds.DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
    ref_u, ref_s = None, None
    for call in range(11):                                  # classes are rebuilt after calls 1 and 9
        us, st = trk.compute_control_batched(obs, xdes, unom, x_obs, obs_r)
        us, st = us.cpu().numpy().copy(), st.cpu().numpy().copy()
        if ref_u is None:
            ref_u, ref_s = us, st
            assert 0.05 < st.mean() < 0.3 and (np.abs(us[..., 0] - unom[..., 0]).max(axis=1) > 1e-6).mean() > 0.5
        else:
            np.testing.assert_array_equal(us, ref_u)
            np.testing.assert_array_equal(st, ref_s)
    obs2, xdes2, unom2, _, _ = c4_scene(E, D, seed=12)      # stale classes, new data
    us, st = trk.compute_control_batched(obs2, xdes2, unom2, x_obs, obs_r)
    us, st = us.double().cpu().numpy(), st.cpu().numpy()
    for e in range(0, E, 97):
        x = O.obs_to_lin_model(obs2[e], 9)
        u_ref, status = O.cbf_filter(x, xdes2[e], unom2[e], 2, cbf.Kcbf.reshape(-1), cbf.umax, 0.1, 1.0, O.CF2P, np.array(x_obs), obs_r)
        assert st[e] == status
        if status == 0:
            np.testing.assert_allclose(us[e], u_ref, atol=2e-5, rtol=0)
    env.close()


@pytest.mark.parametrize("scene,z,tol40,tol220", [("under", -3.0, 5e-6, 1e-5), ("level", 0.5, 5e-6, 1e-1)])
def test_c4_full_size_properties(mds, scene, z, tol40, tol220):
    """BASELINE config 4 at its full size (16 384 envs x 16 drones) over the bench's window (220 control steps = 20 warm-up + T = 200),
    on the bench's two scenes: 'under' (spheres at z = -3: the C4 headline -- every QP feasible, a third of the envs iterating) and
    'level' (SURVEY 8d's spheres at z = 0.5: a third of the envs infeasible).  Size-independent properties: replicated envs stay
    bitwise equal wherever they sit in the batch (the longest-first dispatch permutes the envs between calls), a 64-env batch holding
    the same envs gives the same bits, every row is finite with a unit quaternion and a 0/1 status.  Against the oracle: a strided
    sample of 8 envs -- statuses equal at EVERY one of the 220 steps, state within 5e-6 at step 40 and, at step 220, within
    north_star's 1e-5 on 'under' (measured 2.6e-6, tests/tools/c4_parity_probe.py).  'level' is a chaotic closed loop -- the float64
    kernel itself ends 6e-11 from the float64 oracle, a 1e4-fold amplification of its rounding in 220 steps -- and fp32 agrees to
    2.5e-2 there (gate 1e-1); its statuses still equal the oracle's at every step.  The persistent rollout kernel
    (mds_rollout_cbf_geometric_fused) must reproduce the step-by-step loop's statuses for ALL 16 384 envs at every step."""
    from tests import helpers as H2
    E, D, steps = 16384, 16, 220
    xyz, rpy, P = H2.c2_setup(E, D, seed=1000, phase="c3")      # (bench.py's generator and seed)
    P[..., 4] = 0.5 + 0.3 * np.arange(D)
    xyz[..., 2] = 0.5 + 0.3 * np.arange(D)
    for dst in (8192, 16384 - 64 - 5):                         # copies of envs 0..63
        xyz[dst:dst + 64], P[dst:dst + 64] = xyz[0:64], P[0:64]
    x_obs = [np.array([[sx * 0.5, sy * 0.5, z], [0, 0, 0]]) for sx in (-1, 1) for sy in (-1, 1)]
    obs_r = [0.1] * 4
    poles = np.array([-2.2, -2.4])
    idx = np.arange(0, E, E // 8)

    def run(lo, hi, fused=False):
        env = mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=D, initial_xyzs=xyz[lo:hi], initial_rpys=rpy[lo:hi],
                             physics=mds.Physics.DYN, pyb_freq=100, ctrl_freq=100, num_envs=hi - lo, dtype="float32")
        env.set_trajectories(P[lo:hi])
        cbf = mds.DroneCBF(env, [mds.LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2, cbf_poles=poles)
        trk = mds.DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
        env.step(mds.torch.zeros((hi - lo, D, 4), dtype=env.dtype))
        o40 = its = None
        if fused:
            slog = mds.torch.empty((steps, hi - lo), dtype=mds.torch.int32, device=env.device)
            o, _ = env.rollout_cbf_geometric_fused(0.0, steps, trk, x_obs, obs_r, steps_per_launch=44, status_log=slog)
            o, hist = o.clone(), slog
        else:
            t, hist, its = 0.0, [], []
            for k in range(steps):
                o, st = env.step_cbf_geometric(t, trk, x_obs, obs_r)
                hist.append(st.clone())
                if k >= 20:
                    its.append((cbf.last_iterations() > 0).float().mean())
                if k == 39 and hi - lo == E:
                    o40 = o[idx].double().cpu().numpy()
                t += env.CTRL_TIMESTEP
            o, hist, its = o.clone(), mds.torch.stack(hist), mds.torch.stack(its).cpu().numpy()
        out = (o, hist, cbf.Kcbf.reshape(-1).copy(), np.array(cbf.umax, dtype=np.float64), o40, its)
        env.close()
        return out

    obs, hist, Kcbf, umax, o40, its = run(0, E)
    for dst in (8192, 16384 - 64 - 5):
        assert mds.torch.equal(obs[0:64], obs[dst:dst + 64])
        assert mds.torch.equal(hist[:, 0:64], hist[:, dst:dst + 64])
    so, sh, *_ = run(0, 64)                                      # no dispatch classes at this size
    assert mds.torch.equal(so, obs[0:64]) and mds.torch.equal(sh, hist[:, 0:64])
    assert mds.torch.isfinite(obs).all()
    assert (obs[..., 3:7].norm(dim=-1) - 1).abs().max().item() < 1e-5
    assert set(np.unique(hist.cpu().numpy()).tolist()) <= {0, 1}
    fb = hist[20:].float().mean(dim=1).cpu().numpy()             # share of infeasible envs at each step of the bench window
    if scene == "under":
        assert fb.max() == 0.0, f"the headline scene must keep every QP feasible over the window: {fb.max()}"
        assert 0.1 < its.mean() < 0.9 and its[-1] > 0.05         # ... and active: a sizeable share of the envs iterate, also at the end
    else:
        assert 0.1 < fb.mean() < 0.9
    _, ohist, orec = H2.oracle_cbf_closed_loop(xyz[idx], rpy[idx], P[idx], steps, Kcbf, umax, 0.1, 1.0, x_obs, obs_r, record_at=(40, 220))
    np.testing.assert_array_equal(hist[:, idx].cpu().numpy(), np.array(ohist))
    e40 = np.abs(o40[..., :16] - orec[40][..., :16]).max()
    e220 = np.abs(obs[idx].double().cpu().numpy()[..., :16] - orec[220][..., :16]).max()
    assert e40 < tol40 and e220 < tol220, (scene, e40, e220)
    # The persistent kernel at full size against the step-by-step loop.  The two contract FMAs differently (observations equal to
    # rounding); 'under' amplifies that to 1e-6 at most and every one of the 3.6 M env-step statuses is the same.  'level' is chaotic
    # and a third of its QPs sit at the feasibility boundary: measured (MI355X, round 3), 35 env-steps of 3 604 480 differ, in 18 of the
    # 16 384 envs, each of which took the other branch once and went its own way from there -- the same thing a different compiler
    # version would do to the step-by-step loop itself.  Gate: no env on 'under', at most 64 envs (0.4 %) on 'level'.
    fo, fh, *_ = run(0, E, fused=True)
    envs_diff = int((fh != hist).any(dim=0).sum().item())
    assert envs_diff <= (0 if scene == "under" else 64), f"statuses of {envs_diff} envs differ between the persistent rollout and the step-by-step loop"
    same = ~(fh != hist).any(dim=0)[idx].cpu().numpy()
    ef = (fo[idx].double() - obs[idx].double()).abs()[..., :16].amax(dim=(1, 2)).cpu().numpy()
    assert ef[same].max() < 2 * tol220, ef


@pytest.mark.parametrize("nominal", ["geometric", "lqr_omega"])
def test_one_launch_cbf_step_matches_the_three_launch_path(mds, nominal, monkeypatch):
    """k_cbf_step (MDS_CBF_FUSED=1: nominal controller, the wave's 64 / D QPs and the low level + physics step in one launch) against
    the default three launches on the same scene: per-env statuses and iteration counts equal at every step, observations equal to
    rounding (the stages share their device functions; the launches may contract FMAs differently), float64 and float32; the two-chain
    rollout of the one-launch form equals its step-by-step loop bitwise.  D = 16 (4 envs per wave), a ragged last wave (E = 70 -> 1120
    drones = 17 full waves + 32 lanes), obstacles among the drones so that rows go active and some envs fall back."""
    from multidronesim_amd.control.lqr.lqr_omega_controller import LQROmegaController
    E, D, steps = 70, 16, 60
    xyz, rpy, P = H.c2_setup(E, D, phase="c3", offset=1.5)
    P[..., 4] = 0.5 + 0.3 * np.arange(D)
    xyz[..., 2] = 0.5 + 0.3 * np.arange(D)
    x_obs = [np.array([[sx * 0.5, sy * 0.5, 0.5], [0, 0, 0]]) for sx in (-1, 1) for sy in (-1, 1)]
    obs_r = [0.1] * 4
    for dtype, tol in (("float64", 1e-9), ("float32", 2e-4)):
        out = {}
        for fused in ("0", "1"):
            if nominal == "lqr_omega":
                monkeypatch.setenv("MDS_CBF_FUSED", fused)       # the environment switch (read at mds_cbf_configure) ...
            env = mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=mds.Physics.DYN,
                                 pyb_freq=100, ctrl_freq=100, num_envs=E, dtype=dtype)
            env.set_trajectories(P)
            if nominal == "lqr_omega":
                LQROmegaController(env, mds.LinearizedOmegaModel(env), None)
                env.set_cbf_nominal("lqr_omega")
            cbf = mds.DroneCBF(env, [mds.LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2,
                               cbf_poles=np.array([-2.2, -2.4]))
            trk = mds.DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
            if nominal != "lqr_omega":
                env.set_cbf_step_kernel(fused == "1")           # ... and the API call (mds_cbf_set_step_kernel)
            env.step(mds.torch.zeros((E, D, 4), dtype=env.dtype))
            t, hist, its = 0.0, [], []
            for k in range(steps):
                o, st = env.step_cbf_geometric(t, trk, x_obs, obs_r)
                hist.append(st.cpu().numpy().copy())
                its.append(cbf.last_iterations().cpu().numpy().copy())
                t += env.CTRL_TIMESTEP
            assert env.cbf_last_step_kernel() == int(fused)       # the library reports which form it launched
            out[fused] = (o.double().cpu().numpy().copy(), np.array(hist), np.array(its), env.get_state())
            if fused == "1":                                    # the C loop on two chains issues the same launches
                b = mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=mds.Physics.DYN,
                                   pyb_freq=100, ctrl_freq=100, num_envs=E, dtype=dtype)
                b.set_trajectories(P)
                if nominal == "lqr_omega":
                    LQROmegaController(b, mds.LinearizedOmegaModel(b), None)
                    b.set_cbf_nominal("lqr_omega")
                cb2 = mds.DroneCBF(b, [mds.LinearizedOmegaModel(b) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2,
                                   cbf_poles=np.array([-2.2, -2.4]))
                tb = mds.DroneQPTracker(cb2, num_robots=D, xdim=9, env=b)
                if nominal != "lqr_omega":
                    b.set_cbf_step_kernel(True)
                b.set_rollout_streams(2)
                b.step(mds.torch.zeros((E, D, 4), dtype=b.dtype))
                ob, sb = b.rollout_cbf_geometric(0.0, steps, tb, x_obs, obs_r)
                assert b.cbf_last_step_kernel() == 1
                np.testing.assert_array_equal(ob.double().cpu().numpy(), out["1"][0])
                np.testing.assert_array_equal(b.get_state(), out["1"][3])
                b.close()
            env.close()
        np.testing.assert_array_equal(out["0"][1], out["1"][1])
        np.testing.assert_array_equal(out["0"][2], out["1"][2])
        assert 0.0 < out["0"][1].mean() < 1.0 and out["0"][2].max() >= 2        # fallbacks and real iterations both occur
        assert np.abs(out["0"][0][..., :16] - out["1"][0][..., :16]).max() < tol
        np.testing.assert_allclose(out["0"][3], out["1"][3], atol=tol)


def test_four_envs_per_wave_qp_kernel_matches_the_default(mds, monkeypatch):
    """k_cbf_filter_q4 (MDS_CBF_Q4=1: one env per 16-lane DPP row, four per wavefront, the same active-set iteration with per-row
    reductions) against the one-env-per-wave kernel on the same closed loop: statuses equal at every step, observations equal to
    rounding (a tie between equally violated rows may be broken differently: the minimiser is the same), E = 70 (two envs of the last
    wave unused), D = 16 and D = 8, float64 and float32; and against the oracle's statuses for D = 6 (lanes 6..15 of a row idle)."""
    from tests import helpers as H2
    for D, E, steps in ((16, 70, 50), (8, 37, 50)):
        xyz, rpy, P = H2.c2_setup(E, D, phase="c3", offset=1.5)
        P[..., 4] = 0.5 + 0.3 * np.arange(D)
        xyz[..., 2] = 0.5 + 0.3 * np.arange(D)
        x_obs = [np.array([[sx * 0.5, sy * 0.5, 0.5], [0, 0, 0]]) for sx in (-1, 1) for sy in (-1, 1)]
        obs_r = [0.1] * 4
        for dtype, tol in (("float64", 1e-9), ("float32", 2e-4)):
            out = {}
            for q4 in ("0", "1"):
                monkeypatch.setenv("MDS_CBF_Q4", q4)
                env = mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=mds.Physics.DYN,
                                     pyb_freq=100, ctrl_freq=100, num_envs=E, dtype=dtype)
                env.set_trajectories(P)
                cbf = mds.DroneCBF(env, [mds.LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2,
                                   cbf_poles=np.array([-2.2, -2.4]))
                trk = mds.DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
                env.step(mds.torch.zeros((E, D, 4), dtype=env.dtype))
                t, hist = 0.0, []
                for k in range(steps):
                    o, st = env.step_cbf_geometric(t, trk, x_obs, obs_r)
                    hist.append(st.cpu().numpy().copy())
                    t += env.CTRL_TIMESTEP
                out[q4] = (o.double().cpu().numpy().copy(), np.array(hist), int(cbf.last_iterations().max().item()))
                env.close()
            np.testing.assert_array_equal(out["0"][1], out["1"][1])
            assert np.abs(out["0"][0][..., :16] - out["1"][0][..., :16]).max() < tol
            if D == 16:
                assert 0.0 < out["1"][1].mean() < 1.0 and out["1"][2] >= 2           # fallbacks and real iterations both occur
    monkeypatch.setenv("MDS_CBF_Q4", "1")
    E, D, steps = 8, 6, 150
    xyz, rpy, P = H2.c2_setup(E, D, phase="c3", offset=0.0, omega=1.0)
    xyz[..., 2] = 0.5 + 0.25 * np.arange(D)
    P[..., 4] = 0.5 + 0.12 * np.arange(D)
    x_obs = [np.array([[sx * 0.5, sy * 0.5, 0.5], [0, 0, 0]]) for sx in (-1, 1) for sy in (-1, 1)]
    obs_r = [0.1] * 4
    env = mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=mds.Physics.DYN,
                         pyb_freq=100, ctrl_freq=100, num_envs=E, dtype="float64")
    env.set_trajectories(P)
    cbf = mds.DroneCBF(env, [mds.LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2)
    trk = mds.DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
    oobs, ohist = H2.oracle_cbf_closed_loop(xyz, rpy, P, steps, cbf.Kcbf.reshape(-1), cbf.umax, 0.1, 1.0, x_obs, obs_r)
    env.step(mds.torch.zeros((E, D, 4), dtype=env.dtype))
    t = 0.0
    for k in range(steps):
        gobs, st = env.step_cbf_geometric(t, trk, x_obs, obs_r)
        np.testing.assert_array_equal(st.cpu().numpy(), ohist[k])
        t += env.CTRL_TIMESTEP
    assert np.abs(gobs.double().cpu().numpy()[..., :16] - oobs[..., :16]).max() < 1e-9
    env.close()


@pytest.mark.parametrize("streams", [1, 2])
def test_cbf_rollout_equals_stepwise(mds, streams):
    """mds_rollout_cbf_geometric (C loop; with two streams the env halves run as independent chains with their own cost
    classes) against the same number of mds_step_cbf_geometric calls: bitwise equal observations, states and statuses.
    2080 envs x 16 drones: the halves split at env 1040 (a 256-drone batch boundary), both above the 1024-env threshold of
    the longest-first dispatch."""
    from tests import helpers as H2
    E, D, steps = 2080, 16, 26
    xyz, rpy, P = H2.c2_setup(E, D, phase="c3")
    P[..., 4] = 0.5 + 0.3 * np.arange(D)
    xyz[..., 2] = 0.5 + 0.3 * np.arange(D)
    x_obs = [np.array([[sx * 0.5, sy * 0.5, 0.5], [0, 0, 0]]) for sx in (-1, 1) for sy in (-1, 1)]
    obs_r = [0.1] * 4
    out = []
    for mode in ("steps", "rollout"):
        env = mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=mds.Physics.DYN,
                             pyb_freq=100, ctrl_freq=100, num_envs=E, dtype="float32")
        env.set_trajectories(P)
        cbf = mds.DroneCBF(env, [mds.LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2,
                           cbf_poles=np.array([-2.2, -2.4]))
        trk = mds.DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
        env.set_rollout_streams(streams)
        env.step(mds.torch.zeros((E, D, 4), dtype=env.dtype))
        if mode == "steps":
            t = 0.0
            for k in range(steps):
                o, st = env.step_cbf_geometric(t, trk, x_obs, obs_r)
                t += env.CTRL_TIMESTEP
        else:
            o, st = env.rollout_cbf_geometric(0.0, steps - 6, trk, x_obs, obs_r)
            t = 0.0
            for k in range(steps - 6):
                t += env.CTRL_TIMESTEP                                                       # the time the loop would have reached
            o, st = env.rollout_cbf_geometric(t, 6, trk, x_obs, obs_r)                       # a second call chains on
        out.append((o.cpu().numpy().copy(), st.cpu().numpy().copy(), env.get_state()))
        env.close()
    assert 0.0 < out[0][1].mean() < 1.0
    for a, b in zip(*out):
        np.testing.assert_array_equal(a, b)


def test_cbf_rollout_order3_two_chains_equals_stepwise(mds):
    """The order-3 loop (yank-omega LQR nominal reading the RPM echo of obs, YankOmega low level) through
    mds_rollout_cbf_geometric with the env halves on two streams: bitwise the step-by-step loop.  1088 envs x 4 drones
    (halves of 544 envs = 2176 drones, a 256-drone batch boundary)."""
    from multidronesim_amd.control import LQRYankOmegaController, YankOmegaController
    from tests import helpers as H2
    E, D, steps = 1088, 4, 20
    xyz, rpy, P = H2.c2_setup(E, D, phase="c3", offset=0.0, omega=0.5)
    xyz[..., 2] = 0.5 + 0.6 * np.arange(D)
    P[..., 4] = 0.5 + 0.6 * np.arange(D)
    rng = np.random.default_rng(3)
    xyz[..., :2] += 0.05 * rng.standard_normal((E, D, 2))             # envs differ
    x_obs = [np.array([[0.0, 0.0, -0.3], [0, 0, 0], [0, 0, 0]])]
    obs_r = [0.1]
    out = []
    for mode in ("steps", "rollout"):
        env = mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=mds.Physics.DYN,
                             pyb_freq=100, ctrl_freq=100, num_envs=E, dtype="float32")
        env.set_trajectories(P)
        LQRYankOmegaController(env, mds.LinearizedYankOmegaModel(env), YankOmegaController(env))
        cbf = mds.DroneCBF(env, [mds.LinearizedYankOmegaModel(env) for _ in range(D)], safety_radius=0.125, zscale=2.0, order=3,
                           cbf_poles=np.array([-3.0, -3.6, -5.6]))
        trk = mds.DroneQPTracker(cbf, order=3, num_robots=D, xdim=10, env=env)
        env.set_cbf_nominal("lqr_yank_omega")
        env.set_rollout_streams(2)
        env.step(mds.torch.full((E, D, 4), O.CF2P.HOVER_RPM, dtype=env.dtype))
        if mode == "steps":
            t = 0.0
            for k in range(steps):
                o, st = env.step_cbf_geometric(t, trk, x_obs, obs_r)
                t += env.CTRL_TIMESTEP
        else:
            o, st = env.rollout_cbf_geometric(0.0, steps, trk, x_obs, obs_r)
        out.append((o.cpu().numpy().copy(), st.cpu().numpy().copy(), env.get_state()))
        env.close()
    assert np.isfinite(out[0][0]).all()
    for a, b in zip(*out):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("nominal", ["geometric", "lqr_omega"])
def test_cbf_persistent_rollout_matches_the_stepwise_loop(mds, nominal):
    """k_cbf_rollout (mds_rollout_cbf_geometric_fused: several control steps per launch, a workgroup owns whole envs, state and nominal
    input stay on the chip, the workgroup's QPs handed out heaviest first) against the step-by-step loop on the same scene: per-env
    statuses equal at EVERY step (status_log), last-step iteration counts equal, the logged observations of the last steps and the
    final state equal to rounding (shared device functions; the kernels contract FMAs differently), float64, float32 and compensated
    fp32.  D = 16, a ragged last workgroup (E = 70 -> 1120 drones = 2 full 512-drone workgroups + 96 lanes), launches of 25 + 25 + 10
    steps, a 7-slot observation ring that wraps, obstacles among the drones so that rows go active and some envs are infeasible."""
    from multidronesim_amd.control.lqr.lqr_omega_controller import LQROmegaController
    E, D, steps, spl, slots = 70, 16, 60, 25, 7
    xyz, rpy, P = H.c2_setup(E, D, phase="c3", offset=1.5)
    P[..., 4] = 0.5 + 0.3 * np.arange(D)
    xyz[..., 2] = 0.5 + 0.3 * np.arange(D)
    x_obs = [np.array([[sx * 0.5, sy * 0.5, 0.5], [0, 0, 0]]) for sx in (-1, 1) for sy in (-1, 1)]
    obs_r = [0.1] * 4

    def make(dtype):
        env = mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=mds.Physics.DYN,
                             pyb_freq=100, ctrl_freq=100, num_envs=E, dtype=dtype)
        env.set_trajectories(P)
        if nominal == "lqr_omega":
            LQROmegaController(env, mds.LinearizedOmegaModel(env), None)
            env.set_cbf_nominal("lqr_omega")
        cbf = mds.DroneCBF(env, [mds.LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2,
                           cbf_poles=np.array([-2.2, -2.4]))
        trk = mds.DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
        env.step(mds.torch.zeros((E, D, 4), dtype=env.dtype))
        return env, cbf, trk

    # fp32: the scene is chaotic where envs are infeasible (the two kernels' rounding differences grow): measured max 3.1e-4 on one of
    # 17 920 logged entries (2.2e-3 relative on 4 of 4 480 RPM entries) with the packed row build, every other entry below 2e-4
    for dtype, tol in (("float64", 1e-9), ("float32", 1e-3), ("float32c", 1e-3)):
        env, cbf, trk = make(dtype)
        t, hist, olog = 0.0, [], []
        for k in range(steps):
            o, st = env.step_cbf_geometric(t, trk, x_obs, obs_r)
            hist.append(st.cpu().numpy().copy())
            olog.append(o.double().cpu().numpy().copy())
            t += env.CTRL_TIMESTEP
        its = cbf.last_iterations().cpu().numpy().copy()
        ref_state = env.get_state()
        env.close()
        hist = np.array(hist)
        assert 0.0 < hist.mean() < 1.0 and its.max() >= 2          # infeasible envs and real iterations both occur

        b, cb2, tb = make(dtype)
        ring = mds.torch.full((slots, E, D, 20), float("nan"), dtype=b.dtype, device=b.device)
        slog = mds.torch.full((steps, E), -1, dtype=mds.torch.int32, device=b.device)
        ob, sb = b.rollout_cbf_geometric_fused(0.0, steps, tb, x_obs, obs_r, steps_per_launch=spl, obs_log=ring, first_slot=3, status_log=slog)
        assert b.cbf_last_step_kernel() == 2
        np.testing.assert_array_equal(slog.cpu().numpy(), hist)                          # statuses at every step
        np.testing.assert_array_equal(sb.cpu().numpy(), hist[-1])
        np.testing.assert_array_equal(cb2.last_iterations().cpu().numpy(), its)
        r = ring.double().cpu().numpy()
        for k in range(steps - slots, steps):                                            # the ring holds the last `slots` steps
            np.testing.assert_allclose(r[(3 + k) % slots][..., :16], olog[k][..., :16], rtol=0, atol=tol, err_msg=f"{dtype} step {k}")
            assert np.mean(np.abs(r[(3 + k) % slots][..., :16] - olog[k][..., :16]) > 0.2 * tol) < 1e-3     # ... and nearly all far closer
            np.testing.assert_allclose(r[(3 + k) % slots][..., 16:], olog[k][..., 16:], rtol=10 * tol, atol=0)           # RPM echo
            assert np.mean(np.abs(r[(3 + k) % slots][..., 16:] / olog[k][..., 16:] - 1) > tol) < 2e-3
        np.testing.assert_array_equal(ob.double().cpu().numpy(), r[(3 + steps - 1) % slots])
        np.testing.assert_allclose(b.get_state(), ref_state, rtol=0, atol=tol)
        # without a log only the last observation is written; a second call continues the same loop
        b2, _, tb2 = make(dtype)
        b2.rollout_cbf_geometric_fused(0.0, 35, tb2, x_obs, obs_r, steps_per_launch=spl)
        t35 = 0.0
        for _ in range(35):
            t35 += b2.CTRL_TIMESTEP                   # the loop's own clock: t += CTRL_TIMESTEP per step (not 35 * dt)
        o2, s2 = b2.rollout_cbf_geometric_fused(t35, steps - 35, tb2, x_obs, obs_r, steps_per_launch=spl)
        np.testing.assert_array_equal(o2.double().cpu().numpy(), ob.double().cpu().numpy())
        np.testing.assert_array_equal(s2.cpu().numpy(), sb.cpu().numpy())
        b.close()
        b2.close()


def test_cbf_persistent_rollout_rejects_what_it_does_not_cover(mds):
    """RK4 / drag / env effects return MDS_EUNSUPPORTED (the caller uses mds_rollout_cbf_geometric); any drone count up to 16 is covered
    (round 4; test_cbf_persistent_rollout_small_shapes)."""
    E, D = 4, 6
    xyz, rpy, P = H.c2_setup(E, D, phase="c3", offset=1.5)
    env = mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=mds.Physics.DYN,
                         pyb_freq=100, ctrl_freq=100, num_envs=E, dtype="float32", integrator="rk4")
    env.set_trajectories(P)
    cbf = mds.DroneCBF(env, [mds.LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2)
    trk = mds.DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
    env.step(mds.torch.zeros((E, D, 4), dtype=env.dtype))
    from multidronesim_amd import MdsError
    with pytest.raises(MdsError) as ei:
        env.rollout_cbf_geometric_fused(0.0, 5, trk)
    assert ei.value.status == -6                    # MDS_EUNSUPPORTED
    env.rollout_cbf_geometric(0.0, 5, trk)          # the general loop serves it
    env.close()


@pytest.mark.parametrize("D,n_obs,spl,E", [(4, 0, 1, 11), (8, 2, 3, 11), (4, 16, 7, 11), (8, 0, 64, 11),
                                             # any D <= 16 (an env padded to 4 / 8 / 16 lanes): the reference's own scripts run 2 and 7 drones
                                             (2, 1, 5, 11), (3, 2, 50, 11), (7, 1, 9, 11), (5, 4, 4, 11), (12, 3, 13, 11), (1, 2, 6, 11), (16, 4, 10, 11),
                                             # several workgroups, more than 64 envs per workgroup (D = 4: 128), a partial last workgroup
                                             (4, 2, 10, 333), (8, 3, 10, 301), (7, 2, 10, 200), (2, 1, 10, 300)])
def test_cbf_persistent_rollout_small_shapes(mds, D, n_obs, spl, E):
    """The persistent kernel away from the C4 shape: any drone count up to 16 -- 4 and 8 drones per env (128 / 64 envs per workgroup), the
    reference's own 2 (simulations/CBFTest.py:31) and 7 (CBFTestOrd3.py:31), 1, 3, 5, 12: an env occupies 4, 8 or 16 lanes, the rest
    idle --, no obstacles and sixteen, one control step per launch and more steps per launch than the call has, a single partial
    workgroup and several workgroups with a partial last one -- statuses equal to the step-by-step loop's at every step, state to rounding."""
    steps = 40
    xyz, rpy, P = H.c2_setup(E, D, phase="c3", offset=1.0)
    P[..., 4] = 0.5 + 0.15 * np.arange(D)                         # 15 cm apart: closer than the pair distance, rows go active
    xyz[..., 2] = 0.5 + 0.3 * np.arange(D)
    rng = np.random.default_rng(3)
    x_obs = [np.array([[*rng.uniform(-1.2, 1.2, size=2), rng.uniform(0.2, 1.5)], [0, 0, 0]]) for _ in range(n_obs)] or None
    obs_r = [0.1] * n_obs if n_obs else None
    out = {}
    for form in ("step", "fused"):
        env = mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=mds.Physics.DYN,
                             pyb_freq=100, ctrl_freq=100, num_envs=E, dtype="float32")
        env.set_trajectories(P)
        cbf = mds.DroneCBF(env, [mds.LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2)
        trk = mds.DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
        env.step(mds.torch.zeros((E, D, 4), dtype=env.dtype))
        if form == "step":
            t, hist = 0.0, []
            for k in range(steps):
                o, st = env.step_cbf_geometric(t, trk, x_obs, obs_r)
                hist.append(st.cpu().numpy().copy())
                t += env.CTRL_TIMESTEP
            hist = np.array(hist)
        else:
            slog = mds.torch.full((steps, E), -1, dtype=mds.torch.int32, device=env.device)
            o, st = env.rollout_cbf_geometric_fused(0.0, steps, trk, x_obs, obs_r, steps_per_launch=spl, status_log=slog)
            assert env.cbf_last_step_kernel() == 2
            hist = slog.cpu().numpy()
        out[form] = (o.double().cpu().numpy().copy(), hist, env.get_state(), cbf.last_iterations().cpu().numpy().copy())
        env.close()
    np.testing.assert_array_equal(out["step"][1], out["fused"][1])
    np.testing.assert_array_equal(out["step"][3], out["fused"][3])
    # (two fp32 kernels that contract FMAs differently, 40 steps of a crowded closed loop: 2.4e-4 in a body rate at D = 12)
    np.testing.assert_allclose(out["fused"][0][..., :16], out["step"][0][..., :16], rtol=0, atol=1e-4 if D in (4, 8) and E == 11 else 1e-3)
    np.testing.assert_allclose(out["fused"][2], out["step"][2], rtol=0, atol=1e-4 if D in (4, 8) and E == 11 else 1e-3)


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_cbf_step_kernel_persistent_per_step(mds, dtype):
    """mds_cbf_set_step_kernel(h, 2): step_cbf_geometric is one launch of the several-steps-per-launch kernel with one step, and
    rollout_cbf_geometric runs it at 25 steps per launch -- bitwise the fused rollout's results whatever the steps per launch; with an
    action output asked for, and on a shape the kernel does not cover, the call falls back to the QP launch + low-level launch."""
    E, D, steps = 37, 16, 30
    xyz, rpy, P = H.c2_setup(E, D, phase="c3", offset=1.5)
    P[..., 4] = 0.5 + 0.3 * np.arange(D)
    xyz[..., 2] = 0.5 + 0.3 * np.arange(D)
    x_obs = [np.array([[sx * 0.5, sy * 0.5, 0.5], [0, 0, 0]]) for sx in (-1, 1) for sy in (-1, 1)]
    obs_r = [0.1] * 4

    def make(nd=D):
        a, b, c = (xyz, rpy, P) if nd == D else H.c2_setup(E, nd, phase="c3", offset=1.5)
        env = mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=nd, initial_xyzs=a, initial_rpys=b, physics=mds.Physics.DYN,
                             pyb_freq=100, ctrl_freq=100, num_envs=E, dtype=dtype)
        env.set_trajectories(c)
        cbf = mds.DroneCBF(env, [mds.LinearizedOmegaModel(env) for _ in range(nd)], safety_radius=0.1, zscale=1.0, order=2)
        trk = mds.DroneQPTracker(cbf, num_robots=nd, xdim=9, env=env)
        env.step(mds.torch.zeros((E, nd, 4), dtype=env.dtype))
        return env, cbf, trk

    ref, _, tr = make()
    o_ref, s_ref = ref.rollout_cbf_geometric_fused(0.0, steps, tr, x_obs, obs_r, steps_per_launch=50)
    o_ref, s_ref, x_ref = o_ref.cpu().numpy().copy(), s_ref.cpu().numpy().copy(), ref.get_state()
    ref.close()

    env, _, trk = make()
    env.set_cbf_step_kernel("persistent")
    t = 0.0
    for k in range(steps):
        o, st = env.step_cbf_geometric(t, trk, x_obs, obs_r)
        assert env.cbf_last_step_kernel() == 2
        t += env.CTRL_TIMESTEP
    np.testing.assert_array_equal(o.cpu().numpy(), o_ref)
    np.testing.assert_array_equal(st.cpu().numpy(), s_ref)
    np.testing.assert_array_equal(env.get_state(), x_ref)
    env.close()

    env, _, trk = make()
    env.set_cbf_step_kernel(2)
    o, st = env.rollout_cbf_geometric(0.0, steps, trk, x_obs, obs_r)
    assert env.cbf_last_step_kernel() == 2
    np.testing.assert_array_equal(o.cpu().numpy(), o_ref)
    np.testing.assert_array_equal(st.cpu().numpy(), s_ref)
    env.close()

    env, _, trk = make(6)                       # D = 6: covered since round 4 (an env padded to 8 lanes)
    env.set_cbf_step_kernel(2)
    env.step_cbf_geometric(0.0, trk, x_obs, obs_r)
    assert env.cbf_last_step_kernel() == 2
    env.close()
    a, b, c = H.c2_setup(E, 6, phase="c3", offset=1.5)      # drag physics: not covered -> form 0, no error
    env = mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=6, initial_xyzs=a, initial_rpys=b, physics=mds.Physics.PYB_DRAG,
                         pyb_freq=100, ctrl_freq=100, num_envs=E, dtype=dtype)
    env.set_trajectories(c)
    cbf6 = mds.DroneCBF(env, [mds.LinearizedOmegaModel(env) for _ in range(6)], safety_radius=0.1, zscale=1.0, order=2)
    trk = mds.DroneQPTracker(cbf6, num_robots=6, xdim=9, env=env)
    env.step(mds.torch.zeros((E, 6, 4), dtype=env.dtype))
    env.set_cbf_step_kernel(2)
    env.step_cbf_geometric(0.0, trk, x_obs, obs_r)
    assert env.cbf_last_step_kernel() == 0
    env.rollout_cbf_geometric(0.01, 3, trk, x_obs, obs_r)
    assert env.cbf_last_step_kernel() == 0
    env.close()


def test_cbf_persistent_rollout_can_be_captured_into_a_hip_graph(mds):
    """mds_rollout_cbf_geometric_fused only enqueues (kernel launches and one device-to-device copy of the last ring slot): a caller may
    capture it into a hipGraph and replay it -- one replay of a captured 12-step call (5 + 5 + 2 steps per launch, observation ring,
    status log) == the eager call, bit for bit, from the same state."""
    torch = mds.torch
    E, D, steps = 21, 16, 12
    xyz, rpy, P = H.c2_setup(E, D, phase="c3", offset=1.5)
    P[..., 4] = 0.5 + 0.3 * np.arange(D)
    xyz[..., 2] = 0.5 + 0.3 * np.arange(D)
    x_obs = [np.array([[sx * 0.5, sy * 0.5, 0.5], [0, 0, 0]]) for sx in (-1, 1) for sy in (-1, 1)]
    obs_r = [0.1] * 4
    res = []
    for captured in (False, True):
        env = mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=mds.Physics.DYN,
                             pyb_freq=100, ctrl_freq=100, num_envs=E, dtype="float32")
        env.set_trajectories(P)
        cbf = mds.DroneCBF(env, [mds.LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2)
        trk = mds.DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
        env.step(torch.zeros((E, D, 4), dtype=env.dtype))
        cbf.configure(x_obs, obs_r)                                    # (uploads the obstacles: set-up, before any capture)
        ring = torch.full((4, E, D, 20), float("nan"), dtype=env.dtype, device=env.device)
        slog = torch.full((steps, E), -1, dtype=torch.int32, device=env.device)
        call = lambda: env.rollout_cbf_geometric_fused(0.0, steps, trk, x_obs, obs_r, steps_per_launch=5, obs_log=ring, first_slot=1, status_log=slog)
        if captured:
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side):
                    call()
            torch.cuda.current_stream().wait_stream(side)
            assert (slog.cpu().numpy() == -1).all()                    # capturing ran nothing
            g.replay()
        else:
            call()
        torch.cuda.synchronize()
        res.append((ring.cpu().numpy().copy(), slog.cpu().numpy().copy(), env.get_state(), env._obs.cpu().numpy().copy()))
        env.close()
    for a, b in zip(*res):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("dtype,tol,z,steps,E", [("float64", 1e-10, -3.0, 220, 512), ("float32", 1e-5, -3.0, 220, 512), ("float64", 1e-7, -3.0, 1000, 512),
                                                  ("float64", 1e-5, 0.5, 220, 512), ("float32", 1e-5, -3.0, 220, 16384), ("float64", 1e-4, 0.5, 220, 16384)])
def test_c4_under_scene_512_envs_against_the_c_oracle_at_every_step(mds, dtype, tol, z, steps, E):
    """The C4 headline scene on 512 envs of the bench's generator (16 drones, four spheres at z = -3) over the bench window, against the
    plain-C restatement (oracle/c_oracle.c: a second checker, fast enough for 1.8 M drone-steps): the persistent kernel's per-env
    statuses equal the oracle's at EVERY one of the 220 steps (all 0 over the window: the unique-minimiser branch), the state at step
    220 within `tol` (fp32: north_star's 1e-5).  Also in float64 over 1000 steps -- past t = 4 s every scene turns infeasible for
    most envs: 34 k infeasible env-steps, every status equal -- and on SURVEY 8d's `level` scene (z = 0.5: 23 k infeasible env-steps in
    the window, every status equal, the chaotic loop's state within 1e-5; measured 1.1e-14 / 2.7e-6 / 1.6e-10 / 1.7e-7)."""
    from oracle import c_oracle as CO
    D = 16                                                      # E = 16 384: BASELINE config 4 at its FULL size, every env, every step
    if E > 512:
        E, note = H.full_size_or_slice(E)                       # (a slice only on a host with few cores: the C oracle runs 58 M drone-steps)
        if note:
            print("[c4 vs C oracle]" + note)
    xyz, rpy, P = H.c2_setup(E, D, seed=1000, phase="c3")       # bench.py's generator and seed
    P[..., 4] = 0.5 + 0.3 * np.arange(D)
    xyz[..., 2] = 0.5 + 0.3 * np.arange(D)
    x_obs = [np.array([[sx * 0.5, sy * 0.5, z], [0, 0, 0]]) for sx in (-1, 1) for sy in (-1, 1)]
    obs_r = [0.1] * 4
    env = mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=mds.Physics.DYN,
                         pyb_freq=100, ctrl_freq=100, num_envs=E, dtype=dtype)
    env.set_trajectories(P)
    cbf = mds.DroneCBF(env, [mds.LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2, cbf_poles=np.array([-2.2, -2.4]))
    trk = mds.DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
    ref, rst, its, _ = CO.CbfLoopC(xyz, rpy, CO.cbf_params(cbf.Kcbf.reshape(-1), cbf.umax, 0.1, 1.0, x_obs, obs_r)).run(P, steps, threads=H.oracle_threads())
    assert its > 0
    if z < 0 and steps == 220:
        assert rst[20:].sum() == 0                              # the headline scene: feasible over the whole bench window
    else:
        assert rst.sum() > 0                                    # the long run / SURVEY 8d's scene: infeasible env-steps are part of the test
    env.step(mds.torch.zeros((E, D, 4), dtype=env.dtype))
    slog = mds.torch.empty((steps, E), dtype=mds.torch.int32, device=env.device)
    obs, _ = env.rollout_cbf_geometric_fused(0.0, steps, trk, x_obs, obs_r, steps_per_launch=44, status_log=slog)
    got_st = slog.cpu().numpy()
    bad_envs = np.unique(np.nonzero(got_st != rst)[1])
    print(f"[c4 vs C oracle] {dtype} z={z} steps={steps} E={E}: envs with any status difference {bad_envs.size} of {E}, infeasible env-steps {int(rst.sum())}")
    np.testing.assert_array_equal(got_st, rst)
    got = obs.double().cpu().numpy().reshape(E, D, 20)
    err = np.abs(got[..., :16] - ref[..., :16]).max()
    print(f"[c4 vs C oracle] max |state error| at step {steps}: {err:.3e}")
    assert err < tol
    env.close()


# gates of the long-horizon fp32 figures: the measurement (profiles/r04_c4_fp32_long.log) rounded up
# (measured: step 220 2.6e-6 / 2.8e-6; 511 of 512 envs with all 1000 statuses equal, the first difference at step 976; step 1000 over those envs 1.2e-2 / 1.5e-2)
C4_LONG_GATES = {"float32": dict(err220=1e-5, agree_min=500, err1000=5e-2), "float32c": dict(err220=1e-5, agree_min=500, err1000=5e-2)}


@pytest.mark.parametrize("dtype", ["float32", "float32c"])
def test_c4_under_scene_512_envs_fp32_over_1000_steps_reported(mds, dtype):
    """What fp32 keeps of north_star's "1e-5 over 1000 steps" on the CBF loop (BASELINE config 4's generator, `under` scene, 512 envs x 16
    drones, the plain-C float64 oracle at every step): the closed loop with the QP in it amplifies rounding 1e3-1e4 x (float64 itself
    ends 1.6e-10 from the oracle), and past t = 4 s the scene turns infeasible for most envs, where a status flips on a rounding error.
    REPORTED, and gated at what is measured: the state error at step 220 (inside 1e-5), the envs whose 1000 statuses all equal the
    oracle's, and the state error at step 1000 over those envs."""
    from oracle import c_oracle as CO
    E, D, steps = 512, 16, 1000
    xyz, rpy, P = H.c2_setup(E, D, seed=1000, phase="c3")
    P[..., 4] = 0.5 + 0.3 * np.arange(D)
    xyz[..., 2] = 0.5 + 0.3 * np.arange(D)
    x_obs = [np.array([[sx * 0.5, sy * 0.5, -3.0], [0, 0, 0]]) for sx in (-1, 1) for sy in (-1, 1)]
    obs_r = [0.1] * 4
    env = mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=mds.Physics.DYN,
                         pyb_freq=100, ctrl_freq=100, num_envs=E, dtype=dtype)
    env.set_trajectories(P)
    cbf = mds.DroneCBF(env, [mds.LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2, cbf_poles=np.array([-2.2, -2.4]))
    trk = mds.DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
    bpar = CO.cbf_params(cbf.Kcbf.reshape(-1), cbf.umax, 0.1, 1.0, x_obs, obs_r)
    loop = CO.CbfLoopC(xyz, rpy, bpar)
    ref220, rst220, _, _ = loop.run(P, 220, threads=H.oracle_threads())
    ref, rst2, _, _ = loop.run(P, steps - 220, t0=2.2, threads=H.oracle_threads())
    rst = np.concatenate([rst220, rst2])
    env.step(mds.torch.zeros((E, D, 4), dtype=env.dtype))
    slog = mds.torch.empty((steps, E), dtype=mds.torch.int32, device=env.device)
    o220, _ = env.rollout_cbf_geometric_fused(0.0, 220, trk, x_obs, obs_r, steps_per_launch=44, status_log=slog[:220])
    e220 = np.abs(o220.double().cpu().numpy().reshape(E, D, 20)[..., :16] - ref220[..., :16]).max()
    obs, _ = env.rollout_cbf_geometric_fused(2.2, steps - 220, trk, x_obs, obs_r, steps_per_launch=52, status_log=slog[220:])
    got_st = slog.cpu().numpy()
    agree = (got_st == rst).all(axis=0)
    first_diff = np.where(~agree, (got_st != rst).argmax(axis=0), steps)
    err_env = np.abs(obs.double().cpu().numpy().reshape(E, D, 20)[..., :16] - ref[..., :16]).reshape(E, -1).max(axis=1)
    e1000 = err_env[agree].max() if agree.any() else float("nan")
    print(f"[c4 fp32 long horizon] {dtype}: |state err| at step 220 {e220:.3e} (all statuses of the window equal: {bool((got_st[:220] == rst[:220]).all())}); "
          f"envs whose 1000 statuses all equal the oracle's: {int(agree.sum())} of {E} (first difference at step {int(first_diff.min())}, infeasible "
          f"env-steps in the oracle {int(rst.sum())}); |state err| at step 1000 over those envs {e1000:.3e}, over all envs {err_env.max():.3e}")
    g = C4_LONG_GATES[dtype]
    assert (got_st[:220] == rst[:220]).all() and e220 < g["err220"]
    assert agree.sum() >= g["agree_min"] and (not agree.any() or e1000 < g["err1000"])
    env.close()


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-10), ("float32", 1e-5)])
def test_cbftest_default_nominal_512_envs_against_the_c_oracle_at_every_step(mds, dtype, tol):
    """simulations/CBFTest.py as it runs by default -- LQROmegaController nominal (:290-293), default CBF poles (-2.2, -2.4, -2.6 -> the
    first two gains), one sphere at the lemniscate centre (:421-425) -- on 512 envs x 16 drones for 300 control steps against the plain-C
    oracle: every status of every env at every step through the persistent kernel, every drone's state at the end (measured: 5.7
    active-set iterations per env-step, 332 infeasible env-steps, all 153 600 statuses equal; state f64 1.2e-14, fp32 9.7e-7)."""
    from multidronesim_amd.control.lqr.lqr_omega_controller import LQROmegaController
    from oracle import c_oracle as CO
    E, D, steps = 512, 16, 300
    xyz, rpy, P = H.c2_setup(E, D, seed=77, phase="c3", omega=0.5)             # CBFTest.py:418: omega = 0.5
    P[..., 4] = 0.5 + 0.3 * np.arange(D)
    xyz[..., 2] = 0.5 + 0.3 * np.arange(D)
    x_obs, obs_r = [np.array([[0.0, 0.0, 0.5], [0, 0, 0]])], [0.1]
    env = mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=mds.Physics.DYN,
                         pyb_freq=100, ctrl_freq=100, num_envs=E, dtype=dtype)
    env.set_trajectories(P)
    ctrl = LQROmegaController(env, mds.LinearizedOmegaModel(env), None)
    env.set_cbf_nominal("lqr_omega")
    cbf = mds.DroneCBF(env, [mds.LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.1, zscale=1.0, order=2)
    trk = mds.DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
    ref, rst, its, _ = CO.CbfLoopC(xyz, rpy, CO.cbf_params(cbf.Kcbf.reshape(-1), cbf.umax, 0.1, 1.0, x_obs, obs_r)).run(
        P, steps, threads=H.oracle_threads(), K_lqr_omega=ctrl.K)
    env.step(mds.torch.zeros((E, D, 4), dtype=env.dtype))
    slog = mds.torch.empty((steps, E), dtype=mds.torch.int32, device=env.device)
    obs, _ = env.rollout_cbf_geometric_fused(0.0, steps, trk, x_obs, obs_r, steps_per_launch=50, status_log=slog)
    got_st = slog.cpu().numpy()
    diff_envs = np.unique(np.nonzero(got_st != rst)[1]).size
    err = np.abs(obs.double().cpu().numpy().reshape(E, D, 20)[..., :16] - ref[..., :16]).max()
    print(f"[CBFTest default nominal vs C oracle] {dtype}: {int(rst.sum())} infeasible env-steps, {its / (E * steps):.2f} iterations per env-step, "
          f"envs with any status difference {diff_envs}, max |state err| {err:.3e}")
    assert its > 0
    np.testing.assert_array_equal(got_st, rst)
    assert err < tol
    env.close()


@pytest.mark.parametrize("form", ["step", "persistent"])
@pytest.mark.parametrize("dtype,tol", [("float64", 1e-8), ("float32", 5e-3)])
def test_order3_loop_256_envs_against_the_c_oracle_at_every_step(mds, dtype, tol, form):
    """simulations/CBFTestOrd3.py's loop (yank-omega LQR nominal, order-3 ECBF with its 3 coupled QP variables per drone and the
    thrust-state box, YankOmega low level) on 256 envs x 8 drones x 200 steps against the plain-C oracle: the status of every env at
    every step, every drone's state at the end (float64: 2.7e-12, all 51 200 statuses equal; fp32: 3 borderline envs get another status
    somewhere and part ways, the other 253 agree at every step and end within 7.3e-4 -- this loop integrates its thrust state from the
    observation's RPM echo and runs 4 active-set iterations per env-step: rounding is amplified more than in the order-2 loops)."""
    from multidronesim_amd.control import LQRYankOmegaController, YankOmegaController
    from oracle import c_oracle as CO
    E, D, steps = 256, 8, 200
    xyz, rpy, P = H.c2_setup(E, D, seed=5, phase="c3", offset=3.0, omega=0.5)     # CBFTestOrd3.py:450
    xyz[..., 2] = 0.5 + 0.4 * np.arange(D)
    P[..., 4] = 0.5 + 0.4 * np.arange(D)
    x_obs, obs_r = [np.array([[0.0, 0.0, -0.3], [0, 0, 0], [0, 0, 0]])], [0.1]
    env = mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=mds.Physics.DYN,
                         pyb_freq=100, ctrl_freq=100, num_envs=E, dtype=dtype)
    env.set_trajectories(P)
    ctrl = LQRYankOmegaController(env, mds.LinearizedYankOmegaModel(env), YankOmegaController(env))
    cbf = mds.DroneCBF(env, [mds.LinearizedYankOmegaModel(env) for _ in range(D)], safety_radius=0.125, zscale=2.0, order=3,
                       cbf_poles=np.array([-3.0, -3.6, -5.6]))
    trk = mds.DroneQPTracker(cbf, order=3, num_robots=D, xdim=10, env=env)
    env.set_cbf_nominal("lqr_yank_omega")
    L = CO.CbfLoopC(xyz, rpy, CO.cbf_params(cbf.Kcbf.reshape(-1), cbf.umax, 0.125, 2.0, x_obs, obs_r, order=3), first_rpm=O.CF2P.HOVER_RPM)
    ref, rst, its, _ = L.run3(P, steps, ctrl.K, threads=H.oracle_threads())
    env.step(mds.torch.full((E, D, 4), O.CF2P.HOVER_RPM, dtype=env.dtype))
    if form == "persistent":      # k_cbf_rollout_o3 (round 4): the same loop, 37 control steps per launch (a ragged last one), statuses of every step
        slog = mds.torch.full((steps, E), -1, dtype=mds.torch.int32, device=env.device)
        gobs, st = env.rollout_cbf_geometric_fused(0.0, steps, trk, x_obs, obs_r, steps_per_launch=37, status_log=slog)
        assert env.cbf_last_step_kernel() == 2
        got_st = slog.cpu().numpy()
    else:
        t, hist = 0.0, []
        for k in range(steps):
            gobs, st = env.step_cbf_geometric(t, trk, x_obs, obs_r)
            hist.append(st.clone())
            t += env.CTRL_TIMESTEP
        got_st = mds.torch.stack(hist).cpu().numpy()
    bad = np.unique(np.nonzero(got_st != rst)[1])
    good = np.setdiff1d(np.arange(E), bad)
    err = np.abs(gobs.double().cpu().numpy().reshape(E, D, 20)[good][..., :16] - ref[good][..., :16]).max()
    print(f"[order 3 vs C oracle] {dtype} {form}: {its / (E * steps):.2f} iterations per env-step, {int(rst.sum())} infeasible env-steps, "
          f"envs with any status difference {bad.size}, max |state err| on the others {err:.3e}")
    assert its > 0
    if dtype == "float64":
        np.testing.assert_array_equal(got_st, rst)
    else:
        # fp32: an env whose QP sits on the edge of infeasibility (35 env-steps of this scene are infeasible) can get the other status,
        # and from there it flies a different control (fallback vs filtered): measured 3 envs of 256; the rest agree at every step
        assert bad.size <= 8, bad
    assert err < tol
    env.close()


@pytest.mark.parametrize("D,n_obs,E", [(7, 0, 1), (3, 1, 5), (16, 2, 9), (8, 16, 4)])
def test_order3_persistent_rollout_small_shapes(mds, D, n_obs, E):
    """k_cbf_rollout_o3 away from the 8-drone test shape: the reference's own 7 drones in one env (simulations/CBFTestOrd3.py:31), 3, and 16
    drones (48 coupled variables), with and without spheres (16 of them: 280 rows per env), an observation ring that wraps, the call
    continued by a second one -- statuses and solver iteration counts equal to the step-by-step loop's at every step, state to rounding."""
    from multidronesim_amd.control import LQRYankOmegaController, YankOmegaController
    steps = 30
    xyz, rpy, P = H.c2_setup(E, D, seed=D, phase="c3", offset=2.0, omega=0.5)
    xyz[..., 2] = 0.5 + 0.35 * np.arange(D)
    P[..., 4] = 0.5 + 0.3 * np.arange(D)
    rng = np.random.default_rng(D)
    x_obs = [np.array([[*rng.uniform(-1.0, 1.0, size=2), rng.uniform(-0.5, 0.3)], [0, 0, 0], [0, 0, 0]]) for _ in range(n_obs)] or None
    obs_r = [0.1] * n_obs if n_obs else None
    out = {}
    for form in ("step", "fused"):
        env = mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=mds.Physics.DYN,
                             pyb_freq=100, ctrl_freq=100, num_envs=E, dtype="float64")
        env.set_trajectories(P)
        LQRYankOmegaController(env, mds.LinearizedYankOmegaModel(env), YankOmegaController(env))
        cbf = mds.DroneCBF(env, [mds.LinearizedYankOmegaModel(env) for _ in range(D)], safety_radius=0.125, zscale=2.0, order=3,
                           cbf_poles=np.array([-3.0, -3.6, -5.6]))
        trk = mds.DroneQPTracker(cbf, order=3, num_robots=D, xdim=10, env=env)
        env.set_cbf_nominal("lqr_yank_omega")
        env.step(mds.torch.full((E, D, 4), O.CF2P.HOVER_RPM, dtype=env.dtype))
        if form == "step":
            t, hist, its = 0.0, [], []
            for k in range(steps):
                o, st = env.step_cbf_geometric(t, trk, x_obs, obs_r)
                hist.append(st.cpu().numpy().copy())
                its.append(cbf.last_iterations().cpu().numpy().copy())
                t += env.CTRL_TIMESTEP
            hist, ring = np.array(hist), None
        else:
            slog = mds.torch.full((steps, E), -1, dtype=mds.torch.int32, device=env.device)
            ring = mds.torch.zeros((4, E, D, 20), dtype=env.dtype, device=env.device)
            env.rollout_cbf_geometric_fused(0.0, 19, trk, x_obs, obs_r, steps_per_launch=8, status_log=slog[:19], obs_log=ring)
            assert env.cbf_last_step_kernel() == 2
            t19 = 0.0
            for _ in range(19):
                t19 += env.CTRL_TIMESTEP
            o, st = env.rollout_cbf_geometric_fused(t19, steps - 19, trk, x_obs, obs_r, steps_per_launch=8, status_log=slog[19:], obs_log=ring, first_slot=19 % 4)
            hist = slog.cpu().numpy()
            its = [cbf.last_iterations().cpu().numpy().copy()]
            np.testing.assert_array_equal(ring[(steps - 1) % 4].cpu().numpy(), o.cpu().numpy())      # the last slot of the ring is the observation returned
        out[form] = (o.double().cpu().numpy().copy(), hist, env.get_state(), its[-1])
        env.close()
    np.testing.assert_array_equal(out["step"][1], out["fused"][1])
    np.testing.assert_array_equal(out["step"][3], out["fused"][3])
    np.testing.assert_allclose(out["fused"][0][..., :16], out["step"][0][..., :16], rtol=0, atol=1e-9)
    np.testing.assert_allclose(out["fused"][2], out["step"][2], rtol=0, atol=1e-9)


def test_order3_degenerate_envs_do_not_run_to_the_iteration_cap(mds):
    """tests/golden/o3_degenerate_envs.npz: three infeasible order-3 envs whose fp32 solve used to add dependent rows past a full active set
    and cycle to the cap (5 376 iterations, 8 ms, LDS written out of bounds: found by the host emulation under UBSan) -- now status 1 after
    a few dozen iterations, in fp32 and float64, through the filter kernel."""
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "o3_degenerate_envs.npz"))
    D = 8
    for dtype in ("float32", "float64"):
        env = make_env(mds, 3, D, dtype)
        cbf = mds.DroneCBF(env, [mds.LinearizedYankOmegaModel(env) for _ in range(D)], safety_radius=0.125, zscale=2.0, order=3,
                           cbf_poles=np.array([-3.0, -3.6, -5.6]))
        trk = mds.DroneQPTracker(cbf, order=3, num_robots=D, xdim=10, env=env)
        obs = np.stack([d[f"obs{j}"] for j in range(3)])
        xdes = np.stack([d[f"xdes{j}"] for j in range(3)])
        unom = np.stack([d[f"unom{j}"] for j in range(3)])
        us, st = trk.compute_control_batched(obs, xdes, unom, [np.array([[0.0, 0.0, -3.0], [0, 0, 0], [0, 0, 0]])], [0.1])
        it = cbf.last_iterations().cpu().numpy()
        assert (st.cpu().numpy() == 1).all() and it.max() < 100, (dtype, st, it)
        np.testing.assert_allclose(us.double().cpu().numpy(), unom, atol=1e-6)
        env.close()
