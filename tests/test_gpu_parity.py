"""GPU parity tests: the HIP path (through the C-ABI of include/mds.h, via the ctypes layer)
against the float64 oracle on the same seeded inputs, and against the golden vectors minted
from the reference.  Tolerances: north_star's 1e-5 absolute per state for fp32 state; the f64
dtype must agree to ~1e-10 (same algorithm, same precision)."""
import ctypes as C
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
        pytest.fail("no HIP device: -m gpu tests need a real MI355X (there is no CPU fallback)")
    import multidronesim_amd
    multidronesim_amd.load_library()
    from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
    import types
    return types.SimpleNamespace(CtrlAviary=CtrlAviary, DroneModel=DroneModel, Physics=Physics, torch=torch)


def make_env(mds, E, D, xyz, rpy, dtype="float32", pyb=100, ctrl=100, physics=None, integrator="euler", model=None):
    return mds.CtrlAviary(drone_model=model or mds.DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy,
                          physics=physics or mds.Physics.DYN, pyb_freq=pyb, ctrl_freq=ctrl, num_envs=E, dtype=dtype,
                          integrator=integrator)


def np_obs(t):
    return t.detach().double().cpu().numpy().reshape(-1, 20)


# ---------------------------------------------------------------------------------------------
# a1-a4: physics step
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kw", [dict(), dict(integrator="rk4"), dict(physics="drag"), dict(pyb=240, ctrl=48),
                                dict(model="cf2x")])
def test_step_f64_matches_oracle_1000_steps(mds, kw):
    n = 96
    xyz, rpy, ph = H.open_loop_setup(n)
    pyb, ctrl = kw.get("pyb", 240), kw.get("ctrl", 240)
    steps = 1000 if pyb == ctrl else 200
    consts = O.CF2X if kw.get("model") == "cf2x" else O.CF2P
    ora = O.AviaryOracle(xyz, rpy, consts, pyb, ctrl, "dyn_drag" if kw.get("physics") else "dyn", kw.get("integrator", "euler"))
    env = make_env(mds, n, 1, xyz[:, None, :], rpy[:, None, :], "float64", pyb, ctrl,
                   mds.Physics.PYB_DRAG if kw.get("physics") else mds.Physics.DYN, kw.get("integrator", "euler"),
                   mds.DroneModel.CF2X if kw.get("model") == "cf2x" else None)
    np.testing.assert_allclose(np_obs(env._computeObs()), ora.obs(), atol=1e-14)
    for k in range(steps):
        a = H.open_loop_rpm(k, ora.CTRL_TIMESTEP, ph)
        obs = ora.step(a)
        gobs, r, term, trunc, info = env.step(mds.torch.as_tensor(a.reshape(n, 1, 4)))
    np.testing.assert_allclose(np_obs(gobs), obs, atol=1e-9, rtol=1e-12)
    assert (r, term, trunc, info) == (-1, False, False, {"answer": 42})
    np.testing.assert_allclose(env.get_state().reshape(n, 13),
                               np.concatenate([ora.pos, ora.quat, ora.vel, ora.rates], axis=1), atol=1e-9)
    env.close()


@pytest.mark.parametrize("dtype,gate_500,gate_1000", [("float32", 1e-5, 3e-5), ("float32c", 5e-6, 1e-5)])
def test_step_f32_open_loop(mds, dtype, gate_500, gate_1000):
    """fp32 state, uncontrolled near-hover flight at 240 Hz.  Open loop the quadrotor is a chain of integrators, so the rounding of
    the stored state grows ~t^2.5: plain fp32 holds 1e-5 for 500 steps and 1.4e-5 at 1000 (gate 3e-5); with compensated accumulation
    (MDS_F32C: two-sum inside the step, the residuals of the three body rates kept between steps: +32 B per drone-step) north_star's
    1e-5 holds over the full 1000 steps (measured 6.2e-6; with all 13 residuals, round 2's +104 B: 3.2e-6).  The closed-loop
    configs below hold 1e-5 over 1000 steps in plain fp32."""
    n = 256
    xyz, rpy, ph = H.open_loop_setup(n)
    ora = O.AviaryOracle(xyz, rpy, pyb_freq=240, ctrl_freq=240)
    env = make_env(mds, n, 1, xyz[:, None, :], rpy[:, None, :], dtype, 240, 240)
    s0 = env.get_state().reshape(n, 13)      # start both from the initial state as stored (fp32-rounded, or value + residual)
    ora.pos, ora.quat, ora.vel, ora.rates = s0[:, 0:3].copy(), s0[:, 3:7].copy(), s0[:, 7:10].copy(), s0[:, 10:13].copy()
    for k in range(1000):
        a = H.open_loop_rpm(k, ora.CTRL_TIMESTEP, ph)
        obs = ora.step(a)
        gobs, *_ = env.step(mds.torch.as_tensor(a.reshape(n, 1, 4), dtype=mds.torch.float32))
        if k == 499:
            assert np.abs(np_obs(gobs)[:, :16] - obs[:, :16]).max() < gate_500
    err = np.abs(np_obs(gobs)[:, :16] - obs[:, :16]).max()
    assert err < gate_1000, err
    np.testing.assert_allclose(np_obs(gobs)[:, 16:], obs[:, 16:], rtol=1e-7)
    st = env.get_state().reshape(n, 13)
    assert np.abs(st - np.concatenate([ora.pos, ora.quat, ora.vel, ora.rates], axis=1)).max() < gate_1000
    env.close()


def test_compensated_fp32_storage_round_trips_and_equals_fp32_paths(mds):
    """MDS_F32C bookkeeping: set_state / get_state carry value + residual for the body rates (a float64 rate survives to ~1e-14, the
    other components to fp32 rounding), reset clears the residuals; the fused geometric step, its C rollout on two chains and the
    segment-table kernel accept the dtype (closed loop within 1e-5 of the oracle over 300 steps, two-chain rollout bitwise equal to
    the step-by-step loop); so do the state-in-registers kernels, mds_step_lqr and mds_step_dslpid (round 3)."""
    torch = mds.torch
    E, D = 37, 7
    xyz, rpy, P = H.c2_setup(E, D, phase="c3")
    env = make_env(mds, E, D, xyz, rpy, "float32c")
    assert env.dtype == torch.float32
    rng = np.random.default_rng(5)
    st = rng.normal(size=(E * D, 13))
    st[:, 3:7] /= np.linalg.norm(st[:, 3:7], axis=1, keepdims=True)
    env.set_state(st)
    got = env.get_state().reshape(-1, 13)
    np.testing.assert_allclose(got[:, 10:13], st[:, 10:13], rtol=0, atol=1e-13)         # rates: value + residual
    np.testing.assert_allclose(got[:, :10], st[:, :10], rtol=0, atol=4e-7)              # the rest: fp32 values
    assert np.abs(got[:, :10] - st[:, :10]).max() > 1e-9
    env.set_trajectories(P)                                   # re-bases the local frame (origin = centres)
    np.testing.assert_allclose(env.get_state().reshape(-1, 13), st, rtol=0, atol=1e-6)      # origin is an fp32 value
    env.reset()
    np.testing.assert_array_equal(env.get_state().reshape(-1, 13)[:, 10:13], 0.0)         # rates and their residuals cleared
    env.close()
    env = make_env(mds, E, D, xyz, rpy, "float32c")           # (a fresh env: positions carry no residual, so env and b must be set up alike)
    env.set_trajectories(P)
    obs_o, _ = H.oracle_closed_loop(xyz, rpy, P, 300)
    env.step(torch.zeros((E, D, 4), dtype=env.dtype))
    b = make_env(mds, E, D, xyz, rpy, "float32c")
    b.set_trajectories(P)
    b.step(torch.zeros((E, D, 4), dtype=b.dtype))
    b.set_rollout_streams(2)
    t = 0.0
    for k in range(300):
        g = env.step_geometric(t)
        t += env.CTRL_TIMESTEP
    assert np.abs(np_obs(g)[:, :16] - obs_o[:, :16]).max() < 1e-5
    r = b.rollout_geometric(0.0, 300, obs_every_step=True)
    assert b.last_rollout_streams() == 2
    np.testing.assert_array_equal(r.cpu().numpy(), g.cpu().numpy())
    np.testing.assert_array_equal(b.get_state(), env.get_state())
    # the state-in-registers rollout continues the same loop: 50 more steps in one launch against 50 more single steps
    for k in range(50):
        g = env.step_geometric(t)
        t += env.CTRL_TIMESTEP
    t300 = 0.0
    for _ in range(300):
        t300 += b.CTRL_TIMESTEP
    r2, _ = b.rollout_geometric_fused(t300, 50)
    assert np.abs(np_obs(r2)[:, :16] - np_obs(g)[:, :16]).max() < 2e-6
    np.testing.assert_allclose(b.get_state(), env.get_state(), rtol=0, atol=2e-6)
    env.close()
    b.close()
    # mds_step_lqr and mds_step_dslpid on the compensated dtype: same closed loops as fp32 to rounding, rate residuals kept
    from multidronesim_amd.control import LQRController
    from multidronesim_amd.model import LinearizedModel
    outs = {}
    for dt in ("float32", "float32c"):
        e2 = make_env(mds, E, D, xyz, rpy, dt)
        e2.set_trajectories(P)
        LQRController(e2, LinearizedModel(e2))
        e2.step(torch.zeros((E, D, 4), dtype=e2.dtype))
        t = 0.0
        for k in range(100):
            o = e2.step_lqr(t)
            t += e2.CTRL_TIMESTEP
        outs[dt] = np_obs(o).copy()
        e2.close()
    assert np.isfinite(outs["float32c"]).all() and np.abs(outs["float32c"][:, :16] - outs["float32"][:, :16]).max() < 1e-4
    tp = xyz + np.array([0.0, 0.0, 0.5])
    for dt in ("float32", "float32c"):
        e3 = make_env(mds, E, D, xyz, rpy, dt, pyb=240, ctrl=240)
        for k in range(100):
            o = e3.step_dslpid(tp, np.zeros_like(tp))
        outs[dt] = np_obs(o).copy()
        e3.close()
    assert np.isfinite(outs["float32c"]).all() and np.abs(outs["float32c"][:, :16] - outs["float32"][:, :16]).max() < 1e-4
    with pytest.raises(RuntimeError):
        make_env(mds, 2, 2, *H.c2_setup(2, 2)[:2], "float32c", physics=mds.Physics.PYB_GND)


@pytest.mark.parametrize("dtype,atol_v,rtol_w", [("float64", 1e-9, 1e-9), ("float32", 2e-3, 2e-4)])
def test_step_accelerations_match_the_reference_tree(mds, dtype, atol_v, rtol_w):
    """a3 against the reference's own files (tests/golden/dyn_wrench_accel.npz, minted from utils/model_conversions.py:69-83 and
    model/dynamics.py:83-106 with the env's constants): after ONE physics substep of k_step, (v' - v) / dt and (w' - w) / dt are
    the reference's v_dot and w_dot, RPM beyond [0, MAX_RPM] included.  fp32: the quotient amplifies the state's rounding by
    1 / dt = 240 (|v| ~ 2 -> 2.4e-7 * 240)."""
    d = np.load(os.path.join(G, "dyn_wrench_accel.npz"))
    n = d["rpm"].shape[0]
    env = make_env(mds, 1, n, d["pos"], np.zeros((n, 3)), dtype, 240, 240)
    st = np.hstack([d["pos"], d["quat"], d["vel"], d["rates"]])
    env.set_state(st)
    s0 = env.get_state().reshape(n, 13)
    obs, *_ = env.step(mds.torch.as_tensor(d["rpm"], dtype=env.dtype, device=env.device).reshape(1, n, 4))
    s1 = env.get_state().reshape(n, 13)
    dt = 1.0 / 240
    np.testing.assert_allclose((s1[:, 7:10] - s0[:, 7:10]) / dt, d["v_dot"], rtol=0, atol=atol_v)
    wd = (s1[:, 10:13] - s0[:, 10:13]) / dt
    assert np.abs(wd - d["w_dot"]).max() <= rtol_w * np.abs(d["w_dot"]).max()
    np.testing.assert_allclose(np_obs(obs)[:, 16:20], np.clip(d["rpm"], 0, float(d["max_rpm"])), rtol=1e-6 if dtype == "float32" else 1e-14)
    env.close()


@pytest.mark.parametrize("dtype,atol", [("float64", 1e-12), ("float32", 2e-6)])
def test_step_attitude_update_is_the_flow_of_the_reference_kinematics(mds, dtype, atol):
    """[UPSTREAM] _integrateQ in k_step against the kinematics the reference tree states (model/dynamics.py:62-66, :102: body-frame
    rates, R_dot = R hat(w)): after one 240 Hz substep from the attitudes and rates of tests/golden/attitude_flow.npz (|w| up to
    ~900 rad/s), R(q') = R(q) expm(hat(w') dt) with w' the kernel's own NEW body rate (the update order: q moves with the new
    omega), and q' stays a unit quaternion."""
    from scipy.linalg import expm
    d = np.load(os.path.join(G, "attitude_flow.npz"))
    n = d["quat"].shape[0]
    pos = np.zeros((n, 3))
    env = make_env(mds, 1, n, pos, np.zeros((n, 3)), dtype, 240, 240)
    env.set_state(np.hstack([pos, d["quat"], np.zeros((n, 3)), d["w"]]))
    s0 = env.get_state().reshape(n, 13)
    env.step(mds.torch.full((1, n, 4), float(O.CF2P.HOVER_RPM), dtype=env.dtype, device=env.device))
    s1 = env.get_state().reshape(n, 13)
    hat = lambda w: np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    want = np.array([O.quat_to_rotmat_scipy(q) @ expm(hat(w) / 240) for q, w in zip(s0[:, 3:7], s1[:, 10:13])])
    np.testing.assert_allclose(O.quat_to_rotmat_scipy(s1[:, 3:7]), want, rtol=0, atol=atol)
    np.testing.assert_allclose(np.linalg.norm(s1[:, 3:7], axis=1), 1.0, atol=atol)
    assert np.abs(s1[:, 10:13] - s0[:, 10:13]).max() > 1e-3      # w x Jw moved the rates: w' is not the fixture's w
    env.close()


def test_step_clips_action_and_reference_shapes(mds):
    """E=1 with NumPy in -> reference shapes: obs [D,20]; RPM clipped to [0, MAX_RPM] lands in obs[16:20]."""
    D = 3
    xyz = np.array([[0, 0, 1.0], [1, 0, 1.0], [0, 1, 1.0]])
    env = make_env(mds, 1, D, xyz, np.zeros((D, 3)), "float64", 240, 240)
    a = np.array([[-5.0, 1e6, 100.0, 200.0]] * D)
    obs, *_ = env.step(a)
    assert isinstance(obs, np.ndarray) and obs.shape == (D, 20)
    np.testing.assert_allclose(obs[:, 16:20], [[0.0, O.CF2P.MAX_RPM, 100.0, 200.0]] * D, rtol=1e-15)
    ora = O.AviaryOracle(xyz, np.zeros((D, 3)), pyb_freq=240, ctrl_freq=240)
    np.testing.assert_allclose(obs, ora.step(a), atol=1e-12)
    for name in ("M", "G", "L", "KF", "KM", "MAX_RPM", "MAX_THRUST", "HOVER_RPM", "MAX_XY_TORQUE", "MAX_Z_TORQUE", "GRAVITY"):
        np.testing.assert_allclose(getattr(env, name), getattr(O.CF2P, name), rtol=1e-14)
    assert env.CTRL_TIMESTEP == 1 / 240 and env.J.shape == (3, 3)
    o2, info = env.reset()
    np.testing.assert_allclose(o2[:, 0:3], xyz)
    np.testing.assert_allclose(o2[:, 16:20], 0)
    assert env.pos.shape == (D, 3) and env.getDroneIds().shape == (D,)
    env.render()
    env.close()
    with pytest.raises(ValueError):
        make_env(mds, 1, 1, np.zeros((1, 3)), np.zeros((1, 3)), pyb=240, ctrl=100)


def test_compute_obs_last_rpm_tracking(mds):
    """[UPSTREAM] _computeObs() after a step carries the last clipped action.  The library keeps that copy when the env
    is used the reference's way (one env), with DYN_DRAG, or on request; a batched DYN env skips the 16 B per
    drone-step and _computeObs() then reports NaN in those columns instead of stale values."""
    D = 2
    xyz = np.array([[0, 0, 1.0], [1, 0, 1.0]])
    hover = O.CF2P.HOVER_RPM
    for E, kw, tracked in ((1, {}, True), (3, {}, False), (3, dict(track_last_rpm=True), True), (3, dict(physics=mds.Physics.PYB_DRAG), True)):
        env = mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=np.zeros((D, 3)),
                             physics=kw.pop("physics", mds.Physics.DYN), pyb_freq=100, ctrl_freq=100, num_envs=E, dtype="float32", **kw)
        assert (np_obs(env._computeObs())[:, 16:20] == 0).all()                      # after reset: zeros, always
        act = mds.torch.full((E, D, 4), hover * 1.01, dtype=env.dtype)
        act[..., 1] = 1e6
        obs, *_ = env.step(act)
        want = np_obs(obs)
        got = np_obs(env._computeObs())
        np.testing.assert_array_equal(got[:, :16], want[:, :16])
        if tracked:
            np.testing.assert_array_equal(got[:, 16:20], want[:, 16:20])
            env.set_trajectories(np.tile(np.array([1.0, 1.0, 0, 0, 1.0, 0, 0]), (E, D, 1)))
            o2 = np_obs(env.step_geometric(0.0))                                     # fused kernel keeps the copy too
            np.testing.assert_array_equal(np_obs(env._computeObs())[:, 16:20], o2[:, 16:20])
            o3 = np_obs(env.rollout_geometric_fused(0.01, 5)[0])                     # and the multi-step kernel
            np.testing.assert_array_equal(np_obs(env._computeObs())[:, 16:20], o3[:, 16:20])
        else:
            assert np.isnan(got[:, 16:20]).all()
            env.reset()
            assert (np_obs(env._computeObs())[:, 16:20] == 0).all()
        env.close()


@pytest.mark.parametrize("n", [1, 63, 64, 65, 255, 257, 1000])
def test_ragged_sizes(mds, n):
    """n not a multiple of the wave (64) / workgroup (256): tail lanes and the LDS obs staging."""
    xyz, rpy, ph = H.open_loop_setup(n, seed=n)
    ora = O.AviaryOracle(xyz, rpy, pyb_freq=240, ctrl_freq=240)
    env = make_env(mds, n, 1, xyz[:, None, :], rpy[:, None, :], "float64", 240, 240)
    guard = mds.torch.full((n + 8, 20), 777.0, dtype=mds.torch.float64, device=env.device)   # canary after the obs rows
    env._obs = guard[:n].reshape(n, 1, 20)
    for k in range(5):
        a = H.open_loop_rpm(k, ora.CTRL_TIMESTEP, ph)
        obs = ora.step(a)
        gobs, *_ = env.step(mds.torch.as_tensor(a.reshape(n, 1, 4)))
    np.testing.assert_allclose(np_obs(gobs), obs, atol=1e-12)
    assert (guard[n:] == 777.0).all()
    env.close()


# ---------------------------------------------------------------------------------------------
# golden vectors through the C-ABI (a5, a7-a10)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,tol_rel,tol_aux", [("float64", 1e-12, 1e-12), ("float32", 2e-6, 4e-6)])
def test_geometric_compute_golden(mds, dtype, tol_rel, tol_aux):
    from multidronesim_amd.control.geometric import GeometricControl
    d = np.load(os.path.join(G, "geometric_compute.npz"))
    n = d["obs"].shape[0]
    env = make_env(mds, n, 1, np.zeros((n, 1, 3)), np.zeros((n, 1, 3)), dtype)
    ctrl = GeometricControl(env)
    rpm, force, w_des, R_des = ctrl.compute_batched(d["obs"].reshape(n, 1, 20), d["des"].reshape(n, 1, 11), return_omegas=True)
    rpm = rpm.double().cpu().numpy().reshape(n, 4)
    assert np.abs(rpm / d["rpm"] - 1).max() < tol_rel
    assert np.abs(force.double().cpu().numpy().reshape(n) - d["force"]).max() < tol_aux
    assert np.abs(w_des.double().cpu().numpy().reshape(n, 3) - d["w_des"]).max() < 5 * tol_aux
    assert np.abs(R_des.double().cpu().numpy().reshape(n, 3, 3) - d["R_des"]).max() < tol_aux
    rpm_only = ctrl.compute_batched(d["obs"].reshape(n, 1, 20), d["des"].reshape(n, 1, 11)).double().cpu().numpy()
    np.testing.assert_array_equal(rpm_only.reshape(n, 4), rpm)
    env.close()


def test_reference_signatures_single_drone(mds):
    """GeometricControl(env).compute(obs) and Lemniscate(...)(t) with the reference's call shapes,
    on the SURVEY.md 8c spot values (incl. the no-op-transpose quirk)."""
    from multidronesim_amd.control.geometric import GeometricControl
    from multidronesim_amd.trajectories.Lemniscate import Lemniscate
    d = np.load(os.path.join(G, "geometric_compute.npz"))
    env = make_env(mds, 1, 2, np.zeros((2, 3)), np.zeros((2, 3)), "float64")
    traj = Lemniscate(center=np.array([0, 0, .5]), omega=1.5, yaw_rate=0.3)
    pos, vel, acc, yaw, om = traj(0.37)
    np.testing.assert_allclose(pos, [0.3505205637, 0.6651959776, 0.5], atol=1e-9)
    np.testing.assert_allclose(vel, [0.1534443003, -1.3181327684, 0], atol=1e-9)
    np.testing.assert_allclose(acc, [-4.0263528439, 0.2337315706, 0], atol=1e-9)
    np.testing.assert_allclose([yaw, om], [0.3480011356, 0.9366776206], atol=1e-9)
    assert traj.get_total_time() == pytest.approx(2 * np.pi / 1.5)
    ctrl = GeometricControl(env)
    ctrl.set_desired_trajectory(0, pos, vel, acc, yaw, om)
    rpm = ctrl.compute(d["spot_obs"])
    np.testing.assert_allclose(rpm, [14426.96568, 14788.12966, 11884.78050, 16030.20029], atol=1e-4)
    force, w_des, R_des = ctrl.compute(d["spot_obs"], return_omegas=True)
    np.testing.assert_allclose(w_des, [0.5956908439, -0.0106877434, 0.7451911107], atol=1e-9)
    assert R_des.shape == (3, 3)
    env.close()


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-12), ("float32", 1e-5)])
def test_lemniscate_eval_golden(mds, dtype, tol):
    g = np.load(os.path.join(G, "lemniscate.npz"))
    env = make_env(mds, 1, 3, np.zeros((3, 3)), np.zeros((3, 3)), dtype)
    env.set_trajectories(g["params"])
    des = mds.torch.zeros((3, 11), dtype=env.dtype, device=env.device)
    from multidronesim_amd._device import stream_ptr
    for i, t in enumerate(g["ts"]):
        rc = env._lib.mds_lemniscate_eval(env._h, C.c_double(float(t)), C.c_void_p(des.data_ptr()), C.c_void_p(stream_ptr(env.device)))
        assert rc == 0
        assert np.abs(des.double().cpu().numpy() - g["out"][:, i]).max() < tol
    env.close()


@pytest.mark.parametrize("dtype,rtol", [("float64", 1e-12), ("float32", 1e-6)])
def test_mixer_golden(mds, dtype, rtol):
    from multidronesim_amd.utils.model_conversions import action_to_input, input_to_action
    d = np.load(os.path.join(G, "mixer.npz"))
    n = d["u"].shape[0]
    env = make_env(mds, n, 1, np.zeros((n, 1, 3)), np.zeros((n, 1, 3)), dtype)
    np.testing.assert_allclose(input_to_action(env, d["u"]), d["rpm"], rtol=rtol)
    np.testing.assert_allclose(action_to_input(env, d["act"]), d["u_back"], rtol=rtol * 20, atol=1e-9 if dtype == "float32" else 1e-18)
    np.testing.assert_allclose(action_to_input(env, d["act"], cap_rpm=False), d["u_back_nocap"], rtol=rtol * 20,
                               atol=1e-9 if dtype == "float32" else 1e-18)
    np.testing.assert_allclose(input_to_action(env, d["u"][0]), d["rpm"][0], rtol=rtol)   # (4,) reference shape
    env.close()


def test_quadrotor_dynamics_golden(mds):
    from multidronesim_amd.model.dynamics import QuadrotorDynamics
    d = np.load(os.path.join(G, "dynamics_deriv.npz"))
    q = QuadrotorDynamics(sim_freq=100)
    np.testing.assert_allclose(q.dynamics(0.0, d["state"], d["u"]), d["out_hb"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(q.dynamics(0.0, d["state"][0], d["u"][0]), d["out_hb"][0], rtol=1e-12, atol=1e-12)

    class EnvLike:
        M, G, KF, PYB_FREQ = float(d["env_m"]), float(d["env_g"]), 3.16e-10, 100
        J = np.diag([2.3951e-5, 2.3951e-5, 3.2347e-5])
    q.load_env_params(EnvLike)     # stale Hummingbird J quirk
    np.testing.assert_allclose(q.dynamics(0.0, d["state"], d["u_env"]), d["out_env"], rtol=1e-12, atol=1e-12)
    f32 = q.dynamics(0.0, mds.torch.as_tensor(d["state"], dtype=mds.torch.float32), mds.torch.as_tensor(d["u_env"], dtype=mds.torch.float32))
    np.testing.assert_allclose(f32.double().cpu().numpy(), d["out_env"], rtol=2e-5, atol=2e-5)
    with pytest.raises(ValueError):
        q.step(d["u"][0])


# ---------------------------------------------------------------------------------------------
# fused trajectory + controller + step: BASELINE configs C2 / C3 at oracle-sized batches
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("E,D,phase,dtype,tol", [(64, 4, "c2", "float32", 1e-5), (32, 8, "c3", "float32", 1e-5),
                                                 (16, 4, "c2", "float64", 1e-9), (333, 3, "c3", "float32", 1e-5)])
def test_fused_geometric_1000_steps(mds, E, D, phase, dtype, tol):
    xyz, rpy, P = H.c2_setup(E, D, phase=phase)
    obs, hist = H.oracle_closed_loop(xyz, rpy, P, 1000, record_every=250)
    env = make_env(mds, E, D, xyz, rpy, dtype)
    env.set_trajectories(P)
    env.step(mds.torch.zeros((E, D, 4), dtype=env.dtype))        # EnvGeometric.py:431
    t, k_hist = 0.0, 1
    for k in range(1000):
        gobs = env.step_geometric(t)
        t += env.CTRL_TIMESTEP
        if (k + 1) % 250 == 0:
            g = np_obs(gobs)
            assert np.abs(g[:, :16] - hist[k_hist][:, :16]).max() < tol, (k, np.abs(g[:, :16] - hist[k_hist][:, :16]).max())
            assert np.abs(g[:, 16:] / hist[k_hist][:, 16:] - 1).max() < max(tol, 2e-6)
            k_hist += 1
    env.close()


@pytest.mark.parametrize("dtype,tol,rtol_rpm", [("float64", 1e-8, 1e-9), ("float32", 1e-5, 5e-6), ("float32c", 1e-5, 5e-6)])
def test_fused_loop_against_the_reference_objects_in_the_loop(mds, dtype, tol, rtol_rpm):
    """The fused kernel (trajectory -> GeometricControl -> mixer -> DYN step) over 1000 control steps against
    tests/golden/closed_loop_ref_in_loop.npz, a rollout in which trajectory sampling and controller were the reference's OWN
    Lemniscate / GeometricControl objects (only env.step was the oracle's): every 50th observation within north_star's 1e-5 (fp32)."""
    d = np.load(os.path.join(G, "closed_loop_ref_in_loop.npz"))
    P, every = d["params"], int(d["every"])
    D = P.shape[0]
    env = make_env(mds, 1, D, d["xyz"], np.zeros((D, 3)), dtype)
    env.set_trajectories(P)
    obs, *_ = env.step(mds.torch.zeros((1, D, 4), dtype=env.dtype, device=env.device))
    np.testing.assert_allclose(np_obs(obs), d["obs_log"][0], rtol=0, atol=tol)
    t = 0.0
    for i in range(int(d["steps"])):
        obs, act = env.step_geometric(t, return_action=True)
        t += env.CTRL_TIMESTEP
        if (i + 1) % every == 0:
            k = (i + 1) // every
            g = np_obs(obs)
            assert np.abs(g[:, :16] - d["obs_log"][k][:, :16]).max() < tol, (i + 1)
            np.testing.assert_allclose(g[:, 16:], d["obs_log"][k][:, 16:], rtol=rtol_rpm)
            np.testing.assert_allclose(act.double().cpu().numpy().reshape(D, 4), d["action_log"][k - 1], rtol=rtol_rpm)
    env.close()


def test_fused_matches_unfused_operator_chain(mds):
    """step_geometric == lemniscate_eval -> geometric_compute -> step, operator by operator."""
    from multidronesim_amd.control.geometric import GeometricControl
    from multidronesim_amd._device import stream_ptr
    E, D = 8, 4
    xyz, rpy, P = H.c2_setup(E, D, yaw_rate=0.2)
    a = make_env(mds, E, D, xyz, rpy, "float64")
    b = make_env(mds, E, D, xyz, rpy, "float64")
    a.set_trajectories(P)
    b.set_trajectories(P)
    ctrl = GeometricControl(b)
    des = mds.torch.zeros((E, D, 11), dtype=b.dtype, device=b.device)
    oa, *_ = a.step(mds.torch.zeros((E, D, 4), dtype=a.dtype))
    ob, *_ = b.step(mds.torch.zeros((E, D, 4), dtype=b.dtype))
    t = 0.0
    for k in range(50):
        oa, act_a = a.step_geometric(t, return_action=True)
        assert b._lib.mds_lemniscate_eval(b._h, C.c_double(t), C.c_void_p(des.data_ptr()), C.c_void_p(stream_ptr(b.device))) == 0
        rpm = ctrl.compute_batched(ob, des)
        np.testing.assert_allclose(act_a.cpu().numpy(), rpm.cpu().numpy(), rtol=1e-9)
        ob, *_ = b.step(rpm)
        t += a.CTRL_TIMESTEP
    np.testing.assert_allclose(oa.cpu().numpy(), ob.cpu().numpy(), atol=1e-9)
    a.close()
    b.close()


def test_rollout_equals_stepwise_and_substeps_drag_rk4(mds):
    E, D = 8, 4
    xyz, rpy, P = H.c2_setup(E, D)
    for kw in (dict(), dict(pyb=200, ctrl=100), dict(physics="drag"), dict(integrator="rk4")):
        phys = mds.Physics.PYB_DRAG if kw.get("physics") else mds.Physics.DYN
        envs = [make_env(mds, E, D, xyz, rpy, "float64", kw.get("pyb", 100), kw.get("ctrl", 100), phys, kw.get("integrator", "euler"))
                for _ in range(2)]
        for e in envs:
            e.set_trajectories(P)
            e.step(mds.torch.zeros((E, D, 4), dtype=e.dtype))
        t = 0.0
        for k in range(40):
            o1 = envs[0].step_geometric(t)
            t += 0.01
        o2 = envs[1].rollout_geometric(0.0, 40)
        np.testing.assert_allclose(o1.cpu().numpy(), o2.cpu().numpy(), atol=1e-13)
        obs, _ = H.oracle_closed_loop(xyz, rpy, P, 40, kw.get("pyb", 100), kw.get("ctrl", 100),
                                      "dyn_drag" if kw.get("physics") else "dyn", kw.get("integrator", "euler"))
        np.testing.assert_allclose(np_obs(o1), obs, atol=1e-9, rtol=1e-11)
        for e in envs:
            e.close()


@pytest.mark.parametrize("dtype,physics", [("float32", "dyn"), ("float64", "drag"), ("float16", "dyn")])
def test_two_stream_rollout_is_bit_identical_and_stream_ordered(mds, dtype, physics):
    """mds_set_rollout_streams(2): the two halves of the shard step on two internal streams.  Same kernel per drone, so the
    result must equal the one-stream rollout bit for bit -- with an odd number of 256-drone batches and a ragged tail
    (7 x 199 = 1393 drones = 5 full batches + 113), on a non-default caller stream, reading obs right after the call."""
    torch = mds.torch
    E, D, T = 199, 7, 37
    xyz, rpy, P = H.c2_setup(E, D, phase="c3", yaw_rate=0.2)
    phys = mds.Physics.PYB_DRAG if physics == "drag" else mds.Physics.DYN
    out = []
    for streams in (1, 2):
        env = make_env(mds, E, D, xyz, rpy, dtype, 200, 100, phys)
        env.set_trajectories(P)
        env.set_rollout_streams(streams)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            env.step(torch.zeros((E, D, 4), dtype=env.dtype, device=env.device))
            o = env.rollout_geometric(0.0, T, obs_every_step=True)
            snap = o.clone()                                   # ordered behind both halves by the exit events
            assert env.last_rollout_streams() == streams          # the library reports what it did
            o2 = env.rollout_geometric(T * env.CTRL_TIMESTEP, 3).clone()   # a second call chains behind the first
            assert env.rollout_streams_for(1) == 1 and env.rollout_streams_for(T) == streams
        side.synchronize()
        out.append((snap.cpu().numpy(), o2.cpu().numpy(), env.get_state()))
        env.close()
    for a, b in zip(*out):
        assert np.isfinite(a.astype(np.float64)).all()
        np.testing.assert_array_equal(a, b)
    env2 = make_env(mds, 2, 2, *H.c2_setup(2, 2)[:2], "float32")
    with pytest.raises(Exception):
        env2.set_rollout_streams(3)
    # auto policy: a small shard never splits, whatever the call length; the thread's current device is left alone
    assert env2.last_rollout_streams() == 0 and env2.rollout_streams_for(100000) == 1
    assert torch.cuda.current_device() == env2.device.index
    env2.close()


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-11), ("float32", 1e-5)])
def test_rollout_launch_form_policy_and_equivalence(mds, dtype, tol):
    """mds_set_rollout_form: form 1 = one launch per control step (bit-identical to step_geometric calls), form 2 = the whole-rollout
    kernel in launches of steps_per_launch control steps with every step's observation still written; auto picks form 2 from 2^13
    drones and 8 steps on and form 1 elsewhere.  Same arithmetic:
    the two forms agree to rounding and both match the oracle; a call in form 2 continues a call in form 1 (t accumulates step by step)."""
    torch = mds.torch
    E, D, T = 1024, 8, 120                       # 8 192 drones: the lower edge of the auto window; 120 = 50 + 50 + 20 steps
    xyz, rpy, P = H.c2_setup(E, D, phase="c3")
    outs = {}
    for form in (0, 1):
        env = make_env(mds, E, D, xyz, rpy, dtype)
        env.set_trajectories(P)
        env.step(torch.zeros((E, D, 4), dtype=env.dtype, device=env.device))
        env.set_rollout_form(form)
        assert env.rollout_form_for(T) == (2 if form == 0 else 1) and env.rollout_form_for(7) == 1 and env.last_rollout_form() == 0
        o = env.rollout_geometric(0.0, T, obs_every_step=True).clone()
        assert env.last_rollout_form() == (2 if form == 0 else 1) and env.last_rollout_streams() == 1
        o2 = env.rollout_geometric(T * env.CTRL_TIMESTEP, 33, obs_every_step=False).clone()      # last observation only, chunk of 33
        outs[form] = (o.double().cpu().numpy(), o2.double().cpu().numpy(), env.get_state())
        env.close()
    for a, b in zip(outs[0], outs[1]):
        assert np.isfinite(a).all()
        np.testing.assert_allclose(a[..., :16] if a.shape[-1] == 20 else a, b[..., :16] if b.shape[-1] == 20 else b, atol=tol)
    # form 2 rewrites ONE [n, 20] array step after step (default-policy stores); the same kernel writing every step's rows to their own slot of a
    # [T, n, 20] log (non-temporal stores) must leave, bit for bit, the same last rows and the same state
    env = make_env(mds, E, D, xyz, rpy, dtype)
    env.set_trajectories(P)
    env.step(torch.zeros((E, D, 4), dtype=env.dtype, device=env.device))
    last, log = env.rollout_geometric_fused(0.0, T, log=True)
    np.testing.assert_array_equal(log[-1].double().cpu().numpy(), outs[0][0])
    np.testing.assert_array_equal(last.double().cpu().numpy(), outs[0][0])
    assert np.abs(log[T // 2].double().cpu().numpy() - outs[0][0]).max() > 1e-3          # (the log's slots are different steps)
    env.close()
    np.testing.assert_allclose(outs[0][0][..., 16:], outs[1][0][..., 16:], rtol=max(tol, 1e-9))
    idx = np.arange(0, E, 16)
    oobs, _ = H.oracle_closed_loop(xyz[idx], rpy[idx], P[idx], T)
    assert np.abs(outs[0][0][idx].reshape(-1, 20)[:, :16] - oobs[:, :16]).max() < (1e-9 if dtype == "float64" else 1e-5)
    # forced form 2 on a ragged little shard (3 x 5 = 15 drones), 7 steps per launch, against form 1 bit for bit in the TIME it hands
    # the kernel: 23 steps = 7 + 7 + 7 + 2, then the step-by-step loop continues both
    xs, rs, Ps = H.c2_setup(3, 5, phase="c3")
    # ... also with the drag model (the kernel carries the previous step's RPM), two substeps per control step, and (fp32) the
    # compensated dtype, whose rate residuals cross the launches
    variants = [dict(), dict(physics=mds.Physics.PYB_DRAG, pyb=200)] + ([dict(dt="float32c")] if dtype == "float32" else [])
    for kw in variants:
        res = []
        for form in (2, 1):
            env = make_env(mds, 3, 5, xs, rs, kw.get("dt", dtype), kw.get("pyb", 100), 100, kw.get("physics"))
            env.set_trajectories(Ps)
            env.step(torch.zeros((3, 5, 4), dtype=env.dtype, device=env.device))
            env.set_rollout_form(form, 7)
            env.rollout_geometric(0.0, 23, obs_every_step=True)
            assert env.last_rollout_form() == form
            env.rollout_geometric(23 * env.CTRL_TIMESTEP, 9, want_obs=False)             # no observation wanted at all
            res.append(env.step_geometric(32 * env.CTRL_TIMESTEP).double().cpu().numpy())
            env.close()
        np.testing.assert_allclose(res[0][..., :16], res[1][..., :16], atol=tol, err_msg=str(kw))
    # fp16 storage never leaves form 1 under the auto policy (form 2 rounds the state once per launch, not once per step)
    env = make_env(mds, E, D, xyz, rpy, "float16")
    env.set_trajectories(P)
    assert env.rollout_form_for(1000) == 1
    with pytest.raises(Exception):
        env.set_rollout_form(3)
    env.close()
    # a quarter of a million drones: the state stays in registers (form 2) whatever the precision
    Eb = 32768
    xb, rb, Pb = H.c2_setup(Eb, D, phase="c3")
    env = make_env(mds, Eb, D, xb, rb, dtype)
    assert env.rollout_form_for(20) == 2 and env.rollout_form_for(7) == 1
    env.close()


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_fused_rollout_equals_stepwise_and_logs_every_step(mds, dtype):
    """mds_rollout_geometric_fused (one launch, state in registers) == n calls of mds_step_geometric
    (same templates; two kernels may differ in FMA contraction, so to rounding: 1e-5 / 1e-12; both also match the oracle), and its
    [T,E,D,20] log holds every intermediate observation; ragged n (3 x 85 = 255)."""
    atol = 1e-5 if dtype == "float32" else 1e-12
    E, D, T = 85, 3, 60
    xyz, rpy, P = H.c2_setup(E, D, phase="c3", yaw_rate=0.3)
    a = make_env(mds, E, D, xyz, rpy, dtype)
    b = make_env(mds, E, D, xyz, rpy, dtype)
    for e in (a, b):
        e.set_trajectories(P)
        e.step(mds.torch.zeros((E, D, 4), dtype=e.dtype))
    t, step_obs = 0.0, []
    for k in range(T):
        step_obs.append(a.step_geometric(t).clone())
        t += a.CTRL_TIMESTEP
    last, log = b.rollout_geometric_fused(0.0, T, log=True)
    assert log.shape == (T, E, D, 20)
    ref = mds.torch.stack(step_obs)
    assert (log[..., :16] - ref[..., :16]).abs().max().item() < atol
    assert ((log[..., 16:] - ref[..., 16:]).abs() / ref[..., 16:]).max().item() < atol
    assert mds.torch.equal(last, log[-1])
    np.testing.assert_allclose(a.get_state(), b.get_state(), atol=atol)
    oobs, _ = H.oracle_closed_loop(xyz, rpy, P, T)
    assert np.abs(np_obs(last)[:, :16] - oobs[:, :16]).max() < (1e-5 if dtype == "float32" else 1e-9)
    last2, none = b.rollout_geometric_fused(T * b.CTRL_TIMESTEP, 5)          # no log, continues from the stored state
    for k in range(5):
        o = a.step_geometric(t)
        t += a.CTRL_TIMESTEP
    assert none is None and (last2[..., :16] - o[..., :16]).abs().max().item() < atol
    a.close()
    b.close()


def test_wind_force_matches_oracle(mds):
    """EnvGeometric.py:34,463-467: constant +x world force of 2.5e-4 N on every drone, every step."""
    E, D = 8, 2
    xyz, rpy, P = H.c2_setup(E, D, offset=1.0)
    wind = np.array([2.5e-4, 0.0, 0.0])
    env = make_env(mds, E, D, xyz, rpy, "float64")
    env.set_trajectories(P)
    env.set_wind(wind)
    n = E * D
    Pf = P.reshape(-1, 7)
    ora = O.AviaryOracle(xyz.reshape(-1, 3), rpy.reshape(-1, 3), pyb_freq=100, ctrl_freq=100)
    ora.wind = wind
    obs = ora.step(np.zeros((n, 4)))
    env.step(mds.torch.zeros((E, D, 4), dtype=env.dtype))
    t = 0.0
    for k in range(200):
        pos, vel, acc, yaw, yd = O.lemniscate(t, Pf[:, 0], Pf[:, 1], Pf[:, 2:5], Pf[:, 5], Pf[:, 6])
        obs = ora.step(O.geometric_compute(obs, pos, vel, acc, yaw, yd))
        gobs = env.step_geometric(t)
        t += env.CTRL_TIMESTEP
    np.testing.assert_allclose(np_obs(gobs), obs, atol=1e-9, rtol=1e-11)
    calm = make_env(mds, E, D, xyz, rpy, "float64")
    calm.set_trajectories(P)
    calm.step(mds.torch.zeros((E, D, 4), dtype=calm.dtype))
    assert (calm.rollout_geometric(0.0, 200) - gobs)[..., 0].abs().max().item() > 1e-4     # the wind does something
    env.close()
    calm.close()


def test_fp16_storage_is_stable_and_close(mds):
    """fp16 state storage / fp32 arithmetic (config 5) is a throughput configuration: 2^-11
    relative storage rounding.  Gate: stays finite, unit quaternion, tracks the oracle to 5e-2
    over 100 closed-loop steps."""
    E, D = 16, 2
    xyz, rpy, P = H.c2_setup(E, D, offset=0.0)
    obs, _ = H.oracle_closed_loop(xyz, rpy, P, 100)
    env = make_env(mds, E, D, xyz, rpy, "float16")
    env.set_trajectories(P)
    env.step(mds.torch.zeros((E, D, 4), dtype=env.dtype))
    t = 0.0
    for k in range(100):
        gobs = env.step_geometric(t)
        t += 0.01
    g = np_obs(gobs)
    assert np.isfinite(g).all()
    assert np.abs(np.linalg.norm(g[:, 3:7], axis=1) - 1).max() < 2e-3
    assert np.abs(g[:, :3] - obs[:, :3]).max() < 5e-2
    env.close()


# ---------------------------------------------------------------------------------------------
# full BASELINE size (C3: 65 536 x 8) through size-independent properties
# ---------------------------------------------------------------------------------------------
def test_c3_full_size_properties(mds):
    E, D = 65536, 8
    xyz, rpy, P = H.c2_setup(E, D, phase="c3")
    # (1) replicate invariance: envs 0..255 are copied into envs 1024..1279 -> bitwise equal outputs
    xyz[1024:1280], P[1024:1280] = xyz[0:256], P[0:256]
    env = make_env(mds, E, D, xyz, rpy, "float32")
    env.set_trajectories(P)
    env.step(mds.torch.zeros((E, D, 4), dtype=env.dtype))
    steps = 200
    obs = env.rollout_geometric(0.0, steps).clone()
    assert mds.torch.equal(obs[0:256], obs[1024:1280])
    # (2) a small batch holding only envs 0..63 gives bitwise the same rows (no cross-drone coupling)
    small = make_env(mds, 64, D, xyz[:64], rpy[:64], "float32")      # same kernel variant as the big run
    small.set_trajectories(P[:64])
    small.step(mds.torch.zeros((64, D, 4), dtype=small.dtype))
    so = small.rollout_geometric(0.0, steps)
    assert mds.torch.equal(so, obs[:64])
    # (3) invariants on every row: finite, unit quaternion, RPM inside the clip range
    assert mds.torch.isfinite(obs).all()
    assert (obs[..., 3:7].norm(dim=-1) - 1).abs().max().item() < 1e-5
    assert obs[..., 16:].min().item() >= 9440.3 * (1 - 1e-6) and obs[..., 16:].max().item() <= O.CF2P.MAX_RPM * (1 + 1e-6)
    # (4) a strided sample of 512 envs against the oracle
    idx = np.arange(0, E, E // 512)
    oobs, _ = H.oracle_closed_loop(xyz[idx], rpy[idx], P[idx], steps)
    g = obs[idx].double().cpu().numpy().reshape(-1, 20)
    assert np.abs(g[:, :16] - oobs[:, :16]).max() < 1e-5
    env.close()
    small.close()


# ---------------------------------------------------------------------------------------------
# error behaviour of the C-ABI (no throw, status codes)
# ---------------------------------------------------------------------------------------------
def test_capi_error_codes(mds):
    from multidronesim_amd import _capi as capi
    from multidronesim_amd import MdsError
    lib = capi.load_library()
    env = make_env(mds, 2, 2, np.zeros((2, 3)), np.zeros((2, 3)))
    with pytest.raises(MdsError) as ei:
        env.step_geometric(0.0)                     # no trajectory attached
    assert ei.value.status == -5
    buf = mds.torch.zeros(4 * 20 + 4, dtype=mds.torch.float32, device=env.device)
    act = mds.torch.zeros((4, 4), dtype=mds.torch.float32, device=env.device)
    assert lib.mds_step(env._h, C.c_void_p(act.data_ptr()), C.c_void_p(buf.data_ptr() + 4), None) == -4   # misaligned obs
    assert lib.mds_step(env._h, None, None, None) == -1
    assert lib.mds_step(env._h, C.c_void_p(act.data_ptr()), None, None) == 0                               # obs optional
    with pytest.raises(ValueError):
        env.step(np.zeros((3, 4)))
    env.close()
    with pytest.raises(MdsError):
        env.step(np.zeros((2, 4)))


@pytest.mark.parametrize("dtype,atol", [("float64", 1e-13), ("float32", 2e-6)])
def test_obs_to_model_adapters(mds, dtype, atol):
    """utils/model_conversions.py obs_to_lin_model (dim 9/10/12), obs_to_geo_model, calc_z_thrust through mds_obs_to_model,
    against the oracle (itself pinned on the reference's outputs in tests/golden/geometric_compute.npz)."""
    from multidronesim_amd.utils import calc_z_thrust, obs_to_geo_model, obs_to_lin_model
    rng = np.random.default_rng(3)
    n = 37
    obs = rng.normal(size=(n, 20))
    obs[:, 3:7] *= rng.uniform(0.5, 2.0, size=(n, 1))                 # un-normalised quaternions: Rotation.from_quat normalises
    obs[:, 16:20] = O.CF2P.HOVER_RPM * (1 + 0.1 * rng.normal(size=(n, 4)))
    env = make_env(mds, n, 1, np.zeros((1, 3)), np.zeros((1, 3)), dtype)
    for dim in (9, 10, 12):
        got = obs_to_lin_model(obs, dim, env)
        ref = O.obs_to_lin_model(obs, dim)
        np.testing.assert_allclose(got, ref, atol=atol * max(1.0, np.abs(ref).max()))
    g18 = obs_to_geo_model(obs, env)
    assert g18.shape == (n, 18)
    np.testing.assert_allclose(g18[:, 3:12].reshape(n, 3, 3), O.quat_to_rotmat_scipy(obs[:, 3:7]), atol=atol * 10)
    np.testing.assert_allclose(g18[:, [0, 1, 2, 12, 13, 14, 15, 16, 17]], obs[:, [0, 1, 2, 10, 11, 12, 13, 14, 15]], atol=atol * 10)
    np.testing.assert_allclose(calc_z_thrust(env, obs[3]), O.CF2P.KF * np.sum(obs[3, 16:20] ** 2), rtol=max(atol, 1e-12) * 10)
    env.close()
    d = np.load(os.path.join(G, "mixer.npz"))                                    # the reference's own outputs (mint_golden.py)
    n = d["obs"].shape[0]
    env = make_env(mds, n, 1, np.zeros((1, 3)), np.zeros((1, 3)), dtype)
    for dim, key in ((9, "lin9"), (10, "lin10"), (12, "lin12")):
        np.testing.assert_allclose(obs_to_lin_model(d["obs"], dim, env), d[key], atol=atol * max(1.0, np.abs(d[key]).max()))
    np.testing.assert_allclose(obs_to_geo_model(d["obs"], env), d["geo18"], atol=atol * 10 * max(1.0, np.abs(d["geo18"]).max()))
    obs = d["obs"]
    t = mds.torch.tensor(obs[:5], dtype=env.dtype, device=env.device)            # device tensor in -> device tensor out
    assert obs_to_lin_model(t, 9, env).is_cuda and tuple(obs_to_lin_model(t, 9, env).shape) == (5, 9)
    with pytest.raises(ValueError):
        obs_to_lin_model(obs, 9)
    env.close()


@pytest.mark.parametrize("physics,name", [("PYB_GND", "dyn_gnd"), ("PYB_DW", "dyn_dw"), ("PYB_GND_DRAG_DW", "dyn_gnd_drag_dw")])
@pytest.mark.parametrize("dtype,tol", [("float64", 1e-10), ("float32", 3e-5)])
def test_ground_effect_and_downwash_match_oracle(mds, physics, name, dtype, tol):
    """[UPSTREAM] _groundEffect / _downwash as extra terms of the DYN step (env.step only): envs of 5 drones stacked above
    each other near the ground, 3 physics substeps per control step (env-mates' positions refresh between substeps), CF2P and
    CF2X propeller geometry, one drone rolled past 90 degrees (ground effect off there).  Spec-level: oracle == kernel."""
    for model, consts in ((mds.DroneModel.CF2P, O.CF2P), (mds.DroneModel.CF2X, O.CF2X)):
        E, D = 7, 5
        rng = np.random.default_rng(9)
        xyz = np.zeros((E, D, 3))
        xyz[..., 0:2] = rng.normal(size=(E, D, 2)) * 0.05
        xyz[..., 2] = 0.03 + 0.35 * np.arange(D) + rng.uniform(0, 0.02, size=(E, D))
        rpy = rng.uniform(-0.3, 0.3, size=(E, D, 3))
        rpy[0, 1, 0] = 1.8                                                     # |roll| > pi/2
        env = mds.CtrlAviary(drone_model=model, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=getattr(mds.Physics, physics),
                             pyb_freq=300, ctrl_freq=100, num_envs=E, dtype=dtype)
        ora = O.AviaryOracle(xyz.reshape(-1, 3), rpy.reshape(-1, 3), consts, 300, 100, physics=name, drones_per_env=D)
        ph = rng.uniform(0, 2 * np.pi, size=(E * D, 4))
        ph[:, 1:] = rng.uniform(-1, 1, size=(E * D, 3))
        for k in range(40):
            a = H.open_loop_rpm(k, 0.01, ph, hover=consts.HOVER_RPM)
            obs, *_ = env.step(mds.torch.tensor(a.reshape(E, D, 4), dtype=env.dtype, device=env.device))
            oobs = ora.step(a)
        g = np_obs(obs)
        assert np.abs(g[:, :16] - oobs[:, :16]).max() < tol * max(1.0, np.abs(oobs[:, :16]).max())
        plain = O.AviaryOracle(xyz.reshape(-1, 3), rpy.reshape(-1, 3), consts, 300, 100, physics="dyn")
        for k in range(40):
            pobs = plain.step(H.open_loop_rpm(k, 0.01, ph, hover=consts.HOVER_RPM))
        assert np.abs(pobs[:, 2] - oobs[:, 2]).max() > 0.01                     # the effects are doing something in this scene
        np.testing.assert_allclose(env.get_state()[..., 0:3].reshape(-1, 3), g[:, 0:3], atol=tol * 10)   # double-buffer bookkeeping
        env.close()
    with pytest.raises(RuntimeError):
        mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=2, initial_xyzs=np.zeros((2, 3)), initial_rpys=np.zeros((2, 3)),
                       physics=mds.Physics.PYB_DW, pyb_freq=100, ctrl_freq=100, integrator="rk4")


def _stacked_near_ground(E, D, seed=9):
    """envs of D drones stacked above each other just over the floor, each on its own small Lemniscate around its column"""
    rng = np.random.default_rng(seed)
    cen = np.zeros((E, D, 3))
    cen[..., 0:2] = rng.uniform(-2, 2, size=(E, 1, 2)) + rng.normal(size=(E, D, 2)) * 0.03
    cen[..., 2] = 0.06 + 0.3 * np.arange(D)
    xyz = cen + np.concatenate([rng.normal(size=(E, D, 2)) * 0.05, rng.uniform(0, 0.02, size=(E, D, 1))], axis=-1)
    rpy = rng.uniform(-0.2, 0.2, size=(E, D, 3))
    P = np.zeros((E, D, 7))
    P[..., 0], P[..., 1], P[..., 2:5], P[..., 5] = 0.25, 1.2, cen, 0.15
    P[..., 6] = 2 * np.pi * np.arange(D) / (D + 0.25)
    return xyz, rpy, P


@pytest.mark.parametrize("physics,name", [("PYB_GND", "dyn_gnd"), ("PYB_DW", "dyn_dw"), ("PYB_GND_DRAG_DW", "dyn_gnd_drag_dw")])
@pytest.mark.parametrize("dtype,tol", [("float64", 1e-9), ("float32", 3e-5)])
def test_ground_effect_and_downwash_in_the_controller_paths(mds, physics, name, dtype, tol):
    """[UPSTREAM] _groundEffect / _downwash under the fused controller paths (k_step_ctrl_env: trajectory sample -> GeometricControl
    -> first substep, the action replayed by the second substep through k_step_env), stacked scene near the ground, 200 Hz physics /
    100 Hz control, against the oracle's closed loop with the same physics; the C rollout loop and the whole-rollout entry point
    issue the same steps (bitwise), the latter logging every observation."""
    E, D, steps = 6, 5, 60
    xyz, rpy, P = _stacked_near_ground(E, D)
    Pf = P.reshape(-1, 7)
    ora = O.AviaryOracle(xyz.reshape(-1, 3), rpy.reshape(-1, 3), O.CF2P, 200, 100, physics=name, drones_per_env=D)
    plain = O.AviaryOracle(xyz.reshape(-1, 3), rpy.reshape(-1, 3), O.CF2P, 200, 100, physics="dyn")
    oobs, pobs = ora.step(np.zeros((E * D, 4))), plain.step(np.zeros((E * D, 4)))
    envs = [mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=getattr(mds.Physics, physics),
                           pyb_freq=200, ctrl_freq=100, num_envs=E, dtype=dtype) for _ in range(3)]
    for e in envs:
        e.set_trajectories(P)
        e.step(mds.torch.zeros((E, D, 4), dtype=e.dtype))
    t, step_obs = 0.0, []
    for k in range(steps):
        pos, vel, acc, yaw, yd = O.lemniscate(t, Pf[:, 0], Pf[:, 1], Pf[:, 2:5], Pf[:, 5], Pf[:, 6])
        oobs = ora.step(O.geometric_compute(oobs, pos, vel, acc, yaw, yd))
        pobs = plain.step(O.geometric_compute(pobs, pos, vel, acc, yaw, yd))
        gobs, act = envs[0].step_geometric(t, return_action=True)
        step_obs.append(gobs.clone())
        t += 0.01
    g = np_obs(gobs)
    assert np.abs(g[:, :16] - oobs[:, :16]).max() < tol * max(1.0, np.abs(oobs[:, :16]).max())
    np.testing.assert_allclose(g[:, 16:], oobs[:, 16:], rtol=1e-5 if dtype == "float32" else 1e-10)
    assert np.abs(pobs[:, :3] - oobs[:, :3]).max() > 1e-3                      # the effects are doing something in this closed loop
    np.testing.assert_allclose(envs[0].get_state()[..., 0:3].reshape(-1, 3), g[:, 0:3], atol=tol * 10)   # double-buffer bookkeeping
    r = envs[1].rollout_geometric(0.0, steps, obs_every_step=True)
    assert envs[1].last_rollout_streams() == 1
    np.testing.assert_array_equal(r.cpu().numpy(), gobs.cpu().numpy())
    last, log = envs[2].rollout_geometric_fused(0.0, steps, log=True)
    np.testing.assert_array_equal(log.cpu().numpy(), mds.torch.stack(step_obs).cpu().numpy())
    np.testing.assert_array_equal(last.cpu().numpy(), gobs.cpu().numpy())
    for e in envs:
        e.close()


def test_ground_effect_under_the_lqr_and_cbf_paths(mds):
    """The same physics modes under mds_step_lqr (12-state LQRController) and under the CBF loop (nominal -> QP -> ThrustOmega low
    level -> first substep + replay), float64, against the oracle loops."""
    from multidronesim_amd.control import LQRController
    from multidronesim_amd.model.linearized import LinearizedModel
    from multidronesim_amd.cbf.cbf import DroneCBF
    from multidronesim_amd.cbf.qptracker import DroneQPTracker
    from multidronesim_amd.model.linear_omega import LinearizedOmegaModel
    E, D, steps = 4, 5, 50
    xyz, rpy, P = _stacked_near_ground(E, D, seed=3)
    Pf = P.reshape(-1, 7)
    env = mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=mds.Physics.PYB_GND_DRAG_DW,
                         pyb_freq=200, ctrl_freq=100, num_envs=E, dtype="float64")
    env.set_trajectories(P)
    LQRController(env, LinearizedModel(env))
    ora = O.AviaryOracle(xyz.reshape(-1, 3), rpy.reshape(-1, 3), O.CF2P, 200, 100, physics="dyn_gnd_drag_dw", drones_per_env=D)
    K = O.lqr12_gain(O.CF2P)
    oobs = ora.step(np.zeros((E * D, 4)))
    env.step(mds.torch.zeros((E, D, 4), dtype=env.dtype))
    t = 0.0
    for k in range(steps):
        pos, vel, acc, yaw, yd = O.lemniscate(t, Pf[:, 0], Pf[:, 1], Pf[:, 2:5], Pf[:, 5], Pf[:, 6])
        act, _ = O.lqr12_compute(oobs, pos, vel, yaw, yd, K)
        oobs = ora.step(act)
        gobs = env.step_lqr(t)
        t += 0.01
    assert np.abs(np_obs(gobs)[:, :16] - oobs[:, :16]).max() < 1e-7
    env.close()
    # CBF loop near the ground, one sphere beside the columns
    x_obs, obs_r = [np.array([[0.4, 0.0, 0.4], [0, 0, 0]])], [0.05]
    env = mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=mds.Physics.PYB_GND,
                         pyb_freq=200, ctrl_freq=100, num_envs=E, dtype="float64")
    env.set_trajectories(P)
    cbf = DroneCBF(env, [LinearizedOmegaModel(env) for _ in range(D)], safety_radius=0.05, zscale=1.0, order=2)
    trk = DroneQPTracker(cbf, num_robots=D, xdim=9, env=env)
    oobs, ohist = H.oracle_cbf_closed_loop(xyz, rpy, P, 40, cbf.Kcbf.reshape(-1), cbf.umax, 0.05, 1.0, x_obs, obs_r, pyb_freq=200, ctrl_freq=100,
                                           physics="dyn_gnd")
    env.step(mds.torch.zeros((E, D, 4), dtype=env.dtype))
    t = 0.0
    for k in range(40):
        gobs, st = env.step_cbf_geometric(t, trk, x_obs, obs_r)
        np.testing.assert_array_equal(st.cpu().numpy(), ohist[k])
        t += 0.01
    assert np.abs(gobs.double().cpu().numpy()[..., :16] - oobs[..., :16]).max() < 1e-6
    env.close()


def test_set_origin_rebases_the_local_frame_without_moving_anything(mds):
    """mds_set_origin: the per-drone local-frame origin (fp32 conditioning) can be moved at any time; world-frame state and
    observations stay what they were, and stepping afterwards matches an env that never moved its origin."""
    import ctypes as C2
    from multidronesim_amd import _capi as capi
    E, D = 5, 3
    rng = np.random.default_rng(12)
    xyz = rng.uniform(-20, 20, size=(E, D, 3))
    rpy = rng.uniform(-0.2, 0.2, size=(E, D, 3))
    a = make_env(mds, E, D, xyz, rpy, "float32")
    b = make_env(mds, E, D, xyz, rpy, "float32")
    act = mds.torch.full((E, D, 4), float(a.HOVER_RPM) * 1.02, dtype=a.dtype, device=a.device)
    for _ in range(5):
        a.step(act); b.step(act)
    before = a.get_state().copy()
    org = np.ascontiguousarray(before[..., 0:3].reshape(-1, 3) + rng.normal(size=(E * D, 3)) * 0.1)   # origin near each drone
    capi.check(a._lib.mds_set_origin(a._h, capi.as_double_ptr(org), a._stream()), "mds_set_origin")
    np.testing.assert_allclose(a.get_state(), before, atol=2e-6)
    np.testing.assert_allclose(np_obs(a._computeObs())[:, :16], np_obs(b._computeObs())[:, :16], atol=2e-6)
    for _ in range(20):
        oa, *_ = a.step(act); ob, *_ = b.step(act)
    np.testing.assert_allclose(np_obs(oa)[:, :16], np_obs(ob)[:, :16], atol=3e-5)      # world coordinates up to 20 m: 1 fp32 ulp = 2e-6
    a.close(); b.close()


def test_c5_full_size_properties(mds):
    """BASELINE config 5 at its full size (262 144 envs x 2 drones, fp16 state storage / fp32 arithmetic, 240 Hz, random RPM around
    hover through env.step): replicated envs bitwise equal, a small batch of the same envs bitwise equal, invariants on every
    row, and a strided sample against the float64 oracle at fp16-storage tolerance."""
    torch = mds.torch
    E, D, steps = 262144, 2, 48
    xyz, rpy, _ = H.c2_setup(E, D)
    xyz[100000:100128] = xyz[0:128]
    g = torch.Generator(device="cuda").manual_seed(7)
    acts = [(16000.0 * (1 + 0.05 * torch.randn((E, D, 4), device="cuda", generator=g))).clamp(0, 21000.0) for _ in range(4)]
    for a in acts:
        a[100000:100128] = a[0:128]

    def run(lo, hi):
        env = mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=D, initial_xyzs=xyz[lo:hi], initial_rpys=rpy[lo:hi],
                             physics=mds.Physics.DYN, pyb_freq=240, ctrl_freq=240, num_envs=hi - lo, dtype="float16")
        for k in range(steps):
            o, *_ = env.step(acts[k % 4][lo:hi].to(env.dtype))
        out = o.clone()
        env.close()
        return out

    obs = run(0, E)
    assert torch.equal(obs[0:128], obs[100000:100128])
    assert torch.equal(run(0, 128), obs[0:128])
    f = obs.float()
    assert torch.isfinite(f).all()
    assert (f[..., 3:7].norm(dim=-1) - 1).abs().max().item() < 2e-3            # fp16 storage of the quaternion
    idx = np.arange(0, E, E // 256)
    ora = O.AviaryOracle(xyz[idx].reshape(-1, 3), rpy[idx].reshape(-1, 3), O.CF2P, 240, 240)
    for k in range(steps):
        oo = ora.step(acts[k % 4][idx].to(torch.float16).double().cpu().numpy().reshape(-1, 4))
    gg = f[idx].double().cpu().numpy().reshape(-1, 20)
    # fp16 storage: 4e-3 m is one unit in the last place of a 5 m coordinate, and the stored state is re-rounded every step
    assert np.abs(gg[:, :3] - oo[:, :3]).max() < 5e-2 and np.abs(gg[:, 10:13] - oo[:, 10:13]).max() < 1e-1
    # ... and EVERY one of the 524 288 drones against the plain-C oracle on the host cores (the same fp16-rounded commands)
    from oracle import c_oracle as CO
    av = CO.AviaryC(xyz.reshape(-1, 3), rpy.reshape(-1, 3), 240, 240)
    a16 = [a.to(torch.float16).double().cpu().numpy().reshape(-1, 4) for a in acts]
    for k in range(steps):
        co = av.step(a16[k % 4])
    ga = f.double().cpu().numpy().reshape(-1, 20)
    ep, ev = np.abs(ga[:, :3] - co[:, :3]).max(), np.abs(ga[:, 10:13] - co[:, 10:13]).max()
    print(f"[C5 full size vs C oracle] {E * D} drones x {steps} steps, fp16 storage: max |pos err| {ep:.2e} m, max |vel err| {ev:.2e} m/s")
    assert ep < 5e-2 and ev < 1e-1
    np.testing.assert_allclose(co[np.repeat(idx, D) * D + np.tile(np.arange(D), idx.size)], oo, rtol=0, atol=1e-10)   # the two oracles agree on the sample


@pytest.mark.parametrize("name,physics,integrator", [("euler", "DYN", "euler"), ("rk4", "DYN", "rk4"), ("drag", "PYB_DRAG", "euler")])
def test_step_f64_matches_committed_1000_step_rollouts(mds, name, physics, integrator):
    """The oracle-only long-horizon fixture (tests/golden/dyn_rollouts_1000.npz): env.step in float64 for 1000
    steps at 240 Hz under the fixture's RPM sequence (clipped commands included) against the committed observations."""
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "dyn_rollouts_1000.npz"))
    xyz, rpy, rpm = d["xyz"], d["rpy"], d["rpm"]
    n = xyz.shape[0]
    env = mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=n, initial_xyzs=xyz, initial_rpys=rpy,
                         physics=getattr(mds.Physics, physics), pyb_freq=240, ctrl_freq=240, num_envs=1, dtype="float64",
                         integrator=integrator)
    acts = [mds.torch.as_tensor(rpm[k], dtype=mds.torch.float64, device=env.device).reshape(1, n, 4) for k in range(8)]
    check = {int(k): j for j, k in enumerate(d["check"])}
    for k in range(1, 1001):
        obs, *_ = env.step(acts[(k - 1) % 8])
        if k in check:
            np.testing.assert_allclose(np_obs(obs), d[name][check[k]], rtol=1e-9, atol=1e-9, err_msg=f"{name} step {k}")
    env.close()


@pytest.mark.parametrize("dtype,streams", [("float32", 1), ("float32", 2), ("float16", 2), ("float64", 2)])
def test_rollout_step_equals_env_step_loop(mds, dtype, streams):
    """mds_rollout_step (replayed action table, observations into a log ring; with two streams the halves of the shard are two
    step chains) against the same env.step calls: bitwise equal log rows and state.  1393 drones = 5 batches + 113."""
    torch = mds.torch
    E, D, steps, A, T = (200 if dtype == "float16" else 199), 7, 29, 3, 5     # fp16 rows: n even keeps every log slot 16-byte aligned
    xyz, rpy, _ = H.c2_setup(E, D)
    out = []
    for mode in ("loop", "rollout"):
        env = make_env(mds, E, D, xyz, rpy, dtype, 240, 120, mds.Physics.PYB_DRAG)
        acts = (env.HOVER_RPM * (1 + 0.05 * torch.randn((A, E, D, 4), device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)))).to(env.dtype)
        log = torch.zeros((T, E, D, 20), dtype=env.dtype, device="cuda")
        env.set_rollout_streams(streams)
        if mode == "loop":
            for j in range(steps):
                if j > 0 and j % 11 == 0:
                    env.reset()                                   # episode boundary: back to the initial poses (mds_reset)
                o, *_ = env.step(acts[j % A])
                log[j % T].copy_(o)
        else:
            env.rollout_step(acts, 0, steps - 4, log, episode_len=11)
            env.rollout_step(acts, steps - 4, 4, log, episode_len=11)               # continues the step counter
        out.append((log.cpu().numpy().copy(), env.get_state()))
        env.close()
    assert np.isfinite(out[0][0].astype(np.float64)).all()
    for a, b in zip(*out):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("streams", [1, 2])
def test_rollout_calls_can_be_captured_into_a_hip_graph(mds, streams):
    """The rollout entry points only enqueue (kernel launches, and for two chains event record / wait pairs that fork from and
    join back into the caller's stream), so a caller may capture them into a hipGraph (here through torch.cuda.graph) and replay
    it: 3 replays of a captured 6-step mds_rollout_step == 18 eager steps, bit for bit, one chain and two."""
    torch = mds.torch
    E, D, A, T = 199, 7, 3, 6
    xyz, rpy, _ = H.c2_setup(E, D)
    gen = torch.Generator(device="cuda").manual_seed(5)
    envs = [make_env(mds, E, D, xyz, rpy, "float32", 240, 240) for _ in range(2)]
    acts = (envs[0].HOVER_RPM * (1 + 0.05 * torch.randn((A, E, D, 4), device="cuda", generator=gen))).to(envs[0].dtype)
    logs = [torch.zeros((T, E, D, 20), dtype=envs[0].dtype, device="cuda") for _ in range(2)]
    for e in envs:
        e.set_rollout_streams(streams)
    for r in range(3):
        envs[0].rollout_step(acts, 0, 6, logs[0])                      # first_step 0 every time: what a replayed graph does
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            envs[1].rollout_step(acts, 0, 6, logs[1])
    torch.cuda.current_stream().wait_stream(side)
    assert envs[1].last_rollout_streams() == streams
    state_after_capture = envs[1].get_state()
    np.testing.assert_allclose(state_after_capture[..., 0:3], xyz, atol=1e-6)   # capturing ran nothing (fp32 storage of the initial poses)
    for r in range(3):
        g.replay()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(logs[0].cpu().numpy(), logs[1].cpu().numpy())
    np.testing.assert_array_equal(envs[0].get_state(), envs[1].get_state())
    for e in envs:
        e.close()


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-12), ("float32", 2e-5), ("float16", 5e-2)])
def test_fused_rollout_step_equals_stepwise(mds, dtype, tol):
    """mds_rollout_step_fused (several env.step per launch, state in registers, launches cut at episode boundaries) against
    mds_rollout_step: same log ring and state to rounding (fp16 storage: the fused path rounds the state once per launch)."""
    torch = mds.torch
    E, D, steps, A, T = 200, 7, 37, 3, 5
    xyz, rpy, _ = H.c2_setup(E, D)
    out = []
    for spl in (1, 8):
        env = make_env(mds, E, D, xyz, rpy, dtype, 240, 120, mds.Physics.PYB_DRAG)
        acts = (env.HOVER_RPM * (1 + 0.05 * torch.randn((A, E, D, 4), device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)))).to(env.dtype)
        log = torch.zeros((T, E, D, 20), dtype=env.dtype, device="cuda")
        env.rollout_step(acts, 0, steps - 9, log, episode_len=13, steps_per_launch=spl)
        env.rollout_step(acts, steps - 9, 9, log, episode_len=13, steps_per_launch=spl)
        out.append((log.double().cpu().numpy().copy(), env.get_state(), env._computeObs().double().cpu().numpy().copy()))
        env.close()
    for a, b in zip(*out):
        assert np.isfinite(a).all()
        scale = np.maximum(1.0, np.abs(a))
        assert (np.abs(a - b) / scale).max() < tol


@pytest.mark.parametrize("dtype", ["float32", "float64", "float16"])
def test_state_ptrs_views_match_get_state(mds, dtype):
    """mds_state_ptrs: the zero-copy strided views (+ origin) reproduce mds_get_state after a few steps, and writing through
    a view is seen by the next step."""
    E, D = 37, 5
    xyz, rpy, P = H.c2_setup(E, D)
    env = make_env(mds, E, D, xyz, rpy, dtype)
    env.set_trajectories(P)
    env.step(mds.torch.zeros((E, D, 4), dtype=env.dtype))
    for k in range(5):
        env.step_geometric(0.01 * k)
    v = env.state_views()
    st = env.get_state().reshape(-1, 13)
    got = np.stack([c.double().cpu().numpy() for c in v["comp"]], axis=1)
    got[:, :3] += np.stack([o.double().cpu().numpy() for o in v["origin"]], axis=1)
    tol = 1e-12 if dtype == "float64" else (1e-6 if dtype == "float32" else 2e-3)
    np.testing.assert_allclose(got, st, atol=tol)
    v["comp"][9].fill_(0.25)                                  # vz of every drone, in place
    st2 = env.get_state().reshape(-1, 13)
    np.testing.assert_allclose(st2[:, 9], 0.25, atol=1e-3)
    env.close()


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_observation_rpy_is_the_reference_trees_convention_incl_gimbal_branches(mds, dtype):
    """obs[7:10] of the device (euler_from_quat in csrc/mds_math.hpp, pybullet's getEulerFromQuaternion restated with its +-0.99999
    gimbal branches) on the attitudes of tests/golden/euler_convention.npz -- minted from the reference tree: random attitudes, pitch
    within 1e-6 .. 1e-2 of +-pi/2 on both sides of the branch threshold, and exactly +-pi/2.  float64: the rpy of the fixture to 1e-12
    (same branch everywhere).  float32: away from the branches 5e-6 rad (wrapped); at the branches rounding the quaternion to fp32 may
    land an attitude on the other side of the threshold, where rpy jumps although the attitude does not -- there the check is the one
    that matters to the consumers: rpy_to_rot(rpy) (utils/model_conversions.py:4-19) gives the attitude of the quaternion to 5e-3,
    the bound of pybullet's own branch approximation."""
    from scipy.spatial.transform import Rotation
    d = np.load(os.path.join(G, "euler_convention.npz"))
    q, rpy_ref, g = d["quat"], d["rpy"], d["gimbal"]
    n = len(q)
    env = make_env(mds, n, 1, np.zeros((n, 1, 3)), np.zeros((n, 1, 3)), dtype=dtype)
    st = np.zeros((n, 13))
    st[:, 3:7] = q
    env.set_state(st)
    obs = np_obs(env._computeObs())
    np.testing.assert_allclose(obs[:, 3:7], q, rtol=0, atol=1e-15 if dtype == "float64" else 6e-8)
    rpy = obs[:, 7:10]
    if dtype == "float64":
        np.testing.assert_allclose(rpy, rpy_ref, rtol=0, atol=1e-12)
        assert (rpy[g][:, 0] == 0).all() and g.sum() >= 48
    else:
        wrap = lambda a: (a + np.pi) % (2 * np.pi) - np.pi
        far = np.abs(np.abs(d["euler_in"][:, 1]) - np.pi / 2) > 2e-2              # well away from the threshold (4.47e-3)
        assert far.sum() >= 190
        assert np.abs(wrap(rpy[far] - rpy_ref[far])).max() < 5e-6
        R = Rotation.from_euler("xyz", rpy).as_matrix()                              # == rpy_to_rot (fixture: R_rpy == R_rpy_scipy)
        assert np.abs(R - d["R_quat"]).max() < 5e-3
        assert np.abs(R[far] - d["R_quat"][far]).max() < 5e-6
    env.close()


@pytest.mark.parametrize("pyb,ctrl", [(240, 240), (240, 80), (240, 120)])
def test_ground_effect_steps_can_be_captured_and_replayed(mds, pyb, ctrl):
    """The ground-effect / downwash modes step a double-buffered state and flip the handle's buffers on the host per substep.  A
    captured call bakes the buffers of the moment into its graph: with an odd number of substeps per call (pyb == ctrl: one; 240 / 80:
    three) every replay would read the same stale buffer and the state would never advance.  Under capture such a call ends with a
    copy back into the buffer it started from: 5 replays of a captured mds_step == 5 eager steps, bit for bit (even counts too)."""
    torch = mds.torch
    E, D = 23, 5
    xyz, rpy, _ = H.c2_setup(E, D)
    xyz[..., 2] = 0.05 + 0.2 * np.arange(D)                    # stacked near the ground: both effects act
    envs = [make_env(mds, E, D, xyz, rpy, "float32", pyb, ctrl, mds.Physics.PYB_GND_DRAG_DW) for _ in range(2)]
    act = (envs[0].HOVER_RPM * (1 + 0.03 * torch.randn((E, D, 4), device="cuda", generator=torch.Generator(device="cuda").manual_seed(9)))).to(envs[0].dtype)
    obs = [torch.zeros((E, D, 20), dtype=envs[0].dtype, device="cuda") for _ in range(2)]
    lib, Cc = envs[0]._lib, C

    def step(e, o, stream):
        rc = lib.mds_step(e._h, Cc.c_void_p(act.data_ptr()), Cc.c_void_p(o.data_ptr()), Cc.c_void_p(stream.cuda_stream))
        assert rc == 0, rc
    for r in range(5):
        step(envs[0], obs[0], torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            step(envs[1], obs[1], side)
    torch.cuda.current_stream().wait_stream(side)
    for r in range(5):
        g.replay()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(obs[0].cpu().numpy(), obs[1].cpu().numpy())
    np.testing.assert_array_equal(envs[0].get_state(), envs[1].get_state())
    assert np.abs(envs[0].get_state()[..., 0:3] - xyz).max() > 1e-4          # the state did advance
    # eager calls after the replays continue from the replayed state
    step(envs[0], obs[0], torch.cuda.current_stream())
    step(envs[1], obs[1], torch.cuda.current_stream())
    torch.cuda.synchronize()
    np.testing.assert_array_equal(obs[0].cpu().numpy(), obs[1].cpu().numpy())
    for e in envs:
        e.close()


def test_fused_loop_f64_matches_the_plain_c_oracle_1000_steps(mds):
    """The second, separately written checker (oracle/c_oracle.c: plain C, closed-form mixer inverse, scalar branches) against the
    fused HIP loop in float64 over 1000 control steps -- the C3 generator, non-trivial initial attitudes -- and fp32 at north_star's 1e-5."""
    from oracle import c_oracle as CO
    E, D = 16, 8
    xyz, rpy, P = H.c2_setup(E, D, seed=11, phase="c3")
    rpy = np.random.default_rng(1).uniform(-0.2, 0.2, size=rpy.shape)
    ref, _ = CO.AviaryC(xyz, rpy, 100, 100).geometric_loop(P, 1000)
    for dtype, tol in (("float64", 1e-9), ("float32", 1e-5)):
        env = make_env(mds, E, D, xyz, rpy, dtype)
        env.set_trajectories(P)
        env.step(mds.torch.zeros((E, D, 4), dtype=env.dtype, device=env.device))
        t = 0.0
        for _ in range(1000):
            obs = env.step_geometric(t)
            t += env.CTRL_TIMESTEP
        got = obs.double().cpu().numpy().reshape(-1, 20)
        assert np.abs(got[:, :16] - ref[:, :16]).max() < tol, dtype
        env.close()


@pytest.mark.parametrize("E,D,phase,name", [(4096, 4, "c2", "C2"), (65536, 8, "c3", "C3")])
def test_baseline_configs_2_and_3_at_full_size_every_drone_against_the_c_oracle(mds, E, D, phase, name):
    """BASELINE configs 2 and 3 at their FULL size (4 096 x 4 and 65 536 x 8 drones), T = 1000 control steps, fp32: every
    drone's final observation against the plain-C float64 oracle (oracle/c_oracle.c on the host cores: 0.5 G drone-steps for C3) --
    north_star's 1e-5 on the 13 state components of all 524 288 drones, not of a sample -- through the C rollout loop the bench times
    (two half-shard chains at C3) and, on a 1/16 slice, the 50-steps-per-launch kernel."""
    from oracle import c_oracle as CO
    steps = 1000
    xyz, rpy, P = H.c2_setup(E, D, seed=1000, phase=phase)      # bench.py's generator and seed
    Ec, note = H.full_size_or_slice(E)                           # (every env on a GPU box; a slice on a host with few cores)
    ref, _ = CO.AviaryC(xyz[:Ec].reshape(-1, 3), rpy[:Ec].reshape(-1, 3), 100, 100).geometric_loop(P[:Ec].reshape(-1, 7), steps, threads=H.oracle_threads())
    env = make_env(mds, E, D, xyz, rpy, "float32")
    env.set_trajectories(P)
    env.step(mds.torch.zeros((E, D, 4), dtype=env.dtype, device=env.device))
    obs = env.rollout_geometric(0.0, steps, want_obs=True, obs_every_step=True)
    got = obs.double().cpu().numpy().reshape(-1, 20)[:Ec * D]
    err = np.abs(got[:, :16] - ref[:, :16])
    rel_rpm = np.abs(got[:, 16:] / ref[:, 16:] - 1).max()
    print(f"[{name} full size vs C oracle] {Ec * D} drones x {steps} steps{note}: max |state err| {err.max():.3e} (mean {err.mean():.1e}), rpm rel {rel_rpm:.1e}")
    assert err.max() < 1e-5 and rel_rpm < 1e-5
    env.close()
    Es = min(E // 16, Ec)
    env = make_env(mds, Es, D, xyz[:Es], rpy[:Es], "float32")
    env.set_trajectories(P[:Es])
    env.step(mds.torch.zeros((Es, D, 4), dtype=env.dtype, device=env.device))
    t = 0.0
    for _ in range(steps // 50):
        last = env.rollout_geometric_fused(t, 50)
        last = last[0] if isinstance(last, tuple) else last
        t += 50 * env.CTRL_TIMESTEP
    g2 = last.double().cpu().numpy().reshape(-1, 20)
    assert np.abs(g2[:, :16] - ref[:Es * D, :16]).max() < 1e-5
    env.close()
