"""GPU parity of the call site of QuadrotorDynamics.dynamics -- simulations/CompareModels.py:46-56 -- and the helpers around it,
through the C-ABI (mds_compare_models, mds_linear_xdot, mds_rpy_to_rot, mds_geo_model_to_obs) and the Python mirrors, against the
reference-minted tests/golden/compare_models.npz and the float64 oracle."""
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


def small_env(mds, dtype="float32", E=1, D=2):
    return mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=D, initial_xyzs=np.zeros((E, D, 3)), initial_rpys=np.zeros((E, D, 3)),
                          physics=mds.Physics.DYN, pyb_freq=100, ctrl_freq=100, num_envs=E, dtype=dtype)


def rel(got, ref):
    return float((np.abs(got - ref) / (1 + np.abs(ref))).max())


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-12), ("float32", 3e-5)])
def test_compare_models_golden(mds, dtype, tol):
    """The three arrays of the CompareModels loop on the reference's 320 rows (RPM above MAX_RPM, at 0 and below; either quaternion
    sign; the stale Hummingbird J on the geometric side), NumPy in (float64 on the GPU) and device tensors in `dtype`."""
    from multidronesim_amd.model import LinearizedModel, QuadrotorDynamics
    from multidronesim_amd.simulations.CompareModels import compare_models
    d = np.load(os.path.join(G, "compare_models.npz"))
    env = small_env(mds, dtype)
    lin = LinearizedModel(env)
    np.testing.assert_array_equal(lin.A, d["A"])
    np.testing.assert_array_equal(lin.B, d["B"])
    np.testing.assert_array_equal(lin.Ahat, d["Ahat"])
    np.testing.assert_array_equal(lin.Bhat, d["Bhat"])
    geo = QuadrotorDynamics(env.PYB_FREQ)
    geo.load_env_params(env)
    assert geo.m == float(d["dyn_m"]) and geo.g == float(d["dyn_g"]) and np.allclose(np.diag(geo.J), d["dyn_J"])
    tdt = getattr(mds.torch, dtype)
    obs_t = mds.torch.as_tensor(d["obs"], dtype=tdt, device=env.device)
    a, b, c = compare_models(lin, geo, obs_t)
    assert a.dtype == tdt and a.is_cuda and a.shape == (320, 12)
    for got, ref in ((a, d["xdot_lin"]), (b, d["xdot_geo"]), (c, d["x_lin"])):
        assert rel(got.double().cpu().numpy(), ref) < tol
    # NumPy in -> float64 arithmetic whatever the env's dtype, like the reference
    a64, b64, c64 = compare_models(lin, geo, d["obs"])
    assert isinstance(a64, np.ndarray) and a64.dtype == np.float64
    assert rel(a64, d["xdot_lin"]) < 1e-12 and rel(b64, d["xdot_geo"]) < 1e-12 and rel(c64, d["x_lin"]) == 0.0
    # the per-observation reference calls (one row; a [T, D, 20] history), and they keep working after env.close() like the reference's
    env.close()
    np.testing.assert_allclose(lin.calc_xdot_from_obs(d["obs"][5]), d["xdot_lin"][5], rtol=1e-12, atol=1e-12)
    hist = d["obs"].reshape(40, 8, 20)
    np.testing.assert_allclose(lin.calc_xdot_from_obs(hist), d["xdot_lin"].reshape(40, 8, 12), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(lin.calc_xdot(d["x_free"], d["obs"][:, 16:]), d["xdot_free"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(lin.calc_xdot(d["x_free"][3], d["obs"][3, 16:]), d["xdot_free"][3], rtol=1e-12, atol=1e-12)
    lin.A, lin.B = lin.Ahat, lin.Bhat                  # a caller that swaps in the 'noisy' pair
    np.testing.assert_allclose(lin.calc_xdot(d["x_free"], d["obs"][:, 16:]), d["xdot_free_hat"], rtol=1e-12, atol=1e-12)


def test_compare_models_helpers_golden(mds):
    from multidronesim_amd.utils import model_conversions as mc
    d = np.load(os.path.join(G, "compare_models.npz"))
    np.testing.assert_allclose(mc.rpy_to_rot(d["rpy"]), d["R_of_rpy"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(mc.rpy_to_rot(d["rpy"][0]), d["R_of_rpy"][0], rtol=0, atol=1e-14)
    r32 = mc.rpy_to_rot(mds.torch.as_tensor(d["rpy"], dtype=mds.torch.float32, device="cuda"))
    assert r32.shape == (128, 3, 3) and r32.dtype == mds.torch.float32
    np.testing.assert_allclose(r32.double().cpu().numpy(), d["R_of_rpy"], rtol=0, atol=5e-7)
    np.testing.assert_allclose(mc.geo_model_to_obs(d["x18"]), d["obs16"], rtol=0, atol=1e-14)     # all four from_matrix branches, signs included
    o32 = mc.geo_model_to_obs(mds.torch.as_tensor(d["x18"], dtype=mds.torch.float32, device="cuda"))
    np.testing.assert_allclose(o32.double().cpu().numpy(), d["obs16"], rtol=0, atol=5e-7)
    np.testing.assert_array_equal(mc.geo_x_dot_to_linear(np.arange(12.0)), [3, 4, 5, 9, 10, 11, 6, 7, 8, 0, 1, 2])
    t = mds.torch.arange(24.0, device="cuda").reshape(2, 12)
    assert mc.geo_x_dot_to_linear(t)[1].tolist() == [15, 16, 17, 21, 22, 23, 18, 19, 20, 12, 13, 14]
    # the 9- and 10-state models' calc_xdot is broken in the reference (12-long state against their A): same error here
    from multidronesim_amd.model import LinearizedOmegaModel, LinearizedYankOmegaModel
    env = small_env(mds)
    for cls in (LinearizedOmegaModel, LinearizedYankOmegaModel):
        with pytest.raises(ValueError):
            cls(env).calc_xdot_from_obs(np.zeros(20))
    env.close()


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-11), ("float32", 3e-5), ("float16", 2e-2)])
@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 4099])
def test_compare_models_ragged_sizes_and_null_outputs_vs_oracle(mds, dtype, tol, n):
    """mds_compare_models through ctypes on ragged row counts (partial last wave, partial last workgroup, one row), each output
    alone and all three together, against the oracle on the same rows (rounded to the storage type first)."""
    torch = mds.torch
    env = small_env(mds, dtype)
    rng = np.random.default_rng(n)
    obs = np.zeros((n, 20))
    obs[:, 0:3] = rng.normal(size=(n, 3))
    obs[:, 7:10] = rng.uniform(-1, 1, size=(n, 3))
    obs[:, 3:7] = O.quat_from_euler_bullet(obs[:, 7:10])
    obs[:, 10:16] = rng.normal(size=(n, 6))
    obs[:, 16:20] = 14468.0 * (1 + 0.1 * rng.normal(size=(n, 4)))
    tdt = getattr(torch, dtype)
    obs_t = torch.as_tensor(obs, dtype=tdt, device=env.device)
    obs_r = obs_t.double().cpu().numpy()                       # what the kernel sees
    A, B = O.linearized_AB(noisy=True)
    ra, rb, rc = O.compare_models(obs_r, A, B, dyn_J=(1.05, 1.05, 2.05))
    lib, h = env._lib, env._h
    Ap, Bp = np.ascontiguousarray(A), np.ascontiguousarray(B)
    J = (C.c_double * 3)(1.05, 1.05, 2.05)
    PD = C.POINTER(C.c_double)
    st = C.c_void_p(torch.cuda.current_stream(env.device).cuda_stream)

    def call(want):
        outs = [torch.full((n, 12), float("nan"), dtype=tdt, device=env.device) if w else None for w in want]
        ptr = [C.c_void_p(o.data_ptr()) if o is not None else None for o in outs]
        rc_ = lib.mds_compare_models(h, n, C.c_void_p(obs_t.data_ptr()), Ap.ctypes.data_as(PD), Bp.ctypes.data_as(PD), C.c_double(0.027 * 9.8),
                                     C.c_double(0.027), J, C.c_double(9.8), ptr[0], ptr[1], ptr[2], st)
        assert rc_ == 0
        return [o.double().cpu().numpy() if o is not None else None for o in outs]

    a, b, c = call((True, True, True))
    for got, ref in ((a, ra), (b, rb), (c, rc)):
        assert np.isfinite(got).all()
        if dtype == "float16":                                 # fp16 storage: results up to ~250 rounded to 11 bits
            assert float((np.abs(got - ref) / (1 + np.abs(ref))).max()) < tol
        else:
            assert rel(got, ref) < tol
    for k in range(3):
        want = [j == k for j in range(3)]
        got = call(want)[k]
        np.testing.assert_array_equal(got, (a, b, c)[k])       # the same bits whichever outputs are requested
    assert lib.mds_compare_models(h, n, C.c_void_p(obs_t.data_ptr()), Ap.ctypes.data_as(PD), Bp.ctypes.data_as(PD), C.c_double(0.0), C.c_double(1.0),
                                  J, C.c_double(0.0), None, None, None, st) != 0                     # no output requested: MDS_EINVAL
    assert lib.mds_compare_models(h, 0, C.c_void_p(obs_t.data_ptr()), Ap.ctypes.data_as(PD), Bp.ctypes.data_as(PD), C.c_double(0.0), C.c_double(1.0),
                                  J, C.c_double(0.0), C.c_void_p(obs_t.data_ptr()), None, None, st) == 0  # empty input: nothing to do
    env.close()


def test_compare_models_on_a_logged_rollout_and_linear_roll_out(mds):
    """The script's flow (CompareModels.py:10-46, :84-98): GeometricEnv.do_control leaves [T, D, 20] observations, the env is closed,
    the models are compared on the log (one launch here), the linear model is rolled out with solve_ivp.  The arrays equal the
    oracle's on the same log; the two models' x_dot agree in the rows both copy from the state; the roll-out equals the same
    solve_ivp integration with the oracle as its right-hand side."""
    from multidronesim_amd.simulations import CompareModels as CM
    out = CM.main(["--num_drones", "2", "--duration_sec", "2", "--physics", "dyn", "--controller", "geometric", "--dtype", "float64"], roll_out=True)
    obs = out["observations"]
    assert obs.shape == (200, 2, 20)
    A, B = O.linearized_AB()
    ra, rb, rc = O.compare_models(obs, A, B, dyn_J=(1.05, 1.05, 2.05))
    np.testing.assert_allclose(out["x_dot_linear"], ra, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(out["x_dot_geometric"], rb, rtol=1e-10, atol=1e-10)
    np.testing.assert_array_equal(out["x_lin_obs"], rc)
    # rows both models copy from the state: d(rpy)/dt = w (linear) vs w (geometric "R_dot" slot), d(pos)/dt = v
    np.testing.assert_allclose(out["x_dot_linear"][..., 0:3], out["x_dot_geometric"][..., 0:3], atol=1e-12)
    np.testing.assert_allclose(out["x_dot_linear"][..., 9:12], out["x_dot_geometric"][..., 9:12], atol=1e-12)
    res = out["roll_out"]
    assert res.success and res.y.shape == (12, 200)
    # the same integration with the oracle as the right-hand side (same solver, same tolerances, same zero-order hold)
    from scipy.integrate import solve_ivp
    o0, ts = obs[:, 0], out["obs_ts"]

    def f(t, x):
        k = int(np.argmin(np.abs(ts - t)))
        if ts[k] > t:
            k -= 1
        return O.linear_calc_xdot(x, o0[k][16:], A, B)

    ref = solve_ivp(f, [0, ts[-1]], rc[0, 0], t_eval=ts)
    np.testing.assert_allclose(res.y, ref.y, rtol=1e-8, atol=1e-8)


def test_compare_models_full_size_properties(mds):
    """BASELINE config 3's full shard (65 536 envs x 8 drones = 524 288 observation rows of a real fused step, fp32), too many rows for
    the oracle: size-independent properties of the three arrays.  x_lin is a re-ordering of observation columns (bit-exact); the rows
    of A and B that copy a state component (d rpy/dt = ang_v, d pos/dt = vel) and the geometric model's (w, v) slots reproduce those
    columns bit for bit (1.0 * x + 0 * the rest is exact); the thrust rows of the two models agree to rounding where the attitude is
    level; a 512-row sample equals the oracle."""
    from multidronesim_amd.model import LinearizedModel, QuadrotorDynamics
    from multidronesim_amd.simulations.CompareModels import compare_models
    torch = mds.torch
    E, D = 65536, 8
    xyz, rpy, P = H.c2_setup(E, D, seed=3, phase="c3")
    env = mds.CtrlAviary(drone_model=mds.DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=mds.Physics.DYN,
                         pyb_freq=100, ctrl_freq=100, num_envs=E, dtype="float32")
    env.set_trajectories(P)
    env.step(torch.zeros((E, D, 4), dtype=env.dtype, device=env.device))
    t = 0.0
    for _ in range(30):
        obs = env.step_geometric(t)
        t += env.CTRL_TIMESTEP
    obs = obs.reshape(-1, 20).clone()
    lin, geo = LinearizedModel(env), QuadrotorDynamics(env.PYB_FREQ)
    geo.load_env_params(env)
    a, b, c = compare_models(lin, geo, obs)
    assert a.shape == b.shape == c.shape == (E * D, 12) and all(bool(torch.isfinite(x).all()) for x in (a, b, c))
    assert torch.equal(c, obs[:, [7, 8, 9, 13, 14, 15, 10, 11, 12, 0, 1, 2]])
    assert torch.equal(a[:, 0:3], obs[:, 13:16]) and torch.equal(a[:, 9:12], obs[:, 10:13])
    assert torch.equal(b[:, 0:3], obs[:, 13:16]) and torch.equal(b[:, 9:12], obs[:, 10:13])
    # vertical acceleration: linear (F - m g) / m against geometric R33 F / m - g; equal up to (1 - R33) F / m and rounding
    q = obs[:, 3:7].double()
    r33 = 1 - 2 * (q[:, 0] ** 2 + q[:, 1] ** 2) / (q ** 2).sum(dim=1)
    f_over_m = (a[:, 8].double() + env.G)
    assert float((a[:, 8].double() - b[:, 8].double() - (1 - r33) * f_over_m).abs().max()) < 2e-4
    idx = torch.randperm(E * D, generator=torch.Generator().manual_seed(0))[:512].to(obs.device)
    A, B = O.linearized_AB()
    ra, rb, rc = O.compare_models(obs[idx].double().cpu().numpy(), A, B, dyn_J=(1.05, 1.05, 2.05))
    for got, ref in ((a[idx], ra), (b[idx], rb), (c[idx], rc)):
        assert rel(got.double().cpu().numpy(), ref) < 3e-5
    env.close()
