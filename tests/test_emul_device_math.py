"""CPU tests of the DEVICE arithmetic (multidronesim_amd/csrc/mds_math.hpp compiled with g++,
tests/emul): the same templates the HIP kernels instantiate, checked against the golden
vectors and the oracle.  This pins the restatement before any GPU time is spent; the GPU
parity tests proper are in test_gpu_parity.py."""
import os

import numpy as np
import pytest

from oracle import np_oracle as O
from tests import helpers as H
from tests.emul import emul as E

G = os.path.join(os.path.dirname(__file__), "golden")


def test_fp32_sincos_accuracy():
    x = np.linspace(-3.3, 3.3, 200001).astype(np.float32)
    s, c = E.sincos_f32(x)
    assert np.abs(s.astype(np.float64) - np.sin(x.astype(np.float64))).max() < 1.5e-7
    assert np.abs(c.astype(np.float64) - np.cos(x.astype(np.float64))).max() < 1.5e-7


@pytest.mark.parametrize("dt,tol_rpm_rel,tol_aux", [("f64", 1e-13, 1e-13), ("f32", 1e-6, 2e-6)])
def test_geometric_golden(dt, tol_rpm_rel, tol_aux):
    d = np.load(os.path.join(G, "geometric_compute.npz"))
    rpm, aux = E.Emul(dt).geometric_compute(d["obs"], d["des"])
    assert np.abs(rpm / d["rpm"] - 1).max() < tol_rpm_rel
    assert np.abs(aux[:, 0] - d["force"]).max() < tol_aux
    assert np.abs(aux[:, 1:4] - d["w_des"]).max() < tol_aux * 5
    assert np.abs(aux[:, 4:].reshape(-1, 3, 3) - d["R_des"]).max() < tol_aux


@pytest.mark.parametrize("dt,tol", [("f64", 1e-12), ("f32", 1e-5)])
def test_lemniscate_golden(dt, tol):
    g = np.load(os.path.join(G, "lemniscate.npz"))
    e = E.Emul(dt, num_envs=3)
    e.set_lemniscate(g["params"])
    for i, t in enumerate(g["ts"]):
        assert np.abs(e.lemniscate(t) - g["out"][:, i]).max() < tol


@pytest.mark.parametrize("kw", [dict(), dict(integrator=1), dict(physics=1), dict(pyb_freq=240, ctrl_freq=48)])
def test_open_loop_f64_matches_oracle(kw):
    n = 32
    xyz, rpy, ph = H.open_loop_setup(n)
    pf, cf = kw.get("pyb_freq", 240), kw.get("ctrl_freq", 240)
    ora = O.AviaryOracle(xyz, rpy, pyb_freq=pf, ctrl_freq=cf, integrator="rk4" if kw.get("integrator") else "euler",
                         physics="dyn_drag" if kw.get("physics") else "dyn")
    em = E.Emul("f64", num_envs=n, pyb_freq=pf, ctrl_freq=cf, integrator=kw.get("integrator", 0), physics=kw.get("physics", 0))
    em.set_state(np.concatenate([ora.pos, ora.quat, ora.vel, ora.rates], axis=1))
    for k in range(300):
        a = H.open_loop_rpm(k, ora.CTRL_TIMESTEP, ph)
        obs, eobs = ora.step(a), em.step(a)
    np.testing.assert_allclose(eobs, obs, atol=1e-10, rtol=1e-12)


def test_closed_loop_f32_within_1e5_over_1000_steps():
    xyz, rpy, P = H.c2_setup(16, 4)
    obs, _ = H.oracle_closed_loop(xyz, rpy, P, 1000)
    em = E.Emul("f32", num_envs=16, num_drones=4)
    n = 64
    ora0 = O.AviaryOracle(xyz.reshape(-1, 3), rpy.reshape(-1, 3), pyb_freq=100, ctrl_freq=100)
    em.set_state(np.concatenate([ora0.pos, ora0.quat, ora0.vel, ora0.rates], axis=1))
    em.set_lemniscate(P.reshape(-1, 7))
    em.step(np.zeros((n, 4)))
    t = 0.0
    for k in range(1000):
        eobs, _ = em.step_geometric(t)
        t += 0.01
    assert np.abs(eobs[:, :16] - obs[:, :16]).max() < 1e-5      # north_star tolerance, fp32 state
    assert np.abs(eobs[:, 16:] / obs[:, 16:] - 1).max() < 2e-6  # RPM echo: relative


@pytest.mark.parametrize("dt,atol_v,rtol_w", [("f64", 1e-9, 1e-9), ("f32", 2e-3, 2e-4)])
def test_device_step_accelerations_match_the_reference_tree(dt, atol_v, rtol_w):
    """The device templates (rotor_wrench in its hover-excess form, body_accel, step_euler) against the reference tree's own
    RPM -> wrench -> (v_dot, w_dot) (tests/golden/dyn_wrench_accel.npz), one 240 Hz substep."""
    d = np.load(os.path.join(G, "dyn_wrench_accel.npz"))
    n = d["rpm"].shape[0]
    em = E.Emul(dt, num_envs=n, pyb_freq=240, ctrl_freq=240)
    em.set_state(np.hstack([d["pos"], d["quat"], d["vel"], d["rates"]]))
    s0 = em.get_state()
    obs = em.step(d["rpm"])
    s1 = em.get_state()
    assert np.abs((s1[:, 7:10] - s0[:, 7:10]) * 240 - d["v_dot"]).max() <= atol_v
    assert np.abs((s1[:, 10:13] - s0[:, 10:13]) * 240 - d["w_dot"]).max() <= rtol_w * np.abs(d["w_dot"]).max()
    np.testing.assert_allclose(obs[:, 16:20], np.clip(d["rpm"], 0, float(d["max_rpm"])), rtol=1e-6 if dt == "f32" else 1e-14)


def test_compensated_fp32_open_loop_holds_1e5_over_1000_steps():
    """The compensated accumulation of the device templates (step_euler_wrench_comp, integrate_q_comp) on the CPU build, with the
    residuals the device keeps between control steps: uncontrolled 240 Hz flight, 1000 steps, against the float64 oracle.
    MDS_F32C stores the residuals of the three BODY RATES only (+32 B per drone-step); the study behind that choice, max abs state
    error at step 1000: plain fp32 1.4e-5; residuals of q only 1.1e-5 (no help: the stored quaternion is not the lever, the stored
    rate that turns it is), w only 6.2e-6, p and w 4.8e-6, all thirteen 3.3e-6 (+104 B)."""
    n = 256
    xyz, rpy, ph = H.open_loop_setup(n)

    def run(dt, integrator=0):
        ora = O.AviaryOracle(xyz, rpy, pyb_freq=240, ctrl_freq=240, integrator="rk4" if integrator else "euler")
        em = E.Emul(dt, num_envs=n, pyb_freq=240, ctrl_freq=240, integrator=integrator)
        em.set_state(np.concatenate([ora.pos, ora.quat, ora.vel, ora.rates], axis=1))
        s0 = em.get_state()
        ora.pos, ora.quat, ora.vel, ora.rates = s0[:, 0:3].copy(), s0[:, 3:7].copy(), s0[:, 7:10].copy(), s0[:, 10:13].copy()
        for k in range(1000):
            a = H.open_loop_rpm(k, ora.CTRL_TIMESTEP, ph)
            obs, eo = ora.step(a), em.step(a)
        return np.abs(eo[:, :16] - obs[:, :16]).max()

    errs = {dt: run(dt) for dt in ("f32", "f32c", "f32c:2", "f32c13")}
    assert errs["f32c"] < 8e-6 and errs["f32c"] < 0.6 * errs["f32"], errs              # the device's storage: north_star's 1e-5 holds
    assert errs["f32c:2"] > 0.7 * errs["f32"], errs                                      # quaternion residuals alone buy nothing
    assert errs["f32c13"] < errs["f32c"] < 3 * errs["f32c13"], errs                      # ten more residuals: less than 2x better
    assert run("f32c", integrator=1) < 1e-5                                              # RK4 with the same accumulation


@pytest.mark.parametrize("dt,tol_rel,tol_rot", [("f64", 1e-12, 1e-14), ("f32", 3e-5, 5e-7)])
def test_device_compare_models_row_matches_the_reference(dt, tol_rel, tol_rot):
    """compare_models_row / rpy_to_rot / rot_to_quat_scipy of csrc/mds_math.hpp (what k_compare_models, k_rpy_to_rot and
    k_geo_model_to_obs run per lane) against the reference-minted compare_models.npz.  fp32: the motor-thrust differences behind the
    torques cancel to ~1e-9 N m and are divided by a 2.4e-5 kg m^2 inertia -- errors are gated relative to 1 + |reference|."""
    d = np.load(os.path.join(G, "compare_models.npz"))
    e = E.Emul(dt)
    a, b, c = e.compare_models(d["obs"], d["A"], d["B"], 0.027 * 9.8, float(d["dyn_m"]), d["dyn_J"], float(d["dyn_g"]))
    for got, ref in ((a, d["xdot_lin"]), (b, d["xdot_geo"]), (c, d["x_lin"])):
        assert (np.abs(got - ref) / (1 + np.abs(ref))).max() < tol_rel
    np.testing.assert_allclose(E.rpy_to_rot(d["rpy"], dt), d["R_of_rpy"], rtol=0, atol=tol_rot)
    np.testing.assert_allclose(E.rot_to_quat(d["x18"][:, 3:12], dt), d["obs16"][:, 3:7], rtol=0, atol=tol_rot)
