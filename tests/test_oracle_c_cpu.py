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
