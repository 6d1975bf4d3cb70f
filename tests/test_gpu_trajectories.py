"""GPU parity of the general trajectory path: the reference's trajectory classes (golden vectors minted
from trajectories/*.py) through mds_traj_eval, and the fused step driven by segment tables against the
oracle loop."""
import os
import types

import numpy as np
import pytest

from oracle import np_oracle as O
from oracle import np_trajectories as NT
from tests.golden.mint_golden import trajectory_cases

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def gpu_classes():
    import multidronesim_amd.trajectories as TR
    return types.SimpleNamespace(Lemniscate=TR.Lemniscate, Circle=TR.CircleTrajectory, Line=TR.LineTrajectory, Wait=TR.WaitTrajectory,
                                 Compound=TR.CompoundTrajectory, Rotate=TR.RotateTrajectory)


def oracle_classes():
    return types.SimpleNamespace(Lemniscate=NT.Lemniscate, Circle=NT.Circle, Line=NT.Line, Wait=NT.Wait, Compound=NT.Compound, Rotate=NT.Rotate)


def test_trajectory_call_matches_reference_golden():
    d = np.load(os.path.join(G, "trajectories.npz"))
    cases = trajectory_cases(gpu_classes())
    assert list(d["names"]) == list(cases)
    for name, tr in cases.items():
        np.testing.assert_allclose(tr.get_total_time(), float(d[name + "_total"]), rtol=1e-14)
        ts, want = d[name + "_t"], d[name + "_out"]
        for k in range(0, len(ts), 3):
            pos, vel, acc, yaw, om = tr(float(ts[k]))
            got = np.hstack([pos, vel, acc, yaw, om])
            np.testing.assert_allclose(got, want[k], rtol=0, atol=1e-11, err_msg=f"{name} t={ts[k]}")


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-9), ("float32", 2e-5)])
def test_fused_step_on_segment_tables_matches_oracle(dtype, tol):
    """EnvGeometric-style loop where every drone follows a different kind of trajectory (the commented-out
    CompoundTrajectory of EnvGeometric.py:543-550 among them)."""
    from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
    gc, oc = trajectory_cases(gpu_classes()), trajectory_cases(oracle_classes())
    names = ["compound", "circle", "rotate", "line_s0", "compound_mixed", "wait"]
    D, E, steps = len(names), 3, 400
    gtr, otr = [gc[n] for n in names], [oc[n] for n in names]
    xyz = np.array([np.asarray(o(0.0)[0], dtype=np.float64) + np.array([0.05, -0.03, 0.02]) for o in otr])
    xyz = np.broadcast_to(xyz, (E, D, 3)).copy()
    env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=np.zeros((E, D, 3)), physics=Physics.DYN,
                     pyb_freq=100, ctrl_freq=100, num_envs=E, dtype=dtype)
    env.set_trajectories(gtr)
    n = E * D
    ora = O.AviaryOracle(xyz.reshape(-1, 3), np.zeros((n, 3)), pyb_freq=100, ctrl_freq=100)
    obs = ora.step(np.zeros((n, 4)))
    import torch
    env.step(torch.zeros((E, D, 4), dtype=env.dtype))
    t = 0.0
    for k in range(steps):
        des = np.array([np.hstack([np.asarray(x, dtype=np.float64) * np.ones(np.size(x)) for x in otr[i % D](t)]) for i in range(n)])
        obs = ora.step(O.geometric_compute(obs, des[:, 0:3], des[:, 3:6], des[:, 6:9], des[:, 9], des[:, 10]))
        gobs = env.step_geometric(t)
        t += env.CTRL_TIMESTEP
    g = gobs.double().cpu().numpy().reshape(n, 20)
    per_drone = np.abs(g[:, :16] - obs[:, :16]).max(axis=1).reshape(E, D).max(axis=0)
    err = dict(zip(names, per_drone))
    # the tilted (rotated) Lemniscate is the ill-conditioned one: it rides the 40-degree tilt clamp, and already
    # in float64 its error is 1000x the others' (3e-12 vs 1e-15); in fp32 that factor gives ~1e-3.
    assert max(v for k, v in err.items() if k != "rotate") < tol, err
    assert err["rotate"] < tol * 500, err
    o2 = env.rollout_geometric(t, 10)                       # the C loop also runs the general kernel
    assert np.isfinite(o2.double().cpu().numpy()).all()
    o3, _ = env.rollout_geometric_fused(t + 10 * env.CTRL_TIMESTEP, 5)   # and so does the multi-step kernel (k_rollout_traj)
    assert np.isfinite(o3.double().cpu().numpy()).all()
    env.close()


def test_whole_rollout_on_segment_tables_equals_stepwise():
    """mds_rollout_geometric_fused on general trajectories (k_rollout_traj): T steps in one launch == T mds_step_geometric calls."""
    import torch
    from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
    from multidronesim_amd import trajectories as TR
    E, D, T = 9, 3, 150
    rng = np.random.default_rng(2)
    xyz = rng.uniform(-0.5, 0.5, size=(E, D, 3)) + np.array([0, 0, 1.0])
    def trajs():
        out = []
        for e in range(E):
            for d in range(D):
                a = xyz[e, d]
                out.append(TR.CompoundTrajectory([TR.LineTrajectory(start=a, end=a + np.array([0.4, -0.2, 0.3]), speed=0.6),
                                                  TR.WaitTrajectory(duration=0.3, position=a + np.array([0.4, -0.2, 0.3]), yaw=0.2),
                                                  TR.CircleTrajectory(r=0.3, v=0.5, center=a + np.array([0.4, -0.5, 0.3]), yaw_rate=0.3)]))
        return out
    envs = []
    for _ in range(2):
        env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=np.zeros((E, D, 3)), physics=Physics.DYN,
                         pyb_freq=200, ctrl_freq=100, num_envs=E, dtype="float64")
        env.set_trajectories(trajs())
        env.step(torch.zeros((E, D, 4), dtype=env.dtype))
        envs.append(env)
    a, b = envs
    last, log = a.rollout_geometric_fused(0.0, T, log=True)
    t = 0.0
    for k in range(T):
        o = b.step_geometric(t)
        t += b.CTRL_TIMESTEP
        if k in (0, 70, T - 1):
            np.testing.assert_allclose(log[k].cpu().numpy(), o.cpu().numpy(), atol=1e-9)
    np.testing.assert_allclose(a.get_state(), b.get_state(), atol=1e-9)
    # the same loop through mds_rollout_geometric in launch form 2 (mds_set_rollout_form: the whole-rollout kernel in launches of 40 steps,
    # every step's observation into the one buffer), continuing both envs: == the step-by-step continuation
    a.set_rollout_form(2, 40)
    oa = a.rollout_geometric(T * a.CTRL_TIMESTEP, 90, obs_every_step=True)
    assert a.last_rollout_form() == 2
    t = T * b.CTRL_TIMESTEP
    for k in range(90):
        ob = b.step_geometric(t)
        t += b.CTRL_TIMESTEP
    np.testing.assert_allclose(oa.cpu().numpy(), ob.cpu().numpy(), atol=1e-9)
    np.testing.assert_allclose(a.get_state(), b.get_state(), atol=1e-9)
    a.close(); b.close()


def test_two_stream_rollout_on_segment_tables_is_bit_identical():
    """mds_set_rollout_streams(2) on the general kernel: tables of 1 and 3 pieces (two storage blocks), shared and rotated ones,
    1400 drones (5 full batches + 120): same bits as the one-stream loop."""
    import torch
    from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
    gc = trajectory_cases(gpu_classes())
    names = ["compound", "circle", "rotate", "line_s0", "compound_mixed", "wait", "circle"]
    D, E, T = len(names), 200, 45
    oc = trajectory_cases(oracle_classes())
    xyz = np.array([np.asarray(oc[n](0.0)[0], dtype=np.float64) + np.array([0.05, -0.03, 0.02]) for n in names])
    xyz = np.broadcast_to(xyz, (E, D, 3)).copy()
    out = []
    for streams in (1, 2):
        env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=np.zeros((E, D, 3)), physics=Physics.DYN,
                         pyb_freq=200, ctrl_freq=100, num_envs=E, dtype="float32")
        env.set_trajectories([gc[n] for n in names])
        env.set_rollout_streams(streams)
        env.step(torch.zeros((E, D, 4), dtype=env.dtype))
        o = env.rollout_geometric(0.0, T, obs_every_step=True).clone()
        out.append((o.cpu().numpy(), env.get_state()))
        env.close()
    assert np.isfinite(out[0][0]).all()
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])
