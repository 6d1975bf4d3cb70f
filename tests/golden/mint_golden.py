#!/usr/bin/env python3
"""Mint golden input/output vectors from the reference's own NumPy code.

Run ONLY in the build container (the reference lives at /root/reference there and
does not travel to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/mint_golden.py

The reference packages cannot be imported as packages (utils/__init__ pulls in
gym_pybullet_drones, cbf pulls in cvxopt; neither is installed), so single files are
loaded BY PATH with synthetic parent packages and a placeholder ``cvxopt`` module
(only G,h are built, qp() is never called).  See SURVEY.md section 8c.

Outputs: tests/golden/*.npz -- data only (inputs, expected outputs, seeds, versions).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import scipy
from scipy.spatial.transform import Rotation

REF = os.environ.get("MDS_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def load_reference():
    mc = load("utils.model_conversions", REF + "/utils/model_conversions.py")
    u = types.ModuleType("utils")
    u.__path__ = []
    for k in dir(mc):
        if not k.startswith("_"):
            setattr(u, k, getattr(mc, k))
    u.model_conversions = mc
    sys.modules["utils"] = u
    c = types.ModuleType("control")
    c.__path__ = [REF + "/control"]
    sys.modules["control"] = c
    load("control.base_controller", REF + "/control/base_controller.py")
    geo = load("control.geometric", REF + "/control/geometric.py")
    lem = load("ref_lem", REF + "/trajectories/Lemniscate.py")
    dyn = load("ref_dyn", REF + "/model/dynamics.py")
    cv = types.ModuleType("cvxopt")
    cv.matrix = None
    cv.solvers = types.SimpleNamespace(options={})
    sys.modules["cvxopt"] = cv
    cbf = load("ref_cbf", REF + "/cbf/cbf.py")
    lin_o = load("ref_lin_omega", REF + "/model/linear_omega.py")
    lin_yo = load("ref_lin_yank_omega", REF + "/model/linear_yank_omega.py")
    return dict(mc=mc, geo=geo, lem=lem, dyn=dyn, cbf=cbf, lin_o=lin_o, lin_yo=lin_yo)


class Env:  # mock env, precedent: utils/env_builder.py:4-10
    pass


def make_env():
    env = Env()
    env.M = 0.027
    env.G = 9.8
    env.L = 0.0397
    env.KF = 3.16e-10
    env.KM = 7.94e-12
    env.J = np.diag([2.3951e-5, 2.3951e-5, 3.2347e-5])
    env.CTRL_TIMESTEP = 0.01
    env.PYB_FREQ = 100
    env.MAX_RPM = np.sqrt(2.25 * env.M * env.G / (4 * env.KF))
    env.MAX_THRUST = 4 * env.KF * env.MAX_RPM ** 2
    return env


META = dict(numpy=np.__version__, scipy=scipy.__version__)


def random_obs(rng, n, pos_c, vel_c, euler_max=0.6, w_max=1.0, pos_noise=0.3, vel_noise=0.5):
    obs = np.zeros((n, 20))
    obs[:, 0:3] = pos_c + rng.normal(size=(n, 3)) * pos_noise
    eul = rng.uniform(-euler_max, euler_max, size=(n, 3))
    obs[:, 3:7] = Rotation.from_euler("xyz", eul).as_quat()
    obs[:, 7:10] = eul
    obs[:, 10:13] = vel_c + rng.normal(size=(n, 3)) * vel_noise
    obs[:, 13:16] = rng.uniform(-w_max, w_max, size=(n, 3))
    obs[:, 16:20] = 14468.0 * (1 + 0.05 * rng.normal(size=(n, 4)))
    return obs


def mint_lemniscate(ref):
    Lem = ref["lem"].Lemniscate
    params = [dict(a=1.0, omega=1.5, center=np.array([0, 0, 0.5]), yaw_rate=0.3, phase_shift=0.0),
              dict(a=1.0, omega=1.5, center=np.array([0.2, -0.1, 0.5]), yaw_rate=0.0, phase_shift=-np.pi / 4 * 3),
              dict(a=0.7, omega=0.5, center=np.array([-1.0, 2.0, 1.5]), yaw_rate=0.11, phase_shift=2 * np.pi * 5 / 8.25)]
    ts = np.linspace(0, 30, 301)
    out = np.zeros((len(params), len(ts), 11))
    for k, p in enumerate(params):
        tr = Lem(**p)
        for i, t in enumerate(ts):
            pos, vel, acc, yaw, om = tr(t)
            out[k, i] = np.hstack([pos, vel, acc, yaw, om])
    P = np.array([[p["a"], p["omega"], *p["center"], p["yaw_rate"], p["phase_shift"]] for p in params])
    spot = Lem(center=np.array([0, 0, .5]), omega=1.5, yaw_rate=0.3)(0.37)
    np.savez_compressed(OUT + "/lemniscate.npz", params=P, ts=ts, out=out, spot=np.hstack(spot), **META)
    print("lemniscate", out.shape, "spot", np.hstack(spot))


def mint_geometric(ref):
    env = make_env()
    Lem = ref["lem"].Lemniscate
    G = ref["geo"].GeometricControl
    rng = np.random.default_rng(0)
    n_rand, n_tilt, n_clip = 320, 32, 32
    obs_list, des_list = [], []
    traj = Lem(center=np.array([0, 0, .5]), omega=1.5, yaw_rate=0.3)
    # (a) random states near the trajectory
    for k in range(n_rand):
        t = rng.uniform(0, 20)
        pos, vel, acc, yaw, om = traj(t)
        o = random_obs(rng, 1, pos, vel)[0]
        obs_list.append(o)
        des_list.append(np.hstack([pos, vel, acc, yaw, om]))
    # (b) tilt-clamp cases: large lateral position error
    for k in range(n_tilt):
        t = rng.uniform(0, 20)
        pos, vel, acc, yaw, om = traj(t)
        o = random_obs(rng, 1, pos + np.array([rng.uniform(3, 8) * rng.choice([-1, 1]), rng.uniform(-6, 6), 0.0]),
                       vel)[0]
        obs_list.append(o)
        des_list.append(np.hstack([pos, vel, acc, yaw, om]))
    # (c) min-thrust / negative-thrust clip: far ABOVE the target, moving up fast, flipped attitude
    for k in range(n_clip):
        t = rng.uniform(0, 20)
        pos, vel, acc, yaw, om = traj(t)
        o = random_obs(rng, 1, pos + np.array([0.0, 0.0, rng.uniform(4, 12)]), vel + np.array([0, 0, 3.0]),
                       euler_max=1.4, w_max=6.0)[0]
        obs_list.append(o)
        des_list.append(np.hstack([pos, vel, acc, yaw, om]))
    obs = np.array(obs_list)
    des = np.array(des_list)
    n = obs.shape[0]
    rpm = np.zeros((n, 4))
    force = np.zeros(n)
    w_des = np.zeros((n, 3))
    R_des = np.zeros((n, 3, 3))
    for i in range(n):
        ctrl = G(env)
        ctrl.set_desired_trajectory(0, des[i, 0:3].copy(), des[i, 3:6].copy(), des[i, 6:9].copy(), des[i, 9], des[i, 10])
        rpm[i] = ctrl.compute(obs[i].copy())
        f, w, Rd = ctrl.compute(obs[i].copy(), return_omegas=True)
        force[i], w_des[i], R_des[i] = f, w, Rd
    # spot value of SURVEY.md 8c (G1)
    rng1 = np.random.default_rng(1)
    pos, vel, acc, yaw, om = traj(0.37)
    so = np.zeros(20)
    so[:3] = pos + rng1.normal(size=3) * 0.05
    so[3:7] = Rotation.from_euler('xyz', [0.1, -0.07, 0.2]).as_quat()
    so[10:13] = vel + rng1.normal(size=3) * 0.05
    so[13:16] = rng1.normal(size=3) * 0.2
    ctrl = G(env)
    ctrl.set_desired_trajectory(0, pos, vel, acc, yaw, om)
    spot_rpm = ctrl.compute(so.copy())
    _, spot_w, _ = ctrl.compute(so.copy(), return_omegas=True)
    np.savez_compressed(OUT + "/geometric_compute.npz", obs=obs, des=des, rpm=rpm, force=force, w_des=w_des, R_des=R_des,
             spot_obs=so, spot_des=np.hstack([pos, vel, acc, yaw, om]), spot_rpm=spot_rpm, spot_w_des=spot_w,
             n_rand=n_rand, n_tilt=n_tilt, n_clip=n_clip, **META)
    lo = 9440.3
    print("geometric", n, "spot", spot_rpm, spot_w, "minclip rows:", int((np.abs(rpm - lo) < 1e-6).any(axis=1).sum()),
          "maxclip rows:", int((rpm > 43000).any(axis=1).sum()))


def mint_mixer(ref):
    env = make_env()
    mc = ref["mc"]
    rng = np.random.default_rng(2)
    n = 128
    u = np.zeros((n, 4))
    u[:, 0] = rng.uniform(-0.1, 0.8, size=n)
    u[:, 1:3] = rng.normal(size=(n, 2)) * 2e-3
    u[:, 3] = rng.normal(size=n) * 5e-4
    rpm = np.array([mc.input_to_action(env, ui.copy()) for ui in u])
    act = rng.uniform(-2000, 26000, size=(n, 4))
    u_back = np.array([mc.action_to_input(env, a.copy()) for a in act])
    u_back_nocap = np.array([mc.action_to_input(env, a.copy(), cap_rpm=False) for a in act])
    obs = random_obs(rng, 64, np.zeros(3), np.zeros(3))
    lin9 = np.array([mc.obs_to_lin_model(o, dim=9) for o in obs])
    lin10 = np.array([mc.obs_to_lin_model(o, dim=10, env=env) for o in obs])
    lin12 = np.array([mc.obs_to_lin_model(o, dim=12) for o in obs])
    geo18 = np.array([mc.obs_to_geo_model(o) for o in obs])
    np.savez_compressed(OUT + "/mixer.npz", u=u, rpm=rpm, act=act, u_back=u_back, u_back_nocap=u_back_nocap, obs=obs, lin9=lin9,
             lin10=lin10, lin12=lin12, geo18=geo18, **META)
    print("mixer", rpm.shape)


def mint_dynamics(ref):
    env = make_env()
    Q = ref["dyn"].QuadrotorDynamics
    rng = np.random.default_rng(3)
    n = 128
    state = np.zeros((n, 18))
    state[:, 0:3] = rng.normal(size=(n, 3))
    state[:, 3:12] = Rotation.from_euler("xyz", rng.uniform(-1, 1, size=(n, 3))).as_matrix().reshape(n, 9)
    state[:, 12:15] = rng.normal(size=(n, 3))
    state[:, 15:18] = rng.normal(size=(n, 3)) * 2
    u = np.zeros((n, 4))
    u[:, 0] = rng.uniform(0, 100, size=n)
    u[:, 1:] = rng.normal(size=(n, 3)) * 0.5
    q = Q(sim_freq=100)
    out_hb = np.array([q.dynamics(0.0, s, ui) for s, ui in zip(state, u)])
    q2 = Q(sim_freq=100)
    q2.load_env_params(env)  # stale-J quirk: m,g from env, J stays Hummingbird
    u2 = u.copy()
    u2[:, 0] = rng.uniform(0, 0.6, size=n)
    u2[:, 1:] *= 1e-3
    out_env = np.array([q2.dynamics(0.0, s, ui) for s, ui in zip(state, u2)])
    step_raises = False
    try:
        q.step(u[0])
    except ValueError:
        step_raises = True
    np.savez_compressed(OUT + "/dynamics_deriv.npz", state=state, u=u, out_hb=out_hb, u_env=u2, out_env=out_env,
             env_m=q2.m, env_g=q2.g, env_J=np.diag(q2.J), step_raises=step_raises, **META)
    print("dynamics", out_hb.shape, "step() raises:", step_raises)


def mint_dyn_wrench_accel(ref):
    """The part of [UPSTREAM] _dynamics the reference tree itself restates: RPM -> (thrust, torques) is action_to_input
    (utils/model_conversions.py:69-83, CF2P "+" frame, RPM clipped at MAX_RPM like the simulator) and (v_dot, w_dot) is
    QuadrotorDynamics.dynamics (model/dynamics.py:83-106) once its m, g, J are the env's (CF2P, g = 9.8).  Random attitudes,
    velocities, body rates and RPM (near hover, wide, and beyond MAX_RPM): inputs + the reference's outputs."""
    env = make_env()
    mc = ref["mc"]
    Q = ref["dyn"].QuadrotorDynamics
    q = Q(sim_freq=240)
    q.m, q.g = env.M, env.G
    q.Jxx, q.Jyy, q.Jzz = env.J[0, 0], env.J[1, 1], env.J[2, 2]
    q.J = np.diag([q.Jxx, q.Jyy, q.Jzz])
    q.J_inv = np.linalg.inv(q.J)
    rng = np.random.default_rng(11)
    n = 384
    hover = np.sqrt(env.M * env.G / (4 * env.KF))
    rpm = np.empty((n, 4))
    rpm[:128] = hover * (1 + 0.03 * rng.normal(size=(128, 4)))
    rpm[128:256] = rng.uniform(0, env.MAX_RPM, size=(128, 4))
    rpm[256:] = rng.uniform(-0.1 * env.MAX_RPM, 1.3 * env.MAX_RPM, size=(128, 4))      # clipped on both sides
    quat = Rotation.from_euler("xyz", rng.uniform(-1.2, 1.2, size=(n, 3))).as_quat()
    R = Rotation.from_quat(quat).as_matrix()
    pos = rng.normal(size=(n, 3))
    vel = rng.normal(size=(n, 3)) * 2
    rates = rng.normal(size=(n, 3)) * 3
    u = np.array([mc.action_to_input(env, a.copy()) for a in rpm])
    state = np.hstack([pos, R.reshape(n, 9), vel, rates])
    ydot = np.array([q.dynamics(0.0, s, ui) for s, ui in zip(state, u)])
    np.savez_compressed(OUT + "/dyn_wrench_accel.npz", rpm=rpm, quat=quat, pos=pos, vel=vel, rates=rates, u=u, v_dot=ydot[:, 6:9],
                        w_dot=ydot[:, 9:12], max_rpm=env.MAX_RPM, m=env.M, g=env.G, J=np.diag(env.J), **META)
    print("dyn_wrench_accel", u.shape, "clipped rows:", int(((rpm < 0) | (rpm > env.MAX_RPM)).any(axis=1).sum()))


def mint_attitude_flow(ref):
    """The attitude kinematics the reference tree states: body-frame angular velocity, R_dot = R hat(w)
    (model/dynamics.py:62-66 hat_map, :102 R_dot).  For a constant body rate its flow over dt is R(dt) = R expm(hat(w) dt): the
    reference's own hat_map through scipy's expm, for random attitudes / rates / step sizes (incl. |w| = 0 and |w| dt > pi).
    [UPSTREAM] _integrateQ has to produce that rotation."""
    from scipy.linalg import expm
    q = ref["dyn"].QuadrotorDynamics(sim_freq=240)
    rng = np.random.default_rng(31)
    n = 256
    quat = Rotation.from_euler("xyz", rng.uniform(-3.1, 3.1, size=(n, 3))).as_quat()
    w = rng.normal(size=(n, 3)) * np.concatenate([np.full(64, 0.3), np.full(64, 3.0), np.full(64, 30.0), np.full(64, 300.0)])[:, None]
    w[::37] = 0.0
    dt = np.tile(np.array([1 / 240, 1 / 100, 1 / 48, 0.05]), n // 4)
    R = Rotation.from_quat(quat).as_matrix()
    R_next = np.array([np.matmul(Ri, expm(q.hat_map(wi) * di)) for Ri, wi, di in zip(R, w, dt)])
    R_dot = np.array([np.matmul(Ri, q.hat_map(wi)) for Ri, wi in zip(R, w)])           # dynamics.py:102
    np.savez_compressed(OUT + "/attitude_flow.npz", quat=quat, w=w, dt=dt, R_next=R_next, R_dot=R_dot, **META)
    print("attitude_flow", R_next.shape, "max |w| dt", float(np.max(np.linalg.norm(w, axis=1) * dt)))


def mint_euler_convention(ref):
    """The rpy convention of the observation (obs[7:10]) pinned on the reference tree.

    Upstream fills obs[7:10] with pybullet's getEulerFromQuaternion (not in the tree); every linear-model consumer in the tree reads it
    through obs_to_lin_model (utils/model_conversions.py:36-40) and turns it back into a rotation with the tree's own convention:
    rpy_to_rot (utils/model_conversions.py:4-19, R = Rz Ry Rx) and scipy 'xyz' (control/lqr/lqr_omega_controller.py:97-101), while the
    quaternion of the same observation goes through Rotation.from_quat (obs_to_geo_model, :108-113).  For the two readings of one
    observation to agree, rpy_to_rot(obs[7:10]) must equal the rotation of obs[3:7].  Stored: random quaternions (general attitudes,
    pitch within 1e-6 .. 1e-2 of +-pi/2, and exact +-pi/2), the rotation obs_to_geo_model gives them, and -- for the rpy the oracle's
    euler_from_quat_bullet assigns at minting time -- the reference's rpy_to_rot(rpy) and scipy's from_euler('xyz', rpy) as
    lqr_omega_controller uses it."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import np_oracle as O
    mc = ref["mc"]
    rng = np.random.default_rng(77)
    n_gen, n_near = 192, 96
    eul = np.concatenate([
        np.stack([rng.uniform(-np.pi, np.pi, n_gen), rng.uniform(-1.5, 1.5, n_gen), rng.uniform(-np.pi, np.pi, n_gen)], axis=1),
        # near the gimbal lock: |pitch| = pi/2 - eps, eps from 1e-2 down to 1e-6 (both sides of pybullet's 0.99999 threshold, which
        # sits at eps = 4.47e-3), and exactly pi/2
        np.stack([rng.uniform(-np.pi, np.pi, n_near), np.tile([1.0, -1.0], n_near // 2) * (np.pi / 2 - np.tile(np.repeat(
            [1e-2, 6e-3, 4.5e-3, 4.4e-3, 1e-3, 1e-4, 1e-5, 1e-6], 2), n_near // 16)), rng.uniform(-np.pi, np.pi, n_near)], axis=1),
        np.array([[0.3, np.pi / 2, -0.8], [-1.1, -np.pi / 2, 2.0], [0.0, np.pi / 2, 0.0], [0.0, 0.0, 0.0]]),
    ])
    # quaternions straight from the Euler angles (extrinsic xyz = R = Rz Ry Rx), float64; random sign flips (q and -q are one attitude)
    q = Rotation.from_euler("xyz", eul).as_quat()
    q *= np.where(rng.uniform(size=(len(q), 1)) < 0.5, -1.0, 1.0)
    obs = np.zeros((len(q), 20))
    obs[:, 3:7] = q
    R_quat = np.array([mc.obs_to_geo_model(o)[3:12].reshape(3, 3) for o in obs])          # the tree's reading of the quaternion
    rpy = O.euler_from_quat_bullet(q)                                                      # what the build puts into obs[7:10]
    R_rpy = np.array([mc.rpy_to_rot(r) for r in rpy])                                      # the tree's reading of that rpy
    R_rpy_scipy = Rotation.from_euler("xyz", rpy).as_matrix()                              # lqr_omega_controller.py:98
    sarg = -2.0 * (q[:, 0] * q[:, 2] - q[:, 3] * q[:, 1]) / (q * q).sum(1)
    np.savez(os.path.join(OUT, "euler_convention.npz"), quat=q, euler_in=eul, rpy=rpy, R_quat=R_quat, R_rpy=R_rpy, R_rpy_scipy=R_rpy_scipy,
             gimbal=np.abs(sarg) >= 0.99999, **{"meta_" + k: v for k, v in META.items()})
    g = np.abs(sarg) >= 0.99999
    print("euler_convention: %d attitudes, %d in the gimbal branches; max |rpy_to_rot(rpy) - R(q)| outside them %.2e, inside %.2e" % (
        len(q), g.sum(), np.abs(R_rpy - R_quat)[~g].max(), np.abs(R_rpy - R_quat)[g].max()))


def mint_closed_loop_reference_in_the_loop(ref):
    """The do_control loop of simulations/EnvGeometric.py:431-473 with the REFERENCE's own objects in it: per drone a
    trajectories/Lemniscate.py object sampled at t, a control/geometric.py GeometricControl.compute(obs) (through the reference's
    obs_to_geo_model and input_to_action), t += CTRL_TIMESTEP.  Only env.step is not the reference's (it is [UPSTREAM] PyBullet
    there): the explicit DYN step of oracle/np_oracle.py stands in, and the fixture says so.  8 drones (the C3 phase rule, per-drone
    centres, one drone with a yawing trajectory), 100 Hz, 1000 control steps; observations every 50 steps."""
    import os as _os
    sys.path.insert(0, _os.path.dirname(_os.path.dirname(OUT)))
    from oracle import np_oracle as O
    env = make_env()
    Lem = ref["lem"].Lemniscate
    G = ref["geo"].GeometricControl
    D, steps = 8, 1000
    rng = np.random.default_rng(21)
    cen = np.zeros((D, 3))
    cen[:, :2] = rng.uniform(-5, 5, size=(D, 2))
    cen[:, 2] = 0.5
    ang = 2 * np.pi * np.arange(D) / D
    xyz = cen + np.stack([np.sin(ang), np.cos(ang), np.zeros(D)], axis=1)
    P = np.zeros((D, 7))
    P[:, 0], P[:, 1], P[:, 2:5] = 1.0, 1.5, cen
    P[:, 6] = 2 * np.pi * np.arange(D) / (D + 0.25)
    P[3, 5] = 0.02                                  # one slowly yawing drone (yaw reaches 0.6 rad; the reference controller, with its no-op transpose, leaves its trajectory for good once |yaw| passes ~2 rad: yaw_rate 0.2 diverges after 3.5 s)
    P[5, 0], P[5, 1] = 0.7, 1.1
    trajs = [Lem(a=P[j, 0], omega=P[j, 1], center=P[j, 2:5].copy(), yaw_rate=P[j, 5], phase_shift=P[j, 6]) for j in range(D)]
    ctrl = [G(env) for _ in range(D)]
    ora = O.AviaryOracle(xyz, np.zeros((D, 3)), O.CF2P, 100, 100)
    obs = ora.step(np.zeros((D, 4)))                # EnvGeometric.py:431
    t, log, acts = 0.0, [obs.copy()], []
    for i in range(steps):
        action = np.zeros((D, 4))
        for j in range(D):
            pos, vel, acc, yaw, omega = trajs[j](t)
            ctrl[j].set_desired_trajectory(j, pos, vel, acc, yaw, omega)
            action[j] = ctrl[j].compute(obs[j].copy())
        obs = ora.step(action)
        t += 0.01
        if (i + 1) % 50 == 0:
            log.append(obs.copy())
            acts.append(action.copy())
    np.savez_compressed(OUT + "/closed_loop_ref_in_loop.npz", xyz=xyz, params=P, obs_log=np.array(log), action_log=np.array(acts), steps=steps,
                        every=50, physics="oracle DYN step (np_oracle.AviaryOracle); trajectories and controller: reference objects", **META)
    print("closed loop (reference in the loop)", np.array(log).shape, "final |pos - centre| max", np.abs(obs[:, :3] - cen).max())


def mint_closed_loop_lqr_reference_in_the_loop(ref):
    """The same loop with the controller simulations/EnvGeometric.py runs by default: per drone a reference LQRController
    (control/lqr/lqr_controller.py on model/linearized.py, gain from its own solve_continuous_are) fed by a reference Lemniscate,
    wind 2.5e-4 N along x on every drone from the first control step on (EnvGeometric.py:34,463-467).  Only env.step is the oracle's
    DYN step.  4 drones, 100 Hz, 600 control steps; observations every 50 steps."""
    import contextlib
    import io
    import os as _os
    sys.path.insert(0, _os.path.dirname(_os.path.dirname(OUT)))
    from oracle import np_oracle as O
    lin = load("model.linearized", REF + "/model/linearized.py")
    mod = load("control.lqr.lqr_controller", REF + "/control/lqr/lqr_controller.py")
    env = make_env()
    Lem = ref["lem"].Lemniscate
    D, steps = 4, 600
    xyz = np.array([[np.sin(2 * np.pi * j / D), np.cos(2 * np.pi * j / D), 0.5] for j in range(D)])
    P = np.array([[1.0, 1.5, 0.0, 0.0, 0.5, 0.0, (-np.pi / 4) * (j - 1)] for j in range(D)])         # EnvGeometric.py:540
    trajs = [Lem(a=P[j, 0], omega=P[j, 1], center=P[j, 2:5].copy(), yaw_rate=P[j, 5], phase_shift=P[j, 6]) for j in range(D)]
    with contextlib.redirect_stdout(io.StringIO()):
        ctrl = [mod.LQRController(env, lin.LinearizedModel(env)) for _ in range(D)]
    ora = O.AviaryOracle(xyz, np.zeros((D, 3)), O.CF2P, 100, 100)
    obs = ora.step(np.zeros((D, 4)))
    ora.wind = np.array([2.5e-4, 0.0, 0.0])
    t, log, acts = 0.0, [obs.copy()], []
    for i in range(steps):
        action = np.zeros((D, 4))
        for j in range(D):
            pos, vel, acc, yaw, omega = trajs[j](t)
            ctrl[j].set_desired_trajectory(j, pos, vel, acc, yaw, omega)
            action[j], _ = ctrl[j].compute(obs[j].copy())
        obs = ora.step(action)
        t += 0.01
        if (i + 1) % 50 == 0:
            log.append(obs.copy())
            acts.append(action.copy())
    np.savez_compressed(OUT + "/closed_loop_lqr_ref_in_loop.npz", xyz=xyz, params=P, K=ctrl[0].K, wind=ora.wind, obs_log=np.array(log),
                        action_log=np.array(acts), steps=steps, every=50,
                        physics="oracle DYN step (np_oracle.AviaryOracle); trajectories and controller: reference objects", **META)
    print("closed loop lqr (reference in the loop)", np.array(log).shape, "final |pos - centre| max", np.abs(obs[:, :3] - P[:, 2:5]).max())


def mint_cbf(ref):
    env = make_env()
    cbf = ref["cbf"]
    rng = np.random.default_rng(4)
    for order, Model, poles, xdim, sr, zs in (
            (2, ref["lin_o"].LinearizedOmegaModel, np.array([-2.2, -2.4]), 9, 0.1, 1.0),
            (3, ref["lin_yo"].LinearizedYankOmegaModel, np.array([-3.0, -3.6, -5.6]), 10, 0.125, 2.0)):
        cases = {}
        idx = 0
        raises_when_nobs_gt_n = False
        for N in (2, 3, 7, 16):
            for N_obs in (0, 1, 4):
                models = [Model(env) for _ in range(N)]
                d = cbf.DroneCBF(env, models, safety_radius=sr, zscale=zs, order=order, cbf_poles=poles)
                x = np.zeros((N, xdim))
                x[:, 0:3] = rng.uniform(-0.4, 0.4, size=(N, 3))
                if order == 3:
                    x[:, 3] = env.M * env.G * (1 + 0.2 * rng.normal(size=N))
                x[:, -6:-3] = rng.normal(size=(N, 3)) * 0.7
                x[:, -3:] = rng.uniform(-0.8, 0.8, size=(N, 3)) + np.array([0, 0, 0.5])
                xdes = np.zeros((N, xdim))
                xdes[:, 2] = rng.uniform(-1, 1, size=N)
                if order == 3:
                    xdes[:, 3] = env.M * env.G
                xdes[:, -6:-3] = rng.normal(size=(N, 3)) * 0.5
                xdes[:, -3:] = x[:, -3:] + rng.normal(size=(N, 3)) * 0.1
                d.set_xdes(xdes)
                if N_obs:
                    x_obs = np.zeros((N_obs, order, 3))
                    x_obs[:, 0, :] = rng.uniform(-0.6, 0.6, size=(N_obs, 3)) + np.array([0, 0, 0.5])
                    obs_r = list(rng.uniform(0.05, 0.2, size=N_obs))
                    try:
                        G, h = d._build_ineq_const(x.copy(), False, list(x_obs), obs_r)
                    except ValueError:
                        # reference quirk: getABij(i, j) indexes the AGENT block j for obstacle j
                        # (cbf/cbf.py:388,180-192) -> N_obs > N raises.  Recorded, case skipped.
                        assert N_obs > N
                        raises_when_nobs_gt_n = True
                        continue
                else:
                    x_obs = np.zeros((0, order, 3))
                    obs_r = []
                    G, h = d._build_ineq_const(x.copy(), False, None, None)
                cases[f"c{idx}_x"] = x
                cases[f"c{idx}_xdes"] = xdes
                cases[f"c{idx}_xobs"] = x_obs
                cases[f"c{idx}_obsr"] = np.array(obs_r)
                cases[f"c{idx}_G"] = G
                cases[f"c{idx}_h"] = h
                idx += 1
                Kcbf = np.asarray(d.Kcbf).reshape(-1)
                umax = np.asarray(d.umax)
        np.savez_compressed(OUT + f"/cbf_rows_o{order}.npz", n_cases=idx, order=order, poles=poles, Kcbf=Kcbf, umax=umax,
                 safety_radius=sr, zscale=zs, Fmin=-env.M * env.G, Fmax=env.MAX_THRUST,
                 raises_when_nobs_gt_n=raises_when_nobs_gt_n, **cases, **META)
        print("cbf order", order, "cases", idx, "Kcbf", Kcbf, "umax", umax, "last G", G.shape)


def mint_thrust_omega():
    """control/low_level/thrust_omega_ctrl.py subclasses [UPSTREAM] gym_pybullet_drones BaseControl,
    which is not installed: the fixture is minted with a STUB base class that supplies only what the
    file reads (DRONE_MODEL, KF, reset() -> control_counter).  Marked base-class stubbed."""
    gpd = types.ModuleType("gym_pybullet_drones")
    gpd.__path__ = []
    ctl = types.ModuleType("gym_pybullet_drones.control")
    ctl.__path__ = []
    bc = types.ModuleType("gym_pybullet_drones.control.BaseControl")
    ut = types.ModuleType("gym_pybullet_drones.utils")
    ut.__path__ = []
    en = types.ModuleType("gym_pybullet_drones.utils.enums")
    from enum import Enum

    class DroneModel(Enum):
        CF2X = "cf2x"
        CF2P = "cf2p"
        RACE = "racer"

    class BaseControl:   # stub of [UPSTREAM] BaseControl: urdf constants + counter only
        def __init__(self, drone_model, g=9.8):
            self.DRONE_MODEL = drone_model
            self.GRAVITY = g * 0.027
            self.KF = 3.16e-10
            self.KM = 7.94e-12
            self.reset()

        def reset(self):
            self.control_counter = 0

    en.DroneModel = DroneModel
    bc.BaseControl = BaseControl
    for name, mod in (("gym_pybullet_drones", gpd), ("gym_pybullet_drones.control", ctl),
                      ("gym_pybullet_drones.control.BaseControl", bc), ("gym_pybullet_drones.utils", ut),
                      ("gym_pybullet_drones.utils.enums", en)):
        sys.modules[name] = mod
    to = load("ref_thrust_omega", REF + "/control/low_level/thrust_omega_ctrl.py")
    rng = np.random.default_rng(5)
    out = {}
    for model in ("cf2p", "cf2x"):
        env = make_env()
        env.DRONE_MODEL = DroneModel(model)
        n, T = 24, 40
        u = np.zeros((T, n, 4))
        u[..., 0] = 0.2646 * (1 + 0.3 * rng.normal(size=(T, n)))
        u[:, :4, 0] = rng.uniform(-0.05, 0.02, size=(T, 4))              # negative / tiny thrust -> MIN_PWM clip
        u[..., 1:] = rng.normal(size=(T, n, 3)) * 1.5
        u[:, 4:8, 1:] *= 20.0                                              # torque clip +-3200 and PWM clips
        cur = rng.normal(size=(T, n, 3)) * 0.8
        rpm = np.zeros((T, n, 4))
        ctrls = [to.ThrustOmegaController(env) for _ in range(n)]
        for t in range(T):
            for i in range(n):
                rpm[t, i] = ctrls[i].computeControlFromInput(u[t, i].copy(), 0.01, cur[t, i].copy())
        out[f"{model}_u"], out[f"{model}_cur"], out[f"{model}_rpm"] = u, cur, rpm
        out[f"{model}_integral"] = np.array([c.integral_omega_e for c in ctrls])
    np.savez_compressed(OUT + "/thrust_omega.npz", dt=0.01, base_class="stubbed", **out, **META)
    print("thrust_omega", rpm.shape, "clips: min", int((rpm <= 0.2685 * 20000 + 4070.3 + 1e-9).sum()),
          "max", int((rpm >= 0.2685 * 65535 + 4070.3 - 1e-9).sum()))


def mint_lqr_omega(ref):
    """control/lqr/lqr_omega_controller.py: gain matrix and compute(obs, skip_low_level=True).  Its import of
    control.low_level.thrust_omega_ctrl needs the stubbed [UPSTREAM] BaseControl registered by mint_thrust_omega()."""
    m = types.ModuleType("model")
    m.__path__ = [REF + "/model"]
    sys.modules["model"] = m
    sys.modules["model.linear_omega"] = ref["lin_o"]
    sys.modules["utils.model_conversions"] = ref["mc"]
    ll = types.ModuleType("control.low_level")
    ll.__path__ = [REF + "/control/low_level"]
    sys.modules["control.low_level"] = ll
    load("control.low_level.thrust_omega_ctrl", REF + "/control/low_level/thrust_omega_ctrl.py")
    lq = types.ModuleType("control.lqr")
    lq.__path__ = [REF + "/control/lqr"]
    sys.modules["control.lqr"] = lq
    mod = load("control.lqr.lqr_omega_controller", REF + "/control/lqr/lqr_omega_controller.py")
    env = make_env()
    from enum import Enum
    env.DRONE_MODEL = sys.modules["gym_pybullet_drones.utils.enums"].DroneModel("cf2p")
    ctrl = mod.LQROmegaController(env, ref["lin_o"].LinearizedOmegaModel(env), None)
    rng = np.random.default_rng(6)
    n = 192
    obs = random_obs(rng, n, np.array([0.3, -0.2, 0.8]), np.zeros(3), euler_max=0.5, pos_noise=0.4, vel_noise=0.4)
    obs[:, 9] = rng.uniform(-3.1, 3.1, size=n)                      # yaw over the full circle (wrap of yaw - yaw_des)
    pos_d = np.array([0.3, -0.2, 0.8]) + rng.normal(size=(n, 3)) * 0.2
    vel_d = rng.normal(size=(n, 3)) * 0.3
    yaw_d = rng.uniform(-3.1, 3.1, size=n)
    pos_d[:16] += np.array([0, 0, 3.0])                              # thrust cap high
    pos_d[16:32] -= np.array([0, 0, 3.0])                            # thrust cap low
    u = np.zeros((n, 4))
    for i in range(n):
        ctrl.set_desired_trajectory(0, pos_d[i], vel_d[i], np.zeros(3), yaw_d[i], 0.0)
        _, u[i] = ctrl.compute(obs[i].copy(), skip_low_level=True)
    np.savez_compressed(OUT + "/lqr_omega.npz", K=ctrl.K, obs=obs, pos_d=pos_d, vel_d=vel_d, yaw_d=yaw_d, u=u, **META)
    print("lqr_omega K", ctrl.K.shape, "u0 range", u[:, 0].min(), u[:, 0].max())


def mint_lqr_yank_omega(ref):
    """control/lqr/lqr_YO_controller.py (gain, compute(obs, skip_low_level=True), compute_low_level) over
    control/low_level/yank_omega_ctrl.py.  Needs the modules mint_lqr_omega() registered; the low level rides on the
    base-class-stubbed ThrustOmegaController, so the rpm part is marked stubbed like thrust_omega.npz."""
    sys.modules["model.linear_yank_omega"] = ref["lin_yo"]
    load("control.low_level.yank_omega_ctrl", REF + "/control/low_level/yank_omega_ctrl.py")
    mod = load("control.lqr.lqr_YO_controller", REF + "/control/lqr/lqr_YO_controller.py")
    yo = sys.modules["control.low_level.yank_omega_ctrl"]
    env = make_env()
    env.DRONE_MODEL = sys.modules["gym_pybullet_drones.utils.enums"].DroneModel("cf2p")
    rng = np.random.default_rng(7)
    n, T = 96, 12
    ctrls = [mod.LQRYankOmegaController(env, ref["lin_yo"].LinearizedYankOmegaModel(env), yo.YankOmegaController(env)) for _ in range(n)]
    obs = np.zeros((T, n, 20))
    pos_d = np.array([0.3, -0.2, 0.8]) + rng.normal(size=(T, n, 3)) * 0.2
    vel_d = rng.normal(size=(T, n, 3)) * 0.3
    yaw_d = rng.uniform(-3.1, 3.1, size=(T, n))
    u = np.zeros((T, n, 4))
    rpm = np.zeros((T, n, 4))
    for t in range(T):
        obs[t] = random_obs(rng, n, np.array([0.3, -0.2, 0.8]), np.zeros(3), euler_max=0.5, pos_noise=0.4, vel_noise=0.4)
        obs[t, :, 9] = rng.uniform(-3.1, 3.1, size=n)
        obs[t, :8, 16:20] = 0.0                                           # first obs of a run: RPM 0 -> cur_thrust 0
        obs[t, 8:16, 16:20] *= 1.45                                       # near MAX_RPM
        for i in range(n):
            ctrls[i].set_desired_trajectory(0, pos_d[t, i], vel_d[t, i], np.zeros(3), yaw_d[t, i], 0.0)
            _, u[t, i] = ctrls[i].compute(obs[t, i].copy(), skip_low_level=True)
            rpm[t, i] = ctrls[i].compute_low_level(u[t, i].copy(), obs[t, i].copy())
    np.savez_compressed(OUT + "/lqr_yank_omega.npz", K=ctrls[0].K, obs=obs, pos_d=pos_d, vel_d=vel_d, yaw_d=yaw_d, u=u, rpm=rpm,
                        dt=env.CTRL_TIMESTEP, low_level_base_class="stubbed", **META)
    print("lqr_yank_omega K", ctrls[0].K.shape, "yank range", u[..., 0].min(), u[..., 0].max())


def mint_lqr12(ref):
    """control/lqr/lqr_controller.py on model/linearized.py: the 12-state LQR that simulations/EnvGeometric.py runs by default
    (controllers[0] = 'lqr', :32, :425-427), with the true and the deliberately wrong model (use_noisy_model, :551)."""
    lin = load("model.linearized", REF + "/model/linearized.py")
    mod = load("control.lqr.lqr_controller", REF + "/control/lqr/lqr_controller.py")
    env = make_env()
    rng = np.random.default_rng(8)
    n = 160
    obs = random_obs(rng, n, np.array([0.3, -0.2, 0.8]), np.zeros(3), euler_max=0.4, pos_noise=0.3, vel_noise=0.3)
    obs[:, 9] = rng.uniform(-3.1, 3.1, size=n)
    pos_d = np.array([0.3, -0.2, 0.8]) + rng.normal(size=(n, 3)) * 0.1
    vel_d = rng.normal(size=(n, 3)) * 0.3
    yaw_d = obs[:, 9] + rng.normal(size=n) * 0.3                      # the 1000 rad^-2 attitude weights saturate the motors otherwise
    yaw_d[:24] = rng.uniform(-3.1, 3.1, size=24)                      # ... which these cases do on purpose (min-thrust clip of the mixer)
    om_d = rng.normal(size=n) * 0.5
    out = {}
    import io, contextlib
    for noisy in (False, True):
        with contextlib.redirect_stdout(io.StringIO()):               # the constructor prints its weights
            ctrl = mod.LQRController(env, lin.LinearizedModel(env), use_noisy_model=noisy)
        u = np.zeros((n, 4))
        act = np.zeros((n, 4))
        for i in range(n):
            ctrl.set_desired_trajectory(0, pos_d[i], vel_d[i], np.zeros(3), yaw_d[i], om_d[i])
            act[i], u[i] = ctrl.compute(obs[i].copy())
        tag = "noisy" if noisy else "true"
        out["K_" + tag], out["u_" + tag], out["act_" + tag] = ctrl.K, u, act
    np.savez_compressed(OUT + "/lqr12.npz", obs=obs, pos_d=pos_d, vel_d=vel_d, yaw_d=yaw_d, om_d=om_d, **out, **META)
    lo = 9440.3
    print("lqr12 K", out["K_true"].shape, "min-thrust clips", int((np.abs(out["act_true"] - lo) < 1e-6).sum()), "of", act.size)


def mint_compare_models(ref):
    """The call site of QuadrotorDynamics.dynamics -- simulations/CompareModels.py:46-56, the loop body over logged observation
    rows, through the reference's own objects: LinearizedModel(env).calc_xdot_from_obs(obs) (model/linearized.py:83-104),
    geo_x_dot_to_linear(QuadrotorDynamics[load_env_params(env)].dynamics(None, obs_to_geo_model(obs), action_to_input(env,
    obs[16:]))) and obs_to_lin_model(obs); calc_xdot(x, action) on states that are NOT the observation's (the right-hand side
    roll_out_linear_system integrates, :84-92), with the true and the deliberately wrong (Ahat, Bhat) pair; rpy_to_rot and
    geo_model_to_obs (utils/model_conversions.py:4-19, :116-122).  Rows include RPMs above MAX_RPM (clipped by action_to_input),
    at zero and negative (clipped to 0)."""
    lin = load("model.linearized", REF + "/model/linearized.py")
    mc = ref["mc"]
    env = make_env()
    rng = np.random.default_rng(21)
    n = 320
    obs = random_obs(rng, n, np.array([0.5, -0.4, 1.2]), np.array([0.2, -0.1, 0.05]), euler_max=0.9, w_max=3.0, pos_noise=1.0, vel_noise=1.0)
    obs[:, 9] = rng.uniform(-3.1, 3.1, size=n)
    obs[:, 3:7] = Rotation.from_euler("xyz", obs[:, 7:10]).as_quat()
    obs[::7, 3:7] *= -1.0                                              # the same attitudes with the other quaternion sign
    obs[:48, 16:20] = rng.uniform(-2000.0, 26000.0, size=(48, 4))      # below 0 and above MAX_RPM (21 702): clipped
    obs[48:56, 16:20] = 0.0
    model = lin.LinearizedModel(env)
    Q = ref["dyn"].QuadrotorDynamics
    gd = Q(env.PYB_FREQ)
    gd.load_env_params(env)                                            # m, g from the env; J stays the Hummingbird one (:18)
    xdot_lin = np.array([model.calc_xdot_from_obs(o) for o in obs])
    xdot_geo = np.array([mc.geo_x_dot_to_linear(gd.dynamics(None, mc.obs_to_geo_model(o), mc.action_to_input(env, o[16:]))) for o in obs])
    x_lin = np.array([mc.obs_to_lin_model(o) for o in obs])
    x_free = x_lin + rng.normal(size=x_lin.shape) * 0.3                # calc_xdot on a state of its own
    xdot_free = np.array([model.calc_xdot(x, o[16:]) for x, o in zip(x_free, obs)])
    A0, B0 = model.A.copy(), model.B.copy()
    model.A, model.B = model.Ahat, model.Bhat                          # what a caller gets who swaps in the 'noisy' pair
    xdot_free_hat = np.array([model.calc_xdot(x, o[16:]) for x, o in zip(x_free, obs)])
    rpy = rng.uniform(-np.pi, np.pi, size=(128, 3))
    Rr = np.array([mc.rpy_to_rot(r) for r in rpy])
    # geo_model_to_obs on rotations that exercise all four branches of scipy's from_matrix (largest of R00, R11, R22, trace)
    eul = np.concatenate([rng.uniform(-np.pi, np.pi, size=(192, 3)),
                          np.array([[np.pi, 0, 0], [0, np.pi, 0], [0, 0, np.pi], [3.0, 0.1, -0.1], [0.1, 3.0, 0.1], [0.1, -0.1, 3.0]])])
    x18 = np.zeros((eul.shape[0], 18))
    x18[:, 0:3] = rng.normal(size=(eul.shape[0], 3))
    x18[:, 3:12] = Rotation.from_euler("xyz", eul).as_matrix().reshape(-1, 9)
    x18[:, 12:] = rng.normal(size=(eul.shape[0], 6))
    obs16 = np.array([mc.geo_model_to_obs(x) for x in x18])
    np.savez_compressed(OUT + "/compare_models.npz", obs=obs, A=A0, B=B0, Ahat=model.Ahat, Bhat=model.Bhat, xdot_lin=xdot_lin, xdot_geo=xdot_geo,
                        x_lin=x_lin, x_free=x_free, xdot_free=xdot_free, xdot_free_hat=xdot_free_hat, dyn_m=gd.m, dyn_g=gd.g,
                        dyn_J=np.diag(gd.J), rpy=rpy, R_of_rpy=Rr, x18=x18, obs16=obs16, **META)
    clipped = int(((obs[:, 16:20] > env.MAX_RPM) | (obs[:, 16:20] < 0)).sum())
    print("compare_models", xdot_lin.shape, xdot_geo.shape, "clipped rpm entries", clipped, "quat w<0 rows", int((obs16[:, 6] < 0).sum()))


def trajectory_cases(T):
    """The same constructor arguments are used for the reference classes (minting) and for the oracle /
    GPU classes (tests): T is a namespace with Lemniscate, Circle, Line, Wait, Compound, Rotate."""
    from scipy.spatial.transform import Rotation as Rot
    Rz = Rot.from_euler("xyz", [0.2, -0.3, 0.9]).as_matrix()
    a, b, c = np.array([0.0, 0.0, 0.5]), np.array([1.5, -0.5, 1.0]), np.array([1.5, 2.0, 1.0])
    return {
        "circle": T.Circle(r=0.8, v=0.6, center=np.array([0.2, -0.1, 0.7]), yaw_rate=0.4),
        "circle_rev": T.Circle(r=1.5, v=1.0, center=np.array([0, 0, 1.0]), yaw_rate=-0.7, revolutions=2),
        "wait": T.Wait(position=np.array([0.3, 0.4, 0.5]), duration=2.0, yaw=0.6),
        "line_long": T.Line(start=a, end=b + np.array([3.0, 0, 0]), speed=0.5),
        "line_short": T.Line(start=a, end=a + np.array([0.1, 0.05, 0.0]), speed=1.0),
        "line_s0": T.Line(start=b, end=c + np.array([0, 4.0, 0]), speed=1.0, s0=0.3, sf=0.2),
        "compound": T.Compound([T.Line(start=a, end=b, speed=.5), T.Wait(duration=1, position=b),
                                T.Line(start=b, end=c, speed=1), T.Line(start=c, end=b, speed=1)]),     # EnvGeometric.py:543-550
        "compound_mixed": T.Compound([T.Wait(position=a, duration=0.5, yaw=0.1), T.Lemniscate(a=0.5, omega=1.0, center=a, yaw_rate=0.2),
                                      T.Circle(r=0.5, v=0.5, center=a)]),
        "rotate": T.Rotate(T.Lemniscate(a=1.0, omega=1.5, center=np.array([0, 0, .5]), yaw_rate=0.3, phase_shift=0.4), Rz, np.array([0.1, 0.2, 0.5])),
        "rotate_compound": T.Rotate(T.Compound([T.Line(start=a, end=b, speed=.7), T.Circle(r=0.4, v=0.3, center=b)]), Rz, b),
    }


def mint_trajectories():
    import trajectories as TR     # importable as a package (no third-party imports)
    T = types.SimpleNamespace(Lemniscate=TR.Lemniscate, Circle=TR.CircleTrajectory, Line=TR.LineTrajectory, Wait=TR.WaitTrajectory,
                              Compound=TR.CompoundTrajectory, Rotate=TR.RotateTrajectory)
    cases = trajectory_cases(T)
    out = {}
    for name, tr in cases.items():
        tt = tr.get_total_time()
        ts = np.concatenate([np.linspace(0, 1.25 * tt, 161), [tt, tt * (1 - 1e-9)]])
        rows = np.zeros((len(ts), 11))
        for k, t in enumerate(ts):
            pos, vel, acc, yaw, om = tr(float(t))
            rows[k] = np.hstack([pos, vel, acc * np.ones(3), yaw, om])
        out[name + "_t"], out[name + "_out"], out[name + "_total"] = ts, rows, tt
    np.savez_compressed(OUT + "/trajectories.npz", names=np.array(list(cases)), **out, **META)
    print("trajectories", list(cases))


if __name__ == "__main__":
    ref = load_reference()
    if len(sys.argv) > 1:                      # mint only the named fixtures: python mint_golden.py compare_models ...
        for name in sys.argv[1:]:
            globals()["mint_" + name](ref)
        sys.exit(0)
    mint_lemniscate(ref)
    mint_geometric(ref)
    mint_mixer(ref)
    mint_dynamics(ref)
    mint_dyn_wrench_accel(ref)
    mint_attitude_flow(ref)
    mint_euler_convention(ref)
    mint_closed_loop_reference_in_the_loop(ref)
    mint_closed_loop_lqr_reference_in_the_loop(ref)
    mint_cbf(ref)
    mint_thrust_omega()
    mint_lqr_omega(ref)
    mint_lqr_yank_omega(ref)
    mint_lqr12(ref)
    mint_compare_models(ref)
    sys.path.insert(0, REF)
    mint_trajectories()
